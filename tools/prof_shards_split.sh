#!/bin/bash
# rocprofv3 kernel stats + PMC traffic of the iteration on the shard shapes of N=32768 with the XCD-affine column split
# that the fused P2P transport uses by default (P=8: split 8, P=4: 4, P=2: 2), as P logical row blocks on one GPU
# (explicit variants; the loopback iteration then also runs k_combine_ap, which the fused P2P update does not need).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/shards_split
for PV in 8:10825 4:10824 2:10823; do
  P=${PV%%:*}; export VARIANT=${PV##*:}
  SHARDS=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_shs_$P -- python3 $R/tools/loopback_steps.py > /tmp/prof_shs_$P.log 2>&1
  cp "$(find /tmp/prof_shs_$P -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/shards_split/loopback_P${P}_split_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    SHARDS=$P rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcss_${P}_$c -- python3 $R/tools/loopback_steps.py > /tmp/pmcss_${P}_$c.log 2>&1
    cp "$(find /tmp/pmcss_${P}_$c -name '*counter_collection.csv' | head -1)" $R/gpurun_out/shards_split/shard_P${P}_split_${c}_counter_collection.csv
  done
done
python3 - $R/gpurun_out/shards_split <<'PY'
import csv, glob, sys, os, re
d = sys.argv[1]
for f in sorted(glob.glob(d + "/*kernel_stats.csv")):
    print(os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:5]:
        print("   %-52s calls=%-6s avg=%.2f us" % (r["Name"][:52], r["Calls"], float(r["AverageNs"]) / 1e3))
fused = re.compile(r"k_gemv_colsplit<\d+, \d+, \d+, 1[,>]")
for P in (8, 4, 2):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open("%s/shard_P%d_split_%s_counter_collection.csv" % (d, P, c)))
                if r["Counter_Name"] == c and fused.search(r["Kernel_Name"])]
        tot[c] = sum(vals) / len(vals)
    rows = 32768 // P
    hbm = tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024
    print("P=%d split K1: traffic / algorithmic = %.4f" % (P, hbm / (8.0 * (rows * 32768 + 32768 + rows))))
PY
