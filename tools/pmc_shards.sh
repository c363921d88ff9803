#!/bin/bash
# HBM traffic of K1 on the shard shapes of N=32768 (P = 2, 4, 8 logical row blocks on one GPU): FETCH_SIZE and
# WRITE_SIZE in separate passes; writes gpurun_out/k1_shard_traffic.json
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for P in 2 4 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_${P}_$c -- python3 $R/tools/shard_pmc.py $P 24 > /tmp/pmcs_${P}_$c.log 2>&1
  done
done
python3 - "$R/gpurun_out/k1_shard_traffic.json" <<'PY'
import csv, glob, json, sys
rows = []
for P in (2, 4, 8):
    m = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("/tmp/pmcs_%d_%s/**/*counter_collection.csv" % (P, c), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
                if r["Counter_Name"] == c and "k_gemv_colsplit" in r["Kernel_Name"] and ", 1>" in r["Kernel_Name"]]
        m[c] = (sum(vals) / len(vals), len(vals))
    n = 32768
    rows_per = n // P
    rows.append({"n": n, "nranks": P, "kernel": "k_gemv_colsplit<8,2,4,1> on a %d x %d row block (logical shard on one GPU)" % (rows_per, n),
                 "FETCH_SIZE_KB_mean": m["FETCH_SIZE"][0], "WRITE_SIZE_KB_mean": m["WRITE_SIZE"][0],
                 "hbm_bytes_per_launch": m["FETCH_SIZE"][0] * 1024 * 2 + m["WRITE_SIZE"][0] * 1024,
                 "algorithmic_bytes_per_launch": 8.0 * (rows_per * n + n + rows_per), "launches_sampled": m["FETCH_SIZE"][1]})
json.dump({"rows": rows}, open(sys.argv[1], "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
