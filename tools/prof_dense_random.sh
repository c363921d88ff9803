#!/bin/bash
# K1 on dense incompressible data beside the generated (99.985 % zero) matrix, SAME box, same process
# shape: event medians (with rocm-smi clock samples), rocprofv3 kernel stats, PMC FETCH_SIZE / WRITE_SIZE (separate passes,
# --kernel-trace only, as the guide prescribes), for N = 32768 on one block and on the 4096 x 32768 shard shape (8 logical
# row blocks), and the pure-read ceiling microbenchmark on a constant and on a hash-filled buffer.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_dense_random
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -w --offload-arch=gfx950 -O3 $R/tools/hbm_read_bw.hip -o /tmp/hbm_read_bw
for fill in const hash; do
  /tmp/hbm_read_bw 8192 $fill > $OUT/hbm_read_ceiling_8GiB_$fill.txt
  /tmp/hbm_read_bw 1024 $fill > $OUT/hbm_read_ceiling_1GiB_$fill.txt
done
echo "ceiling done"
for P in 1 8; do
  for M in lap2d hash lap2d hash; do
    N=32768 SHARDS=$P MATRIX=$M SMI=1 STEPS=$((P==1 ? 1000 : 4000)) python3 $R/tools/dense_random_rate.py >> $OUT/events_P${P}.jsonl 2>> $OUT/events.err
  done
  for M in lap2d hash; do
    rm -rf /tmp/prof_dr; N=32768 SHARDS=$P MATRIX=$M rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dr -- python3 $R/tools/dense_random_rate.py > $OUT/stats_P${P}_${M}_line.json 2>> $OUT/prof.err
    cp "$(find /tmp/prof_dr -name '*kernel_stats.csv' | head -1)" $OUT/kernel_stats_P${P}_${M}.csv
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf /tmp/pmc_dr; N=32768 SHARDS=$P MATRIX=$M STEPS=100 WARM=20 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_dr -- python3 $R/tools/dense_random_rate.py > $OUT/pmc_P${P}_${M}_${c}_line.json 2>> $OUT/prof.err
      python3 - "$(find /tmp/pmc_dr -name '*counter_collection.csv' | head -1)" $c $P $M >> $OUT/pmc_summary.jsonl <<'PY'
import csv, json, sys
raw, counter, P, M = sys.argv[1:5]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(raw)) if "k_gemv_colsplit" in r["Kernel_Name"] and ", 1," in r["Kernel_Name"] and r["Counter_Name"] == counter]
rows = 32768 // int(P)
alg = 8.0 * (rows * 32768 + 32768 + rows)
mean = sum(vals) / len(vals)
bytes_ = mean * 1024 * (2 if counter == "FETCH_SIZE" else 1)
print(json.dumps({"P": int(P), "matrix": M, "counter": counter, "launches": len(vals), "KB_mean": mean, "bytes_per_launch": bytes_, "over_algorithmic": bytes_ / alg}))
PY
    done
  done
  echo "P=$P done"
done
cat $OUT/events_P1.jsonl $OUT/events_P8.jsonl | python3 -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l); print(d['shards'], d['matrix'], 'K1 median %.4f ms  frac %.4f  finite %s' % (d['k1_median_ms'], d['frac_of_8TBs'], d['finite']), (d['smi'] or [None])[-1])
"
cat $OUT/pmc_summary.jsonl
grep -h "wgs= 4096 U=16 nt=1" $OUT/hbm_read_ceiling_*.txt
