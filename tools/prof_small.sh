#!/bin/bash
# kernel durations at the reference's small sizes (is the loop launch-bound or kernel-bound?)
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for n in 1024 2048 4096; do
  SIZES=$n VARIANTS=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small_$n -- python3 $R/tools/small_n.py > /tmp/prof_small_$n.log 2>&1
  python3 - "$(find /tmp/prof_small_$n -name '*kernel_stats.csv' | head -1)" $n <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print("N=%s %-40s calls=%s avg=%.2f us" % (sys.argv[2], r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
