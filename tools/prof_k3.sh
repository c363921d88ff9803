#!/bin/bash
# K3 average duration at N=32768 (one GPU and 8 logical shards) under rocprofv3
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
for P in 1 8; do
  SHARDS=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k3_$P -- python3 $R/tools/loopback_steps.py > /tmp/prof_k3_$P.log 2>&1
  python3 - "$(find /tmp/prof_k3_$P -name '*kernel_stats.csv' | head -1)" $P <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "update_xr" in r["Name"] or "colsplit<8, 2, 4, 1>" in r["Name"]:
        print("P=%s %-44s calls=%-6s avg=%.2f us" % (sys.argv[2], r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
