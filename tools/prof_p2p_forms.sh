#!/bin/bash
# one-rank fused P2P iteration, both forms of the exchange, REPS separate-process rocprofv3 runs each (dev tool)
R=$(cd "$(dirname "$0")/.." && pwd); REPS=${1:-2}
cd /tmp && export TMPDIR=/tmp
for T in 0 1; do
  for rep in $(seq 1 $REPS); do
    rm -rf /tmp/tg_$T; TAGGED=$T MODE=p2p SHARDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tg_$T -- python3 $R/tools/loopback_steps.py > /tmp/tg_$T.log 2>&1 || tail -5 /tmp/tg_$T.log
    python3 - "$(find /tmp/tg_$T -name '*kernel_stats.csv' | head -1)" $T $rep <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "update_xr_p2p" in r["Name"] and int(r["Calls"]) >= 100:
        print("TAGGED=%s rep %s: %-40s calls=%s avg=%.2f us min=%.2f" % (sys.argv[2], sys.argv[3], r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
  done
done
