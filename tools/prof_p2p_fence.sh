#!/bin/bash
# Cost of the system-scope acquire fence behind the flag wait of k_update_xr_p2p: the one-rank
# P2P iteration at N=32768 under rocprofv3, with the fence (default) and without it (NOACQ=1).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/p2p_fence
for leg in acq noacq; do
  if [ $leg = noacq ]; then export NOACQ=1; else export NOACQ=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_fence_$leg -- python3 $R/tools/p2p_one_rank.py > /tmp/prof_fence_$leg.log 2>&1
  cp "$(find /tmp/prof_fence_$leg -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/p2p_fence/p2p_one_rank_${leg}_kernel_stats.csv
done
python3 - $R/gpurun_out/p2p_fence <<'PY'
import csv, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.csv")):
    print(os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:3]:
        print("   %-44s calls=%-6s avg=%.2f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
