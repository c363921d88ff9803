#!/bin/bash
# rocprofv3 kernel stats of the iteration on the shard shapes of N=32768 (P logical row blocks on one GPU) and of the
# one-rank P2P iteration (K3 with the exchange inside): the evidence behind DESIGN.md section 6's P=8 estimate.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/shards
for P in 8 4 2; do
  SHARDS=$P rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sh_$P -- python3 $R/tools/loopback_steps.py > /tmp/prof_sh_$P.log 2>&1
  cp "$(find /tmp/prof_sh_$P -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/shards/loopback_P${P}_kernel_stats.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_p2p1 -- python3 $R/tools/p2p_one_rank.py > /tmp/prof_p2p1.log 2>&1
cp "$(find /tmp/prof_p2p1 -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/shards/p2p_one_rank_kernel_stats.csv
python3 - $R/gpurun_out/shards <<'PY'
import csv, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.csv")):
    print(os.path.basename(f))
    for r in list(csv.DictReader(open(f)))[:4]:
        print("   %-44s calls=%-6s avg=%.2f us" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
