#!/bin/bash
# repeatability of the per-kernel numbers DESIGN.md section 6 is built on.  For every shard shape of
# N=32768 (P = 8, 4, 2 logical row blocks on one GPU) and every K1 plan in question, REPS separate-process rocprofv3
# --kernel-trace --stats runs of the loopback iteration (K1 + k_prefold_ap + K3; the exchange is a device copy), plus the
# one-rank fused P2P iteration in both forms (K1 + k_update_xr_p2p / k_update_xr_p2p_tagged).  Writes gpurun_out/shards_repeat/*.csv and summary.json / summary.md.
# usage: tools/prof_shards_repeat.sh [REPS] ["P:VARIANT ..."]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REPS=${1:-3}
CASES=${2:-"8:0 8:10442 4:0 4:10442 4:10825 2:0 2:10822 2:10824"}
OUT=$R/gpurun_out/shards_repeat
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for PV in $CASES; do
  P=${PV%%:*}; V=${PV##*:}
  for rep in $(seq 1 $REPS); do
    d=/tmp/prof_rep_${P}_${V}_$rep; rm -rf $d
    SHARDS=$P VARIANT=$V rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/loopback_steps.py > $OUT/loopback_P${P}_v${V}_rep$rep.json 2> /tmp/prof_rep.err || { tail -5 /tmp/prof_rep.err; exit 1; }
    cp "$(find $d -name '*kernel_stats.csv' | head -1)" $OUT/loopback_P${P}_v${V}_rep${rep}_kernel_stats.csv
    echo "P=$P variant=$V rep=$rep done"
  done
done
for T in 0 1; do
  name=p2p_one_rank; [ $T = 1 ] && name=p2p_tagged_one_rank
  for rep in $(seq 1 $REPS); do
    d=/tmp/prof_rep_p2p_${T}_$rep; rm -rf $d
    TAGGED=$T MODE=p2p SHARDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/loopback_steps.py > $OUT/${name}_rep$rep.json 2> /tmp/prof_rep.err || { tail -5 /tmp/prof_rep.err; exit 1; }
    cp "$(find $d -name '*kernel_stats.csv' | head -1)" $OUT/${name}_rep${rep}_kernel_stats.csv
  done
done
python3 $R/tools/summarise_repeat.py $OUT
