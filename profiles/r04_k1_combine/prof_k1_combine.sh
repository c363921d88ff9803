#!/bin/bash
# k_prefold_ap folded into K1 by arrival tickets (CGX_K1_COMBINE=1) against the prefold kernel, on the
# 4096 x 32768 shard shape (8 logical row blocks on one GPU: the kernels a rank of an 8-GPU run launches on every
# transport but the fused P2P one).  REPS separate-process rocprofv3 --kernel-trace --stats runs each, alternating, SAME box;
# per run also the wall clock per iteration of the loopback loop (K1 + [prefold] + device-copy exchange + K3) x 8 blocks.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
REPS=${1:-3}
OUT=$R/gpurun_out/r04_k1_combine
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rep in $(seq 1 $REPS); do
  for C in 0 1; do
    d=/tmp/prof_comb_${C}_$rep; rm -rf $d
    CGX_K1_COMBINE=$C SHARDS=8 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/loopback_steps.py > $OUT/combine${C}_rep$rep.json 2> /tmp/prof_comb.err || { tail -5 /tmp/prof_comb.err; exit 1; }
    cp "$(find $d -name '*kernel_stats.csv' | head -1)" $OUT/combine${C}_rep${rep}_kernel_stats.csv
    CGX_K1_COMBINE=$C SHARDS=8 WALL=1 python3 $R/tools/loopback_steps.py > $OUT/combine${C}_rep${rep}_wall.json 2>> /tmp/prof_comb.err
    echo "combine=$C rep=$rep done"
  done
done
python3 $R/tools/summarise_repeat.py $OUT
grep -h wall_us $OUT/*_wall.json
