#!/bin/bash
# Round 5, final tree: the reference's commands through the library's default (three runs each with --stats), the finite soak, the
# default beside the per-launch path over the mid sizes.  Output: gpurun_out/r05_final/.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_final
mkdir -p $OUT
CG=$R/conjugate-gradient_amd/cgsolver
for n in 1024 2048 4096 8192 10000; do
  rm -f /tmp/fe.txt
  for i in 1 2 3; do $CG --stats $n /tmp/fe.txt >> $OUT/cgsolver_${n}_final.log 2>&1; done
  cat /tmp/fe.txt >> $OUT/cgsolver_${n}_final.log
done
cd $R
python3 tools/soak_finite.py > $OUT/soak_finite_final_build.txt 2>&1
SIZES= TIMING=4200,4608,5120,5632,6144,7168,7680,8192,8704,9216,9500,10000,10240,11000,11264,11500,12288 VARIANTS=0,-1 python3 tools/stream_check.py > $OUT/default_against_per_launch_final.jsonl 2>&1
SIZES= TIMING=256,512,1024,1448,2048,2560,2896,3072,3584,4096 python3 tools/resident_check.py > $OUT/resident_check_final.jsonl 2>&1
tail -1 $OUT/soak_finite_final_build.txt
grep -h "^[0-9]*,1," $OUT/cgsolver_*_final.log | tr '\n' ' '
