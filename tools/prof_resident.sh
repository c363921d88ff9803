#!/bin/bash
# Round 4 evidence for the LDS-resident solver (csrc/cgx_resident.hip) on one MI355X: parity + time per iteration against
# the per-launch path (tools/resident_check.py), the in-kernel phase profile, rocprofv3 kernel stats of `cgsolver N out
# 2000` for the reference's three small sizes (one k_cg_resident launch of 2000 iterations each: its duration / 2000 is the
# time per iteration on the device's clock), and the reference's result files regenerated (experiments/cg_mi355x.run 1).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_resident
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/tools/resident_check.py > $OUT/resident_check.jsonl 2> $OUT/resident_check.err
grep speedup $OUT/resident_check.jsonl
CGX_RESIDENT_PROFILE=1 SIZES=7 TIMING=256,512,1024,1448,2048 timeout -k 10 300 python3 $R/tools/resident_check.py > /dev/null 2> $OUT/phase_profile.err
grep "resident profile" $OUT/phase_profile.err | awk 'NR%4==0' > $OUT/phase_profile.txt
cat $OUT/phase_profile.txt
for N in 1024 1448 2048 2896 4096; do
  rm -rf /tmp/prof_res_$N
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_res_$N -- $R/conjugate-gradient_amd/cgsolver $N /tmp/res_out_$N.txt 2000 > $OUT/cgsolver_n${N}_2000.txt 2>&1
  cp "$(find /tmp/prof_res_$N -name '*kernel_stats.csv' | head -1)" $OUT/cgsolver_n${N}_2000_kernel_stats.csv
  grep -i "k_cg_resident" $OUT/cgsolver_n${N}_2000_kernel_stats.csv | cut -c1-200
done
echo "results files"
bash $R/experiments/cg_mi355x.run 1 > $OUT/experiments_run.log 2>&1
cat $R/results/strong_scaling.txt $R/results/weak_scaling.txt
mkdir -p $OUT/results && cp $R/results/*.txt $OUT/results/
