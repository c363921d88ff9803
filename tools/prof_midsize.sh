#!/bin/bash
# Evidence for the band 4096 < N <= 16384: per size, one rocprofv3 kernel trace of `cgsolver N out 300`
# (K1 / K3 durations, the period of an iteration and the gaps between the kernels from the dispatch timestamps) and three
# separate PMC passes (FETCH_SIZE, WRITE_SIZE, TCC_HIT_sum + TCC_MISS_sum; --kernel-trace only, as the guide prescribes);
# then the reference's own commands at its top sizes, and a bench.py line at N = 10000.  Output: gpurun_out/r05_midsize/.
#   CGX_MID_VARIANT   gemv_variant for every run (CGX_GEMV_VARIANT), default -1 = the per-launch path
#   CGX_MID_TAG       suffix of the output files (default "launches")
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_midsize
TAG=${CGX_MID_TAG:-launches}
export CGX_GEMV_VARIANT=${CGX_MID_VARIANT:--1}
SIZES=${SIZES:-"5120 6144 8192 10000 12288"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CG=$R/conjugate-gradient_amd/cgsolver
for n in $SIZES; do
  rm -rf /tmp/mid_kt_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mid_kt_$n -- $CG $n /tmp/mid_out.txt 300 > /tmp/mid_kt_$n.log 2>&1
  cp "$(find /tmp/mid_kt_$n -name '*kernel_trace.csv' | head -1)" /tmp/mid_trace_$n.csv
  cp "$(find /tmp/mid_kt_$n -name '*kernel_stats.csv' | head -1)" $OUT/n${n}_${TAG}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/mid_pmc_${n}_$tag
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/mid_pmc_${n}_$tag -- $CG $n /tmp/mid_out.txt 300 > /tmp/mid_pmc.log 2>&1
    cp "$(find /tmp/mid_pmc_${n}_$tag -name '*counter_collection.csv' | head -1)" /tmp/mid_pmc_${n}_$tag.csv
  done
  echo "collected N=$n"
done
python3 $R/tools/summarise_midsize.py "$TAG" $SIZES > $OUT/summary_${TAG}.md
cat $OUT/summary_${TAG}.md
# the reference's own commands at its top sizes (code/MPI/cg.run, code/CUDA/cg.run) and BASELINE config 2
for cmd in "8192 /tmp/o.txt" "10000 /tmp/o.txt" "$R/tests/golden/lap2D_5pt_n100.mtx 1024 16 true /tmp/o.txt"; do
  name=$(echo $cmd | awk '{print $1}' | xargs basename | sed 's/\.mtx//')
  rm -rf /tmp/mid_full_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mid_full_$name -- $CG $cmd > $OUT/cgsolver_${name}_${TAG}.log 2>&1
  cp "$(find /tmp/mid_full_$name -name '*kernel_stats.csv' | head -1)" $OUT/cgsolver_${name}_${TAG}_kernel_stats.csv
  for i in 1 2 3; do $CG --stats $cmd >> $OUT/cgsolver_${name}_${TAG}.log 2>&1; tail -1 /tmp/o.txt >> $OUT/cgsolver_${name}_${TAG}.log; done
done
cd $R
python3 bench.py --matrix-size 10000 --steps 400 --warmup 100 --variant ${CGX_MID_VARIANT:--1} > $OUT/bench_n10000_${TAG}.json 2> $OUT/bench_n10000_${TAG}.err || tail -5 $OUT/bench_n10000_${TAG}.err
tail -c 1500 $OUT/bench_n10000_${TAG}.json
