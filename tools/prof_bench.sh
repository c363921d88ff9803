#!/bin/bash
# the bench line and the rocprofv3 kernel stats of the same program (profiles/r01_bench_*, r01_rocprofv3_kernel_stats_*)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
python3 $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 200 --warmup 50 --no-cpu-baseline > /tmp/prof_bench.log 2>&1
cp "$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/bench_kernel_stats.csv
tail -1 /tmp/prof_bench.log | cut -c1-400
head -5 $R/gpurun_out/bench_kernel_stats.csv | cut -c1-220
