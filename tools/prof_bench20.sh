#!/bin/bash
# rocprofv3 kernel stats of exactly the driver's command (bench.py --steps 20 --warmup 5), for comparison with the
# roofline object of the same command's JSON line.  Run on the GPU box: tools/prof_bench20.sh <tag>
set -e
tag=${1:-r02}
cd "$(dirname "$0")/.."
root=$PWD
mkdir -p gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -o bench20 -- \
    python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $root/gpurun_out/prof_$tag/bench20_line.json 2> $root/gpurun_out/prof_$tag/bench20.err
find $root/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $root/gpurun_out/prof_$tag/kernel_stats.csv
