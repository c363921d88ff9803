#!/bin/bash
# rocprofv3 kernel trace + stats of a banded run (N = 2^24, 60 iterations); the stats CSV goes to gpurun_out/
set -e
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_banded -- python3 $R/tools/banded_pmc.py 16777216 60 > /tmp/prof_banded.log 2>&1
f=$(find /tmp/prof_banded -name '*kernel_stats.csv' | head -1)
mkdir -p $R/gpurun_out
cp "$f" $R/gpurun_out/banded_kernel_stats.csv
cat $R/gpurun_out/banded_kernel_stats.csv
