#!/bin/bash
# The reference's timing window (all of solve(), cg_main.cc:53-55) at its own sizes, through the persistent kernels: seconds in
# OUTFILE over 5 runs each, and one rocprofv3 kernel trace per size (how many launches a solve is).  Output: gpurun_out/r05_window/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_window
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CG=$R/conjugate-gradient_amd/cgsolver
for n in 1024 1448 2048 2896 4096 8192; do
  rm -f /tmp/w_$n.txt
  for i in 1 2 3 4 5 6; do $CG $n /tmp/w_$n.txt > /dev/null 2>&1; done
  echo "n=$n seconds (6 runs, the first one includes the process's first-use costs): $(cut -d, -f3 /tmp/w_$n.txt | tr '\n' ' ')" | tee -a $OUT/window_seconds.txt
  rm -rf /tmp/w_kt_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/w_kt_$n -- $CG $n /tmp/w_prof.txt > /tmp/w_kt_$n.log 2>&1
  cp "$(find /tmp/w_kt_$n -name '*kernel_stats.csv' | head -1)" $OUT/cgsolver_n${n}_kernel_stats.csv
  echo "   launches of one solve (generator and source term included): $(tail -n +2 $OUT/cgsolver_n${n}_kernel_stats.csv | awk -F, '{gsub(/"/,"",$0); n+=$(NF-6)} END{print n}')"
  cut -c1-90 $OUT/cgsolver_n${n}_kernel_stats.csv | head -8
done
