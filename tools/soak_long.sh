#!/bin/bash
# Long soak of the fused exchange (both forms) with real processes sharing one MI355X; every line of progress goes to
# gpurun_out/r03_soak_long.txt.  About 15 minutes.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/${SOAK_OUT:-r03_soak_long.txt}
: > $OUT
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=1
TR="python3 -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
run() { echo "### $*" >> $OUT; timeout -k 10 "$1" "${@:2}" 2>&1 | grep -E "exchanges ok|soak done|DISAGREE|Error|error" >> $OUT; echo "rc=${PIPESTATUS[0]}" >> $OUT; tail -2 $OUT; }
run 420 $TR --nproc-per-node 4 --master-port 29811 $R/tools/p2p_soak.py 4096 10000000 2000 0 1
run 330 $TR --nproc-per-node 3 --master-port 29812 $R/tools/p2p_soak.py 8192 2500000 1500 0 1
run 240 $TR --nproc-per-node 4 --master-port 29813 $R/tools/p2p_soak.py 4096 5000000 2000 0 0
run 120 $TR --nproc-per-node 2 --master-port 29814 $R/tools/p2p_soak.py 1000 2000000 400 0 1
