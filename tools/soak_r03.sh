#!/bin/bash
# Round 3 robustness runs on one MI355X: randomised parity sweep (loopback, all K1 forms incl. the column pieces), the
# fault-injection walk, and the soak of the chunked fused exchange with real processes sharing the GPU.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r03_soak.txt
: > $OUT
run() { echo "### $*" >> $OUT; timeout -k 10 "$1" "${@:2}" >> $OUT 2>&1; echo "rc=$?" >> $OUT; }
run 200 python3 $R/tools/fuzz_parity.py 150 31
run 200 python3 $R/tools/fuzz_parity.py 150 32
run 400 python3 $R/tools/leak_check.py
TR="python3 -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=1
run 300 $TR --nproc-per-node 4 --master-port 29801 $R/tools/p2p_soak.py 4096 1000000 2000 0
run 300 $TR --nproc-per-node 3 --master-port 29802 $R/tools/p2p_soak.py 8192 300000 1500 0
run 300 $TR --nproc-per-node 4 --master-port 29803 $R/tools/p2p_soak.py 4096 300000 2000 1
run 300 $TR --nproc-per-node 4 --master-port 29804 $R/tools/p2p_soak.py 4096 1000000 2000 0 1
run 300 $TR --nproc-per-node 3 --master-port 29805 $R/tools/p2p_soak.py 8192 300000 1500 0 1
grep -E "^###|done|rc=|no leak|MISMATCH|DISAGREE|Error|error" $OUT | tail -40
