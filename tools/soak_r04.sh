#!/bin/bash
# Round 4 soak of the fused exchange after its changes (producer form of the flag exchange; tagged words: own slots for plain
# all-gathers, new tag, scrub on re-layout), real processes sharing one MI355X: device mailboxes and mailboxes in shared HOST
# memory (every rank's traffic over PCIe), and a tagged run that starts 1000 exchanges below the wrap of the 32-bit tag.
# Progress lines go to gpurun_out/r04_soak.txt.  About 10 minutes.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_soak.txt
: > $OUT
export MASTER_ADDR=127.0.0.1 OMP_NUM_THREADS=1
TR="python3 -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
run() { echo "### $*" >> $OUT; timeout -k 10 "$1" "${@:2}" 2>&1 | grep -E "exchanges ok|soak done|DISAGREE|Error|error" >> $OUT; echo "rc=${PIPESTATUS[0]}" >> $OUT; tail -2 $OUT; }
run 150 $TR --nproc-per-node 4 --master-port 29821 $R/tools/p2p_soak.py 4096 2000000 2000 0 0
run 150 $TR --nproc-per-node 4 --master-port 29822 $R/tools/p2p_soak.py 4096 3000000 2000 0 1
run 120 $TR --nproc-per-node 3 --master-port 29823 $R/tools/p2p_soak.py 4096 600000 1500 0 0 1
run 120 $TR --nproc-per-node 3 --master-port 29824 $R/tools/p2p_soak.py 4096 400000 1500 0 1 1
run 120 $TR --nproc-per-node 3 --master-port 29825 $R/tools/p2p_soak.py 8192 600000 700 0 1 0 4294966295
run 90 $TR --nproc-per-node 2 --master-port 29826 $R/tools/p2p_soak.py 1000 400000 300 0 1 1 8589933590
echo "### fuzz_parity 150 s" >> $OUT
timeout -k 10 200 python3 $R/tools/fuzz_parity.py 150 41 2>&1 | tail -3 >> $OUT
tail -4 $OUT
