#!/bin/bash
# EXPERIMENT (adopted for n <= 2048, profiles/r05_nowatch/nowatch.txt): the resident kernel without its watched word (after the pause behind the publish
# straight to the gather of all words; what has not arrived is asked for again), for several lengths of the pause.  The switches it used
# (CGX_RES_FIRST_SLEEP at build time, CGX_STREAM_L2_ROWS != 0 = no watched word) are gone: res_pause / res_watch in cgx_resident.hip.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_nowatch
mkdir -p $OUT
cd $R/conjugate-gradient_amd
for sl in 16 20 24 28 36; do
  rm -f build/cgx_resident.o
  make -s EXTRA="-DCGX_RES_FIRST_SLEEP=$sl" libcgx.so > $OUT/build.log 2>&1
  for nw in 0 1; do
    echo "== pause $sl no_watch $nw" | tee -a $OUT/nowatch.txt
    (cd $R && CGX_STREAM_L2_ROWS=$nw SIZES= TIMING=256,512,1024,1448,2048,2896,4096 timeout -k 10 200 python3 tools/resident_check.py 2>&1 | grep resident_us | python3 -c "
import sys, json
print([(d['n'], d['resident_us_per_iteration']) for d in (json.loads(l) for l in sys.stdin)])") | tee -a $OUT/nowatch.txt
  done
done
rm -f build/cgx_resident.o
make -s libcgx.so > $OUT/build.log 2>&1
