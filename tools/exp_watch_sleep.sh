#!/bin/bash
# EXPERIMENT (closed: nothing beyond +- 0.3 %, profiles/r05_watch/watch.txt): the streaming kernel's watch loop with a pause between polls and / or
# in front of the first one (s_sleep under CGX_WATCH_SLEEP / CGX_WATCH_FIRST_SLEEP, removed again with the experiment).
# Rebuilds cgx_stream.hip on the GPU box per setting.  Output: gpurun_out/r05_watch/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_watch
mkdir -p $OUT
cd $R/conjugate-gradient_amd
for cfg in "" "-DCGX_WATCH_SLEEP=4" "-DCGX_WATCH_SLEEP=16" "-DCGX_WATCH_SLEEP=48" "-DCGX_WATCH_FIRST_SLEEP=16" "-DCGX_WATCH_FIRST_SLEEP=16 -DCGX_WATCH_SLEEP=16" ""; do
  rm -f build/cgx_stream.o
  make -s EXTRA="$cfg" libcgx.so > $OUT/build.log 2>&1
  echo "== $cfg" | tee -a $OUT/watch.txt
  (cd $R && SIZES= TIMING=4608,5120,6144,8192,9216,10000 VARIANTS=50000 timeout -k 10 200 python3 tools/stream_check.py 2>&1 | python3 -c "
import sys, json
print([(d['n'], d.get('v50000_us'), d['record']['watch_repeats']) for d in (json.loads(l) for l in sys.stdin if l.startswith('{'))])") | tee -a $OUT/watch.txt
done
rm -f build/cgx_stream.o
make -s libcgx.so > $OUT/build.log 2>&1
