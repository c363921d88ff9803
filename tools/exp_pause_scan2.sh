#!/bin/bash
# EXPERIMENT (closed, profiles/r05_nowatch/pause_scan2.txt; the macros it set are gone): as tools/exp_pause_scan.sh, for 2048 < n <= 4096 (S = 5 ... 8).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_nowatch
mkdir -p $OUT
cd $R/conjugate-gradient_amd
for cfg in "18 16 7" "20 16 7" "22 16 7" "24 16 7" "26 16 7" "28 16 7"; do
  set -- $cfg
  rm -f build/cgx_resident.o
  make -s EXTRA="-DCGX_P56=$1 -DCGX_P78=$2 -DCGX_WMIN=$3" libcgx.so > $OUT/build.log 2>&1
  echo "== pause S=5,6: $1, S=7,8: $2, watched word from S = $3" | tee -a $OUT/pause_scan2.txt
  (cd $R && SIZES= TIMING=2100,2304,2560,2700,2896,3072 timeout -k 10 200 python3 tools/resident_check.py 2>&1 | grep resident_us | python3 -c "
import sys, json
print([(d['n'], d['resident_us_per_iteration']) for d in (json.loads(l) for l in sys.stdin)])") | tee -a $OUT/pause_scan2.txt
done
rm -f build/cgx_resident.o
make -s libcgx.so > $OUT/build.log 2>&1
