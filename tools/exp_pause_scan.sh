#!/bin/bash
# EXPERIMENT (closed, profiles/r05_nowatch/pause_scan.txt; the macros it set are gone, the values are res_pause in cgx_resident.hip): the resident
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_nowatch
mkdir -p $OUT
cd $R/conjugate-gradient_amd
for cfg in "8 16 20" "10 18 22" "12 20 24" "14 22 26" "16 24 28" "18 26 30"; do
  set -- $cfg
  rm -f build/cgx_resident.o
  make -s EXTRA="-DCGX_RES_PAUSE_12=$1 -DCGX_RES_PAUSE_3=$2 -DCGX_RES_PAUSE_4=$3" libcgx.so > $OUT/build.log 2>&1
  echo "== pause S<=2: $1, S=3: $2, S=4: $3" | tee -a $OUT/pause_scan.txt
  (cd $R && SIZES= TIMING=256,512,768,1024,1280,1448,1536,1800,2048 timeout -k 10 200 python3 tools/resident_check.py 2>&1 | grep resident_us | python3 -c "
import sys, json
print([(d['n'], d['resident_us_per_iteration']) for d in (json.loads(l) for l in sys.stdin)])") | tee -a $OUT/pause_scan.txt
done
rm -f build/cgx_resident.o
make -s libcgx.so > $OUT/build.log 2>&1
