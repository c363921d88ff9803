#!/bin/bash
# Does the cache policy of the streamed matrix loads matter where the matrix (or its streamed part) fits the 256 MiB Infinity
# Cache?  Rebuilds the two persistent kernels with each policy (on the GPU box: hipcc is there) and times an iteration.
# Output: gpurun_out/r05_mall/.  The tree is left with the default build.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_mall
mkdir -p $OUT
cd $R/conjugate-gradient_amd
for pol in "2: nt" "0:" "16: sc1" "1: sc0"; do
  aux=${pol%%:*}; txt=${pol#*:}
  rm -f build/cgx_stream.o build/cgx_resident.o
  make -s EXTRA="-DCGX_STREAM_AUX=$aux -DCGX_RES_STREAM_POLICY='\"$txt\"'" libcgx.so > $OUT/build_$aux.log 2>&1
  echo "== policy aux=$aux '$txt'" | tee -a $OUT/stream.txt $OUT/resident.txt
  (cd $R && SIZES= TIMING=4608,5120,5632,5792,6144,7168,8192 VARIANTS=50000 timeout -k 10 300 python3 tools/stream_check.py) >> $OUT/stream.txt 2>&1
  (cd $R && SIZES= TIMING=3072,3584,4096 timeout -k 10 300 python3 tools/resident_check.py) >> $OUT/resident.txt 2>&1
done
rm -f build/cgx_stream.o build/cgx_resident.o
make -s libcgx.so > $OUT/build_default.log 2>&1
cat $OUT/stream.txt $OUT/resident.txt
