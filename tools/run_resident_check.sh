# The LDS-resident solver's tests, parity + time per iteration against the per-launch path, and a short fuzz run in one GPU call.
set -e
mkdir -p gpurun_out/r04b
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > gpurun_out/r04b/resident_pytest.log 2>&1 || { tail -40 gpurun_out/r04b/resident_pytest.log; exit 1; }
tail -3 gpurun_out/r04b/resident_pytest.log
SIZES=7,2049,2896,4096 TIMING=${TIMING:-1024,2048,2560,2896,3072,3584,4096} timeout -k 10 400 python tools/resident_check.py > gpurun_out/r04b/resident_check.jsonl 2> gpurun_out/r04b/resident_check.err || { tail -20 gpurun_out/r04b/resident_check.err; exit 1; }
grep speedup gpurun_out/r04b/resident_check.jsonl
timeout -k 10 400 python tools/fuzz_resident.py ${FUZZ_SECONDS:-40} 11 > gpurun_out/r04b/fuzz_resident.txt 2>&1 || { tail -20 gpurun_out/r04b/fuzz_resident.txt; exit 1; }
tail -3 gpurun_out/r04b/fuzz_resident.txt
