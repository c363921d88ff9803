# The LDS-resident solver's tests, parity + time per iteration against the per-launch path, the phase profile and a short fuzz run
# in one GPU call.  Output under gpurun_out/r04b/.
set -e
mkdir -p gpurun_out/r04b
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > gpurun_out/r04b/resident_pytest.log 2>&1 || { tail -40 gpurun_out/r04b/resident_pytest.log; exit 1; }
tail -3 gpurun_out/r04b/resident_pytest.log
SIZES=7,2049,2896,4096 TIMING=${TIMING:-1024,2048,2560,2896,3072,3584,4096} timeout -k 10 400 python tools/resident_check.py > gpurun_out/r04b/resident_check.jsonl 2> gpurun_out/r04b/resident_check.err || { tail -20 gpurun_out/r04b/resident_check.err; exit 1; }
grep speedup gpurun_out/r04b/resident_check.jsonl
CGX_RESIDENT_PROFILE=1 SIZES=7 TIMING=2560,3072,3584,4096 timeout -k 10 300 python tools/resident_check.py > /dev/null 2> gpurun_out/r04b/phase_profile.err || true
grep "resident profile" gpurun_out/r04b/phase_profile.err | awk 'NR%4==0'
timeout -k 10 400 python tools/fuzz_resident.py ${FUZZ_SECONDS:-40} 12 > gpurun_out/r04b/fuzz_resident.txt 2>&1 || { tail -20 gpurun_out/r04b/fuzz_resident.txt; exit 1; }
tail -1 gpurun_out/r04b/fuzz_resident.txt
