# The LDS-resident solver's tests and a short fuzz run in one GPU call.  Output under gpurun_out/r04b/.
set -e
mkdir -p gpurun_out/r04b
timeout -k 10 900 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > gpurun_out/r04b/resident_pytest.log 2>&1 || { tail -40 gpurun_out/r04b/resident_pytest.log; exit 1; }
tail -3 gpurun_out/r04b/resident_pytest.log
timeout -k 10 400 python tools/fuzz_resident.py ${FUZZ_SECONDS:-30} 11 > gpurun_out/r04b/fuzz_resident.txt 2>&1 || { tail -20 gpurun_out/r04b/fuzz_resident.txt; exit 1; }
tail -3 gpurun_out/r04b/fuzz_resident.txt
