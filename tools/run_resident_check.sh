set -e
mkdir -p gpurun_out/r04b
for sc in agent system; do
CGX_RESIDENT_SCOPE=$sc CGX_RESIDENT_PROFILE=1 SIZES=7 TIMING=256,1024,1448,2048 timeout -k 10 400 python tools/resident_check.py > gpurun_out/r04b/resident_prof_$sc.jsonl 2> gpurun_out/r04b/resident_prof_$sc.err || { tail -20 gpurun_out/r04b/resident_prof_$sc.err; exit 1; }
echo "scope=$sc"; grep "resident profile" gpurun_out/r04b/resident_prof_$sc.err | awk 'NR%4==0' 
CGX_RESIDENT_SCOPE=$sc SIZES=7 timeout -k 10 400 python tools/resident_check.py > gpurun_out/r04b/resident_check_$sc.jsonl 2> gpurun_out/r04b/resident_check.err || { tail -20 gpurun_out/r04b/resident_check.err; exit 1; }
grep speedup gpurun_out/r04b/resident_check_$sc.jsonl
done
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > gpurun_out/r04b/resident_pytest.log 2>&1 || { tail -40 gpurun_out/r04b/resident_pytest.log; exit 1; }
tail -3 gpurun_out/r04b/resident_pytest.log
