# Full GPU check of the tree: the parity suite, the driver's bench line, smoke().  Output under gpurun_out/r04b/.
set -e
mkdir -p gpurun_out/r04b
python -m pytest tests -m gpu -x -q > gpurun_out/r04b/gputest.log 2>&1 || { tail -30 gpurun_out/r04b/gputest.log; exit 1; }
tail -3 gpurun_out/r04b/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04b/smoke.log 2>&1 || { tail -20 gpurun_out/r04b/smoke.log; exit 1; }
grep smoke gpurun_out/r04b/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04b/bench20.json 2> gpurun_out/r04b/bench20.err
cut -c1-300 gpurun_out/r04b/bench20.json
