#!/bin/bash
# Round 4 bench evidence on one MI355X: the default line (500 steps), the driver's line (--steps 20 --warmup 5) under
# rocprofv3 kernel stats, the weak-scaling base point (N=16384, 200 steps; configs[4] at P=1) with kernel stats, and the
# self-launched 2- and 4-rank lines with all ranks on the one GPU (gloo control plane; IPC mailboxes).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default_n32768.json 2> $OUT/bench_default.err
echo "default line done"
rm -rf /tmp/prof_b20; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $OUT/bench20_line_under_rocprofv3.json 2> $OUT/bench20.err
cp "$(find /tmp/prof_b20 -name '*kernel_stats.csv' | head -1)" $OUT/bench20_kernel_stats.csv
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench20_line.json 2>> $OUT/bench20.err
echo "driver line done"
rm -rf /tmp/prof_weak; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_weak -- python3 $R/bench.py --mode weak --steps 200 --no-cpu-baseline --no-live-pmc > $OUT/weak_base_n16384_line_under_rocprofv3.json 2> $OUT/weak.err
cp "$(find /tmp/prof_weak -name '*kernel_stats.csv' | head -1)" $OUT/weak_base_n16384_kernel_stats.csv
python3 $R/bench.py --mode weak --steps 200 > $OUT/weak_base_n16384_line.json 2>> $OUT/weak.err
echo "weak base done"
CGX_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 100 --warmup 20 --cpu-baseline-iters 3 > $OUT/selflaunch_2ranks_one_gpu.json 2> $OUT/selflaunch2.err
CGX_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 4 --steps 100 --warmup 20 --cpu-baseline-iters 3 > $OUT/selflaunch_4ranks_one_gpu.json 2> $OUT/selflaunch4.err
echo "self-launch lines done"
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    rf = d.get("roofline", {})
    print("%-46s n_gpus=%s value=%s ms/step=%s K1 median=%s frac=%s transport=%s dev_window=%s" % (
        os.path.basename(f), d.get("n_gpus"), d.get("value"), d.get("ms_per_step"), rf.get("median_launch_ms"), rf.get("frac"),
        d.get("config", {}).get("transport"), d.get("device_window_ms_per_step")))
PY
