#!/bin/bash
# Round 5 bench evidence on one MI355X: the default line (500 steps), the driver's line (--steps 20 --warmup 5) plain and under
# rocprofv3 kernel stats, its PMC passes (FETCH_SIZE, WRITE_SIZE: separate, --kernel-trace only), the mid-size lines
# (--matrix-size 8192 and 10000: BASELINE config 2's size), and ONE self-launched 2-rank line with both ranks on the one GPU
# (still complete, with transport_calibration_ms_per_iteration / k1_per_rank / update_kernel.per_rank).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_bench
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default_n32768_line.json 2> $OUT/bench_default.err
echo "default line done"
python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench20_line.json 2> $OUT/bench20.err
rm -rf /tmp/prof_b20; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b20 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $OUT/bench20_line_under_rocprofv3.json 2>> $OUT/bench20.err
cp "$(find /tmp/prof_b20 -name '*kernel_stats.csv' | head -1)" $OUT/bench20_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmcb_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcb_$c -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > /tmp/pmcb_$c.json 2> /tmp/pmcb_$c.log
  python3 - "$(find /tmp/pmcb_$c -name '*counter_collection.csv' | head -1)" $c $OUT/bench20_${c}_counter_collection.csv <<'PY'
import csv, sys
# the timed region's kernel only (the line's reference_sizes leg launches fused K1s of other shapes as well): 4096 workgroups x 256
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_gemv_colsplit<8, 2, 4, 1, false, true>" in r["Kernel_Name"] and r["Grid_Size"] == "1048576" and r["Counter_Name"] == sys.argv[2]][:64]
with open(sys.argv[3], "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
v = [float(r["Counter_Value"]) for r in rows]
print(sys.argv[2], "mean per fused K1 launch:", sum(v) / len(v), "KB over", len(v), "launches")
PY
done
echo "driver line + profiles done"
for n in 8192 10000; do python3 $R/bench.py --matrix-size $n --steps 400 --warmup 100 > $OUT/bench_n${n}_line.json 2> $OUT/bench_n$n.err; done
echo "mid-size lines done"
CGX_BENCH_BACKEND=gloo python3 $R/bench.py --gpus 2 --steps 100 --warmup 20 --cpu-baseline-iters 3 > $OUT/selflaunch_2ranks_one_gpu.json 2> $OUT/selflaunch2.err
echo "self-launch line done"
python3 - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    rf = d.get("roofline", {})
    print("%-40s n_gpus=%s value=%.2f ms/step=%.4f K1 median=%s frac=%s traffic=%s transport=%s calib=%s" % (
        os.path.basename(f), d.get("n_gpus"), d.get("value"), d.get("ms_per_step"), rf.get("median_launch_ms"), rf.get("frac"), rf.get("traffic_over_algorithmic"),
        d.get("config", {}).get("transport"), "yes" if "transport_calibration_ms_per_iteration" in json.dumps(d) else "no"))
    for r in d.get("reference_sizes", []) if isinstance(d.get("reference_sizes"), list) else []:
        print("    n=%d default %.2f us (%s, %.3f of peak) per-launch %.2f us (%.3f)" % (r["n"], r["default"]["us_per_iteration"], r["default"]["kernel"], r["default"]["hbm_roofline_frac"], r["per_launch"]["us_per_iteration"], r["per_launch"]["hbm_roofline_frac"]))
PY
