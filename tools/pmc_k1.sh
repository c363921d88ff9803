#!/bin/bash
# HBM traffic of K1 per launch from rocprofv3 PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, with --kernel-trace only; on gfx950 FETCH_SIZE counts the 128-B requests of
# a wide coalesced read at 64 B, so read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact.
#   nranks 1     : the driver's command, bench.py --steps 20 --warmup 5 --no-cpu-baseline (+ TCC hit/miss pass)
#   nranks 2/4/8 : the shard shapes of N=32768 as P logical row blocks on one GPU (tools/loopback_steps.py), with the K1
#                  plan the library chooses by default for a multi-rank run (the same for every transport since round 3)
#   weak base    : N=16384 on one GPU (configs[4], P=1)
# ONE script writes the whole of profiles/k1_hbm_traffic.json: every row carries the K1 plan (R, U, light, split) it was
# collected on, and bench.py quotes a row only for a run whose plan matches.
# Writes gpurun_out/k1_hbm_traffic.json and the per-dispatch CSVs (cut down to the fused K1 dispatches) under gpurun_out/pmc/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmcb_$tag
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcb_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc > $R/gpurun_out/pmc/bench20_$tag.json 2> /tmp/pmcb_$tag.log
  cp "$(find /tmp/pmcb_$tag -name '*counter_collection.csv' | head -1)" /tmp/pmc_raw_bench20_${tag}.csv
done
for PN in 2:32768 4:32768 8:32768 1:16384; do
  P=${PN%%:*}; N=${PN##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcs_${P}_${N}_$c
    N=$N SHARDS=$P rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_${P}_${N}_$c -- python3 $R/tools/loopback_steps.py > $R/gpurun_out/pmc/shard_P${P}_N${N}_$c.json 2> /tmp/pmcs.log
    cp "$(find /tmp/pmcs_${P}_${N}_$c -name '*counter_collection.csv' | head -1)" /tmp/pmc_raw_shard_P${P}_N${N}_${c}.csv
  done
done
python3 - "$R/gpurun_out" <<'PY'
import csv, json, re, sys, os
out = sys.argv[1]
fused = re.compile(r"k_gemv_colsplit<(\d+), (\d+), \d+, 1(?:, (true|false))?(?:, (?:true|false))?>")
def collect(raw, counter, dst):
    """mean of `counter` over the fused K1 dispatches; writes the CSV cut down to those rows"""
    vals, names, keep, header = [], set(), [], None
    with open(raw) as fh:
        rd = csv.DictReader(fh)
        header = rd.fieldnames
        for r in rd:
            m = fused.search(r["Kernel_Name"])
            if not m:
                continue
            keep.append(r)
            if r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"])); names.add(m.group(0))
    with open(dst, "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=header); w.writeheader(); w.writerows(keep)
    return sum(vals) / len(vals), len(vals), sorted(names)
def plan_of(path):
    doc = json.loads([l for l in open(path) if l.startswith("{")][-1])
    pl = doc["plan"] if "plan" in doc else doc["config"]["k1_plan"]
    return {k: int(pl[k]) for k in ("R", "U", "light", "split")}
rows = []
for P, n in ((1, 32768), (2, 32768), (4, 32768), (8, 32768), (1, 16384)):
    pre = "bench20_" if (P, n) == (1, 32768) else "shard_P%d_N%d_" % (P, n)
    f, nf, names = collect("/tmp/pmc_raw_%sFETCH_SIZE.csv" % pre, "FETCH_SIZE", os.path.join(out, "pmc", pre + "FETCH_SIZE_counter_collection.csv"))
    w, nw, _ = collect("/tmp/pmc_raw_%sWRITE_SIZE.csv" % pre, "WRITE_SIZE", os.path.join(out, "pmc", pre + "WRITE_SIZE_counter_collection.csv"))
    rows_per = n // P
    row = {"n": n, "nranks": P, "plan": plan_of(os.path.join(out, "pmc", pre + "FETCH_SIZE.json")),
           "kernel": ", ".join(names) + (" on a %d x %d row block (logical shard on one GPU)" % (rows_per, n) if P > 1 else ""),
           "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
           "algorithmic_bytes_per_launch": 8.0 * (rows_per * n + n + rows_per), "launches_sampled": nf}
    row["traffic_over_algorithmic"] = row["hbm_bytes_per_launch"] / row["algorithmic_bytes_per_launch"]
    if (P, n) == (1, 32768):
        h, _, _ = collect("/tmp/pmc_raw_bench20_TCC_HIT_sum.csv", "TCC_HIT_sum", os.path.join(out, "pmc", "bench20_TCC_HIT_sum_counter_collection.csv"))
        m, _, _ = collect("/tmp/pmc_raw_bench20_TCC_HIT_sum.csv", "TCC_MISS_sum", os.path.join(out, "pmc", "bench20_TCC_HIT_sum_counter_collection.csv"))
        row["TCC_HIT_sum_mean"], row["TCC_MISS_sum_mean"] = h, m
    rows.append(row)
doc = {"_provenance": "tools/pmc_k1.sh on one MI355X (the ONE script that writes this file): rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc "
                      "TCC_HIT_sum TCC_MISS_sum in separate passes with --kernel-trace only. n=32768 nranks 1: `python3 bench.py --steps 20 "
                      "--warmup 5 --no-cpu-baseline`; nranks 2/4/8 and n=16384: `N=.. SHARDS=P python3 tools/loopback_steps.py` (P logical row "
                      "blocks on ONE GPU: the kernel, plan and row-block shape a real rank launches by default). Correction per "
                      "/opt/skills/guides/MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced "
                      "reads, so read bytes = FETCH_SIZE*1024*2; WRITE_SIZE is exact. Each row names the K1 plan it was collected on "
                      "(cgx_get_gemv_plan); bench.py quotes hbm_bytes_per_launch as roofline.traffic only for a run with the same plan. "
                      "Per-dispatch CSVs (fused K1 dispatches only) beside this round's copy of the file under profiles/.",
       "rows": rows}
json.dump(doc, open(os.path.join(out, "k1_hbm_traffic.json"), "w"), indent=1)
for r in rows:
    print(r["n"], r["nranks"], r["plan"], r["kernel"][:48], "traffic/algorithmic = %.4f" % r["traffic_over_algorithmic"], "launches", r["launches_sampled"])
PY
