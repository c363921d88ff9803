#!/bin/bash
# HBM traffic of K1 per launch from rocprofv3 PMC counters, collected as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE passes, with --kernel-trace only; on gfx950 FETCH_SIZE counts the 128-B requests of
# a wide coalesced read at 64 B, so read bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact.
#   nranks 1     : the driver's command, bench.py --steps 20 --warmup 5 --no-cpu-baseline (+ TCC hit/miss pass)
#   nranks 2/4/8 : the shard shapes of N=32768 as P logical row blocks on one GPU (tools/shard_pmc.py)
# Writes gpurun_out/k1_hbm_traffic.json (copied to profiles/k1_hbm_traffic.json) and the per-dispatch CSVs under gpurun_out/pmc/.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcb_$tag -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > /tmp/pmcb_$tag.log 2>&1
  cp "$(find /tmp/pmcb_$tag -name '*counter_collection.csv' | head -1)" $R/gpurun_out/pmc/bench20_${tag}_counter_collection.csv
done
for P in 2 4 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_${P}_$c -- python3 $R/tools/shard_pmc.py $P 24 > /tmp/pmcs_${P}_$c.log 2>&1
    cp "$(find /tmp/pmcs_${P}_$c -name '*counter_collection.csv' | head -1)" $R/gpurun_out/pmc/shard_P${P}_${c}_counter_collection.csv
  done
done
python3 - "$R/gpurun_out" <<'PY'
import csv, json, re, sys, os
out = sys.argv[1]
fused = re.compile(r"k_gemv_colsplit<\d+, \d+, \d+, 1[,>]")
def mean(path, counter):
    vals, names = [], set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and fused.search(r["Kernel_Name"]):
            vals.append(float(r["Counter_Value"])); names.add(re.search(r"k_gemv_colsplit<[^>]*>", r["Kernel_Name"]).group(0))
    return sum(vals) / len(vals), len(vals), sorted(names)
n = 32768
rows = []
for P in (1, 2, 4, 8):
    pre = "bench20_" if P == 1 else "shard_P%d_" % P
    f, nf, names = mean(os.path.join(out, "pmc", pre + "FETCH_SIZE_counter_collection.csv"), "FETCH_SIZE")
    w, nw, _ = mean(os.path.join(out, "pmc", pre + "WRITE_SIZE_counter_collection.csv"), "WRITE_SIZE")
    rows_per = n // P
    row = {"n": n, "nranks": P, "kernel": ", ".join(names) + (" on a %d x %d row block (logical shard on one GPU)" % (rows_per, n) if P > 1 else ""),
           "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
           "algorithmic_bytes_per_launch": 8.0 * (rows_per * n + n + rows_per), "launches_sampled": nf}
    row["traffic_over_algorithmic"] = row["hbm_bytes_per_launch"] / row["algorithmic_bytes_per_launch"]
    if P == 1:
        h, _, _ = mean(os.path.join(out, "pmc", "bench20_TCC_HIT_sum_counter_collection.csv"), "TCC_HIT_sum")
        m, _, _ = mean(os.path.join(out, "pmc", "bench20_TCC_HIT_sum_counter_collection.csv"), "TCC_MISS_sum")
        row["TCC_HIT_sum_mean"], row["TCC_MISS_sum_mean"] = h, m
    rows.append(row)
doc = {"_provenance": "tools/pmc_shards.sh on one MI355X, round 2: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_HIT_sum "
                      "TCC_MISS_sum in separate passes with --kernel-trace only. nranks 1: `python3 bench.py --steps 20 --warmup 5 "
                      "--no-cpu-baseline`; nranks 2/4/8: `python3 tools/shard_pmc.py P 24` (P logical row blocks of N=32768 on ONE GPU: the "
                      "kernel and row-block shape a real rank launches). Correction per /opt/skills/guides/MI355X_MICROARCH.md section HBM: on "
                      "gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide coalesced reads, so read bytes = FETCH_SIZE*1024*2; WRITE_SIZE "
                      "is exact. Per-dispatch CSVs: profiles/r02_pmc/. bench.py quotes hbm_bytes_per_launch as roofline.traffic.",
       "rows": rows}
json.dump(doc, open(os.path.join(out, "k1_hbm_traffic.json"), "w"), indent=1)
for r in rows:
    print(r["nranks"], r["kernel"][:60], "traffic/algorithmic = %.4f" % r["traffic_over_algorithmic"], "launches", r["launches_sampled"])
PY
