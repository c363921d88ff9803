#!/bin/bash
# EXPERIMENT (closed, profiles/r05_keep/README.md): K1 at N = 32768 with every m-th row group (8 rows, 2 MiB) read with the default
# cache policy, the rest nt.  The kernel-side switch (CGX_KEEP_MOD) was removed again with the experiment: kept as the record of how it was run.
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r05_keep
mkdir -p $OUT
cd $R
for m in 0 128 64 42 32 24 16 0; do
  CGX_KEEP_MOD=$m python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline --no-solve-window --no-live-pmc --no-dense-check --no-reference-sizes > $OUT/keep_$m.json 2> $OUT/keep_$m.err
  python3 - "$m" "$OUT/keep_$m.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("keep_mod", sys.argv[1], "it/s", d["value"], "ms/step", d["ms_per_step"], "K1 frac", d["roofline"]["frac"])
PY
done
