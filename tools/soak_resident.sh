#!/bin/bash
# LIVENESS soak (NaN payload; the finite soak is tools/soak_finite.py) of the LDS-resident solver on one MI355X: a long randomised sweep (tools/fuzz_resident.py) and two runs of ONE solve of
# 15 M iterations each (tol = 0, n = 1024: 229 launches of 65 536 iterations, ~50 s) that must agree bit for bit.  (Far behind
# convergence rsold underflows to 0 and alpha = 0/0, cg.cc:107, as in the reference: from then on the payload of the exchange
# is NaN -- this run soaks the exchange and its tags, 15 M epochs per run, not the arithmetic.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/r04_resident_soak
mkdir -p $OUT
timeout -k 10 500 python3 $R/tools/fuzz_resident.py ${FUZZ_SECONDS:-240} 2026 > $OUT/fuzz_resident_long.txt 2>&1 || { tail -20 $OUT/fuzz_resident_long.txt; exit 1; }
tail -2 $OUT/fuzz_resident_long.txt
timeout -k 10 500 python3 - > $OUT/long_solve.txt 2>&1 <<PY || { tail -20 $OUT/long_solve.txt; exit 1; }
import sys, time, hashlib
import numpy as np, torch
sys.path.insert(0, "$R")
import __graft_entry__ as g
pkg = g.load_package()
n, iters = 1024, 15_000_000
out = []
for run in range(2):
    with pkg.CGSolver(gemv_variant=40000) as s:
        s.generate_lap2d_matrix(n); s.set_max_iter(iters); s.tolerance(0.0); s.init_source_term(1.0 / n)
        x = np.zeros(n); t0 = time.perf_counter(); r = s.solve(x); dt = time.perf_counter() - t0
    out.append((hashlib.sha256(x.tobytes()).hexdigest(), r["iterations"]))
    print("run %d: %d iterations in %.1f s = %.3f us per iteration, residual %.6e, finite %s, sha256(x) %s" % (
        run, r["iterations"], dt, dt / r["iterations"] * 1e6, r["residual_prev"], bool(np.isfinite(x).all()), out[-1][0][:16]), flush=True)
assert out[0] == out[1], out
print("long solve: both runs bit-identical")
PY
cat $OUT/long_solve.txt | grep -v amdgpu.ids
