"""Opt-in banded storage: time per iteration, K1 (banded mat-vec) duration from HIP events, and HBM fractions.
Not the BASELINE metric (that is defined on the dense GEMV: bench.py); this is the measurement row of DESIGN.md's
banded section.   python tools/banded_bench.py [N ...]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
sizes = [int(a) for a in sys.argv[1:]] or [10000, 32768, 1 << 20, 1 << 24, 1 << 26]
variant = int(os.environ.get("VARIANT", "0"))     # 30001: direct K1b, 30002: LDS windows, 0: library default (by size)
rows = []
for n in sizes:
    iters = 2000 if n <= (1 << 20) else (400 if n <= (1 << 24) else 150)
    with pkg.CGSolver(matrix_format=pkg.MATRIX_BANDED, profile_gemv=4, gemv_variant=variant) as s:
        s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(10**9); s.tolerance(0.0)
        nd = len(s.matrix_format(0)[1])
        s.solve_begin(np.zeros(n)); s.solve_steps(iters // 4)
        best, k1 = 1e9, 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_steps(iters); t1 = time.perf_counter()
            best = min(best, (t1 - t0) / iters)
        r = s.solve_end()
        k1 = r["gemv_ms_avg"] * 1e-3
    k1_bytes = 8.0 * n * (nd + 4)            # fused K1: diagonals + p_old + r in, p_new + Ap out
    it_bytes = 8.0 * n * (nd + 4 + 6)        # + K3: Ap, r, p, x in, r, x out
    row = {"n": n, "ndiag": nd, "variant": variant, "k1_us_median": r["gemv_ms_median"] * 1e3, "us_per_iteration": best * 1e6, "iterations_per_s": 1.0 / best,
           "k1_us": k1 * 1e6, "k1_GBs": k1_bytes / k1 / 1e9, "k1_frac_of_8TBs": k1_bytes / k1 / 8e12,
           "iteration_GBs": it_bytes / best / 1e9, "iteration_frac_of_8TBs": it_bytes / best / 8e12,
           "dense_block_bytes": 8.0 * n * n, "banded_block_bytes": 8.0 * n * nd}
    rows.append(row)
    print(json.dumps(row), flush=True)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "banded_bench_%d.json" % variant)
os.makedirs(os.path.dirname(out), exist_ok=True)
json.dump({"rows": rows}, open(out, "w"), indent=1)
