"""Streaming persistent kernel (cgx_stream.hip) against the oracle and against the per-launch path: parity and time per iteration.

PARITY: for every size a fixed-iteration solve and a solve to convergence with gemv_variant = 50000 (the streaming kernel,
also below 4097) against oracle.solve_lap2d.  TIMING: us per iteration (tol = 0, best of 3) of 50000 and of -1 (K1 + K3).
"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from oracle import oracle

sizes = [int(v) for v in os.environ.get("SIZES", "1024,1500,2049,4097,5000,8192,10000").split(",") if v]
timing = [int(v) for v in os.environ.get("TIMING", "5120,6144,8192,10000,12288,16384").split(",") if v]
variants = [int(v) for v in os.environ.get("VARIANTS", "50000,-1").split(",")]
for n in sizes:
    for max_iter, tol in ((min(n, 120), 0.0), (None if n <= 5000 else 300, 1e-10)):
        ref_x, ref = oracle.solve_lap2d(n, max_iter=max_iter, tol=tol)
        row = {"n": n, "max_iter": max_iter, "tol": tol, "oracle_k": ref["iterations"]}
        with pkg.CGSolver(gemv_variant=50000) as s:
            s.generate_lap2d_matrix(n)
            if max_iter is not None: s.set_max_iter(max_iter)
            s.tolerance(tol); s.init_source_term(1.0 / n)
            plan = s.gemv_plan()
            x = np.zeros(n)
            res = s.solve(x)
            rec = s.resident_record()
        dx = float(np.linalg.norm(x - ref_x) / max(np.linalg.norm(ref_x), 1e-300))
        row.update({"variant": plan["variant"], "plan": plan, "k": res["iterations"], "conv": res["converged"], "dx": dx,
                    "res_prev_rel": abs(res["residual_prev"] - ref["residual_prev"]) / max(ref["residual_prev"], 1e-300),
                    "rel_residual": res["rel_residual"], "record": rec})
        print(json.dumps(row), flush=True)
        assert plan["variant"] == 5 and dx < 1e-11, row
for n in timing:
    row = {"n": n}
    for v in variants:
        with pkg.CGSolver(gemv_variant=v) as s:
            s.generate_lap2d_matrix(n); s.set_max_iter(10**8); s.tolerance(0.0); s.init_source_term(1.0 / n)
            s.solve_begin(np.zeros(n)); s.solve_steps(200)
            best = 1e9
            steps = max(200, int(2e5 / (n * n / 1e6)) // 100 * 100)
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_steps(steps); t1 = time.perf_counter()
                best = min(best, (t1 - t0) / steps * 1e6)
            rec = s.resident_record()
            s.solve_end()
        row["v%d_us" % v] = round(best, 2)
        row["v%d_frac" % v] = round(8.0 * (n * n + 2 * n) / (best * 1e-6) / 8e12, 4)
        if v != -1: row["record"] = rec
    print(json.dumps(row), flush=True)
