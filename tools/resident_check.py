"""LDS-resident solver (cgx_resident.hip) against the per-launch path and the oracle: parity and time per iteration.

For every size: a fixed-iteration solve and a solve to convergence with the default choice (resident where it fits) and with
gemv_variant = -1 (the per-launch path), both against oracle.solve_lap2d; then the time per iteration of both (tol = 0).
"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
from oracle import oracle

sizes = [int(v) for v in os.environ.get("SIZES", "2,7,64,300,512,513,1000,1024,1448,2047,2048,2049,2896,3072,3584,4096").split(",") if v]
timing = [int(v) for v in os.environ.get("TIMING", "256,512,1024,1448,2048,2560,2896,3072,3584,4096").split(",")]
out = []
for n in sizes:
    for max_iter, tol in ((min(n, 120), 0.0), (None, 1e-10)):
        ref_x, ref = oracle.solve_lap2d(n, max_iter=max_iter, tol=tol)
        row = {"n": n, "max_iter": max_iter, "tol": tol, "oracle_k": ref["iterations"]}
        for name, v in (("resident", 0), ("launches", -1)):
            with pkg.CGSolver(gemv_variant=v) as s:
                s.generate_lap2d_matrix(n)
                if max_iter is not None: s.set_max_iter(max_iter)
                s.tolerance(tol); s.init_source_term(1.0 / n)
                plan = s.gemv_plan()
                x = np.zeros(n)
                res = s.solve(x)
            dx = float(np.linalg.norm(x - ref_x) / max(np.linalg.norm(ref_x), 1e-300))
            row[name] = {"variant": plan["variant"], "k": res["iterations"], "conv": res["converged"], "dx": dx,
                         "res_prev_rel": abs(res["residual_prev"] - ref["residual_prev"]) / max(ref["residual_prev"], 1e-300),
                         "rel_residual": res["rel_residual"]}
        out.append(row)
        print(json.dumps(row), flush=True)
for n in timing:
    row = {"n": n}
    for name, v in (("resident", 0), ("launches", -1)):
        with pkg.CGSolver(gemv_variant=v) as s:
            s.generate_lap2d_matrix(n); s.set_max_iter(10**8); s.tolerance(0.0); s.init_source_term(1.0 / n)
            s.solve_begin(np.zeros(n)); s.solve_steps(200)
            best = 1e9
            for steps in (2000, 2000, 2000):
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_steps(steps); t1 = time.perf_counter()
                best = min(best, (t1 - t0) / steps * 1e6)
            s.solve_end()
        row[name + "_us_per_iteration"] = round(best, 3)
    row["speedup"] = round(row["launches_us_per_iteration"] / row["resident_us_per_iteration"], 2)
    print(json.dumps(row), flush=True)
