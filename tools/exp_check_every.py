"""What the host's poll of `done` costs on the per-launch path: cgx_solve to convergence at several check_every (iterations between polls)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
for n in [int(v) for v in os.environ.get("SIZES", "10000,12288,16384").split(",")]:
    for every in (16, 64, 256, 100000):
        with pkg.CGSolver(gemv_variant=-1, check_every=every) as s:
            s.generate_lap2d_matrix(n); s.tolerance(1e-10); s.init_source_term(1.0 / n)
            best = 1e9
            for _ in range(3):
                x = np.zeros(n)
                t0 = time.perf_counter(); r = s.solve(x); t1 = time.perf_counter()
                best = min(best, t1 - t0)
            print(json.dumps({"n": n, "check_every": every, "iterations": r["iterations"], "solve_ms": round(best * 1e3, 3), "us_per_iteration": round(best * 1e6 / r["iterations"], 2), "loop_ms": round(r["seconds_loop"] * 1e3, 3)}), flush=True)
