"""Per-iteration time at the reference's own experiment sizes (code/MPI/cg.run: N = 1024..8192) for several K1 shapes
(40000 = the resident persistent kernel where it fits, n <= 4096; -1 = the per-launch default shape)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
sizes = [int(v) for v in os.environ.get("SIZES", "1024,2048,4096,8192,10000,16384").split(",")]
variants = [int(v) for v in os.environ.get("VARIANTS", "40000,-1,10821,10441,10281,10241,10181,10421,10811").split(",")]
for n in sizes:
    rows = []
    for v in variants:
        if v == 40000 and n > 4096:
            continue          # the resident kernel does not take this size
        try:
            probe = pkg.CGSolver(gemv_variant=v); probe.generate_lap2d_matrix(n); probe.init_source_term(1.0 / n); probe.set_max_iter(2)
            probe.solve(np.zeros(n)); probe.close()
        except Exception:
            continue          # a shape the library does not build (or not for this size)
        with pkg.CGSolver(gemv_variant=v) as s:
            s.generate_lap2d_matrix(n); s.set_max_iter(10**6); s.tolerance(0.0); s.init_source_term(1.0 / n)
            s.solve_begin(np.zeros(n)); s.solve_steps(200)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize(); t0 = __import__("time").perf_counter(); s.solve_steps(400); t1 = __import__("time").perf_counter()
                best = min(best, (t1 - t0) / 400 * 1e6)
            s.solve_end()
        rows.append((best, v))
    rows.sort()
    print("N=%5d  " % n + "  ".join("%d:%.1fus" % (v, t) for t, v in rows), flush=True)
