"""first_solve.py without torch in the process (what the cgsolver binary sees).  (dev tool)"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CGX_NO_TORCH"] = "1"   # keep torch out of the process (cgx.lib() would import it first)
import __graft_entry__ as g
pkg = g.load_package()
for n in (2048, 4096, 8192):
    with pkg.CGSolver() as s:
        t0 = time.perf_counter(); s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); tg = time.perf_counter() - t0
        out = []
        for rep in range(3):
            x = np.zeros(n); t1 = time.perf_counter(); r = s.solve(x); out.append((time.perf_counter() - t1) * 1e3)
        print("n=%5d: set-up %.2f ms; solve #1 %.2f ms #2 %.2f ms #3 %.2f ms; loop_s %.2f ms k=%d"
              % (n, tg * 1e3, out[0], out[1], out[2], r["seconds_loop"] * 1e3, r["iterations"]), flush=True)
