"""begin / steps / end of the first and second solve of a process, without torch (dev tool)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CGX_NO_TORCH"] = "1"   # keep torch out of the process (cgx.lib() would import it first)
import __graft_entry__ as g
pkg = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with pkg.CGSolver() as s:
    s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(300); s.tolerance(0.0)
    for rep in range(3):
        x = np.zeros(n)
        t0 = time.perf_counter(); s.solve_begin(x); t1 = time.perf_counter()
        s.solve_steps(100); t2 = time.perf_counter(); s.solve_steps(100); t3 = time.perf_counter(); s.solve_steps(100); t4 = time.perf_counter()
        r = s.solve_end(x); t5 = time.perf_counter()
        print("n=%d rep %d: begin %.3f ms  steps %.3f / %.3f / %.3f ms  end %.3f ms" % (n, rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3), flush=True)
