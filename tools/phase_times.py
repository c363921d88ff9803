import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
for n in [int(v) for v in os.environ.get("SIZES", "1024,2048,4096,8192,32768").split(",")]:
    with pkg.CGSolver() as s:
        s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(300); s.tolerance(0.0)
        for rep in range(3):
            x = np.zeros(n)
            torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_begin(x); torch.cuda.synchronize(); t1 = time.perf_counter()
            s.solve_steps(300); t2 = time.perf_counter(); r = s.solve_end(x); t3 = time.perf_counter()
            print("n=%5d rep %d: begin %.3f ms  steps(300) %.3f ms  end %.3f ms" % (n, rep, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)
