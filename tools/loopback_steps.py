import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); P = int(os.environ.get("SHARDS", "8")); v = int(os.environ.get("VARIANT", "0"))
s = pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=v, profile_gemv=int(os.environ.get("PROFILE", "0")))
s.generate_lap2d_matrix(n); s.set_max_iter(10**6); s.tolerance(0.0); s.init_source_term(1.0 / n)
s.solve_begin(np.zeros(n)); s.solve_steps(300); s.solve_steps(100); r = s.solve_end(); print(r["gemv_ms_avg"])
