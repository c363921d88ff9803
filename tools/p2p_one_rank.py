import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); sep = os.environ.get("SEP", "0") == "1"
noacq = os.environ.get("NOACQ", "0") == "1"   # A/B of the acquire fence behind the flag wait
s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1, rank=0, p2p_separate_exchange=sep, p2p_no_acquire_fence=noacq)
s.generate_lap2d_matrix(n); s.set_max_iter(10**6); s.tolerance(0.0); s.init_source_term(1.0 / n)
s.solve_begin(np.zeros(n)); s.solve_steps(300); r = s.solve_end(); print(r["iterations"], r["residual_prev"])
