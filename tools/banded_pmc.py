"""A few banded iterations at one size, for rocprofv3 --pmc passes (dev tool).  python3 tools/banded_pmc.py N ITERS"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n, iters = int(sys.argv[1]), int(sys.argv[2])
with pkg.CGSolver(matrix_format=pkg.MATRIX_BANDED) as s:
    s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(iters); s.tolerance(0.0)
    r = s.solve(np.zeros(n))
    print(r["iterations"], r["residual_prev"])
