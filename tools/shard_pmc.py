"""P logical row blocks of N=32768 on one GPU, a few iterations: the per-shard K1 launches a real rank would make,
for rocprofv3 --pmc passes (dev tool).  python3 tools/shard_pmc.py SHARDS [ITERS]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
P = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 24
n = 32768
with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK, nranks=P) as s:
    s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(iters); s.tolerance(0.0)
    print(s.solve(np.zeros(n))["iterations"])
