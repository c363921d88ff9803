import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = 32768
for P in (8, 4, 2):
    for v in (10821, 10421, 10441, 10811):
        with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=v) as s:
            s.generate_lap2d_matrix(n)
            for _ in range(3): s.probe_time_gemv(20)
            ms = min(s.probe_time_gemv(30) for _ in range(3))
        print("P=%d variant=%d plain K1 back-to-back: %.4f ms/launch  (%.1f GB/s)" % (P, v, ms, 8.0*(n/P*n+n+n/P)/ms/1e6), flush=True)
