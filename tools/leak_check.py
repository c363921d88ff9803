"""Create / set up / solve / destroy many contexts and watch the device's free memory (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
mtx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lap2D_5pt_n100.mtx")
def free_mb(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
torch.zeros(1, device="cuda"); start = free_mb(); lows = []
for it in range(240):
    kind = it % 6
    with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if kind % 2 else pkg.COMM_SELF, nranks=3 if kind % 2 else 1,
                      matrix_format=pkg.MATRIX_BANDED if kind >= 3 else pkg.MATRIX_DENSE, profile_gemv=kind == 2) as s:
        if kind in (0, 3): s.generate_lap2d_matrix(1500 + it)
        elif kind in (1, 4): s.read_matrix(mtx)
        else: s.set_matrix_dense(np.diag(np.full(700, 4.0)) + np.diag(np.full(699, -1.0), 1) + np.diag(np.full(699, -1.0), -1))
        s.init_source_term(1.0 / s.n()); s.set_max_iter(30)
        s.solve(np.zeros(s.n()))
        if it % 7 == 0:   # a refused matrix must not leak either
            try: s.set_matrix_dense(np.ones((300, 300))) if kind >= 3 else None
            except pkg.CgxError: pass
    if it % 40 == 39:
        lows.append(free_mb()); print("after %d contexts: free %.1f MiB (start %.1f)" % (it + 1, lows[-1], start), flush=True)
assert abs(lows[-1] - lows[0]) < 64, lows
print("no leak: free memory stable within %.1f MiB" % abs(lows[-1] - lows[0]))
