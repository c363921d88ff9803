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

# ---- forced-failure leg: every HIP call of a probe / self-test / solve is made to fail in turn
# (cgx_probe_set_fault_after, cgx_internal.h) and the device's free memory must come back each time.
def forced_failures(name, build, act, max_calls):
    base, hit = None, 0
    for k in range(max_calls):
        s = build()
        s._set_fault_after(k)
        try:
            act(s)
            failed = False
        except pkg.CgxError:
            failed = True
        s._set_fault_after(-1)
        s.close()
        hit += failed
        f = free_mb()
        base = f if base is None else base
        assert abs(f - base) < 8, (name, k, f, base)
        if not failed and k > 0:
            break                                             # k is past the last HIP call of the action: all were covered
    print("%s: %d injected failures, free memory back to %.1f MiB each time" % (name, hit, base), flush=True)
    assert hit >= 5, (name, hit)


def mk_self():
    s = pkg.CGSolver(); s.generate_lap2d_matrix(1200); s.init_source_term(1.0 / 1200); s.set_max_iter(20); return s


def mk_p2p():
    s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1); s.generate_lap2d_matrix(1200); s.init_source_term(1.0 / 1200); s.set_max_iter(20); return s


rng = np.random.default_rng(1)
v = [rng.standard_normal(5000) for _ in range(4)]
forced_failures("cgx_probe_vector_ops", mk_self, lambda s: s.probe_vector_ops(0.5, 0.25, v[0], v[1], v[2], v[3]), 80)
forced_failures("cgx_probe_time_gemv", mk_self, lambda s: s.probe_time_gemv(3), 40)
forced_failures("cgx_p2p_selftest", mk_p2p, lambda s: s.p2p_selftest(2), 60)
forced_failures("cgx_solve (P2P, one rank)", mk_p2p, lambda s: s.solve(np.zeros(1200)), 400)
print("forced failures: no leak")
