"""P logical row blocks of N on one GPU (or one block, SHARDS=1), 300 + 100 iterations: the kernels and row-block shapes a
real rank launches, for rocprofv3 runs (dev tool).  Env: N, SHARDS, VARIANT, PROFILE, MODE=loopback|p2p (p2p: one rank over
its own mailbox, the fused update).  Prints one JSON line with the K1 plan the library chose and the K1 event mean."""
import json, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); P = int(os.environ.get("SHARDS", "8")); v = int(os.environ.get("VARIANT", "0"))
mode = os.environ.get("MODE", "loopback")
if mode == "p2p":
    s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1, rank=0, gemv_variant=v, profile_gemv=int(os.environ.get("PROFILE", "0")),
                     p2p_tagged=os.environ.get("TAGGED", "0") == "1")
    assert s.p2p_selftest(8)
else:
    s = pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=v, profile_gemv=int(os.environ.get("PROFILE", "0")))
s.generate_lap2d_matrix(n); s.set_max_iter(10**6); s.tolerance(0.0); s.init_source_term(1.0 / n)
import time
s.solve_begin(np.zeros(n)); s.solve_steps(300); t0 = time.perf_counter(); s.solve_steps(100); t1 = time.perf_counter(); r = s.solve_end()
print(json.dumps({"n": n, "shards": P, "mode": mode, "variant": v, "plan": s.gemv_plan(0), "gemv_ms_avg": r["gemv_ms_avg"],
                  "combine": os.environ.get("CGX_K1_COMBINE", "0"), "wall_us_per_iteration_all_blocks": (t1 - t0) * 1e4,
                  "residual_prev": r["residual_prev"]}))
