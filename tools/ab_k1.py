"""A/B of K1 shapes with a long warm-up and several interleaved rounds; prints every round (noise is visible)."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768"))
P = int(os.environ.get("SHARDS", "1"))
variants = [int(v) for v in os.environ.get("VARIANTS", "10821,20821").split(",")]
pads = [int(v) for v in os.environ.get("PADS", "0").split(",")]
rounds = int(os.environ.get("ROUNDS", "5"))
steps = int(os.environ.get("STEPS", "60"))
mode = pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF
solvers = {}
for pad in pads:
    for v in variants:
        s = pkg.CGSolver(comm_mode=mode, nranks=P, gemv_variant=v, lda_pad=pad, profile_gemv=1)
        s.generate_lap2d_matrix(n); s.set_max_iter(10**7); s.tolerance(0.0); s.init_source_term(1.0 / n)
        s.solve_begin(np.zeros(n)); s.solve_steps(20)
        solvers[(v, pad)] = s
# warm the clocks
t0 = time.time()
while time.time() - t0 < float(os.environ.get("WARM_S", "2")):
    for s in solvers.values(): s.solve_steps(20)
hist = {k: [] for k in solvers}
for r in range(rounds):
    for k, s in solvers.items():
        plain = s.probe_time_gemv(20) if os.environ.get("PLAIN", "1") == "1" and P == 1 else float("nan")
        torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_steps(steps); t1 = time.perf_counter()
        # gemv stats of the last steps call are only returned by solve_end; use wall per iteration here
        hist[k].append((plain, (t1 - t0) / steps * 1e3))
for k, s in solvers.items():
    res = s.solve_end(); s.close()
    pl = sorted(h[0] for h in hist[k]); it = sorted(h[1] for h in hist[k])
    print(json.dumps(dict(n=n, shards=P, variant=k[0], pad=k[1], plain_ms_med=pl[len(pl)//2], plain_ms_min=pl[0],
                          iter_ms_med=it[len(it)//2], iter_ms_min=it[0], fused_k1_ms_last=res["gemv_ms_avg"],
                          fused_k1_ms_median=res["gemv_ms_median"], fused_k1_ms_min=res["gemv_ms_min"],
                          plain_all=[round(x, 4) for x in (h[0] for h in hist[k])], iter_all=[round(h[1], 4) for h in hist[k]])), flush=True)
