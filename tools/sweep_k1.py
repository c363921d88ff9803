"""Interleaved A/B sweep of K1 shapes inside the real CG iteration (one process, several rounds).

  python tools/sweep_k1.py --n 32768 --variants 0,10821,20821 --pads 0,16 --rounds 3 --steps 40 [--loopback 8]
Prints per (variant,pad): median/min K1 launch ms (HIP events), GB/s, roofline fraction, ms per iteration.
"""
import argparse, json, os, statistics, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=32768)
ap.add_argument("--variants", default="0")
ap.add_argument("--pads", default="0")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--loopback", type=int, default=1)
ap.add_argument("--rows-only", type=int, default=0, help="unused")
ap.add_argument("--out", default="")
a = ap.parse_args()
pkg = g.load_package()
variants = [int(v) for v in a.variants.split(",")]
pads = [int(v) for v in a.pads.split(",")]
res = {}
for rnd in range(a.rounds):
    for pad in pads:
        for v in variants:
            mode = pkg.COMM_LOOPBACK if a.loopback > 1 else pkg.COMM_SELF
            with pkg.CGSolver(comm_mode=mode, nranks=a.loopback, gemv_variant=v, lda_pad=pad, profile_gemv=1) as s:
                s.generate_lap2d_matrix(a.n); s.set_max_iter(10**6); s.tolerance(0.0); s.init_source_term(1.0 / a.n)
                s.solve_begin(np.zeros(a.n)); s.solve_steps(5)
                torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve_steps(a.steps); t1 = time.perf_counter()
                r = s.solve_end()
            res.setdefault((v, pad), []).append((r["gemv_ms_avg"], r["gemv_ms_min"], (t1 - t0) / a.steps * 1e3, r["gemv_bytes"]))
rows = []
for (v, pad), xs in res.items():
    avg = statistics.median(x[0] for x in xs); mn = min(x[1] for x in xs); it = statistics.median(x[2] for x in xs)
    gbs = xs[0][3] / (avg * 1e-3) / 1e9
    rows.append(dict(n=a.n, shards=a.loopback, variant=v, pad=pad, k1_ms_median=avg, k1_ms_min=mn, k1_GBs=gbs, frac=gbs / 8000, iter_ms=it))
rows.sort(key=lambda r: r["k1_ms_median"])
for r in rows:
    print(json.dumps(r), flush=True)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(rows, open(a.out, "w"), indent=1)
