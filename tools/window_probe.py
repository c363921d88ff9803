#!/usr/bin/env python3
"""Where does a short timed window (bench.py --steps 20 --warmup 5) lose time?  

Runs the sequence of bench.py's run() -- solve_begin, W warm-up loop bodies, sync, K timed loop bodies -- on one GPU
and prints, per window: the wall time of the timed solve_steps call, every K1 sample (first launch included), and the
fixed cost obtained from windows of different K.  Legs:
  first    every launch sampled, the first one too: is the first launch after a sync an outlier in event time?
  fixed    no events at all, K = 20/40/100/500: wall = a + b*K, a = fixed cost of the window
  idle     a sleep between the sync and the timed call: does a longer idle make the first launch slower?
Output: one JSON object on stdout (and gpurun_out/window_probe.json when the directory exists).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import __graft_entry__ as g


def window(s, n, warmup, steps, idle=0.0):
    x = np.zeros(n)
    s.set_max_iter(warmup + steps)
    s.tolerance(0.0)
    s.solve_begin(x)
    s.solve_steps(warmup)
    torch.cuda.synchronize()
    if idle:
        time.sleep(idle)
    t0 = time.perf_counter()
    s.solve_steps(steps)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    samples = s.gemv_samples()
    res = s.solve_end(x)
    return wall, samples, res


def main():
    n = int(os.environ.get("N", "32768"))
    pkg = g.load_package()
    out = {"n": n}

    def fresh(**kw):
        s = pkg.CGSolver(device=0, **kw)
        s.generate_lap2d_matrix(n)
        s.init_source_term(1.0 / n)
        window(s, n, 0, 400)          # pre-warm: clocks settled
        return s

    s = fresh(profile_gemv=1, profile_first=True)
    leg = []
    for rep in range(3):
        wall, smp, res = window(s, n, 5, 20)
        leg.append({"wall_ms": wall * 1e3, "ms_per_step": wall * 1e3 / 20, "first_ms": float(smp[0]),
                    "rest_median_ms": float(np.median(smp[1:])), "rest_max_ms": float(smp[1:].max()),
                    "sum_samples_ms": float(smp.sum()), "samples": [round(float(v), 4) for v in smp]})
    out["first"] = leg
    leg = []
    for idle in (0.0, 0.001, 0.01, 0.1, 1.0):
        wall, smp, res = window(s, n, 5, 20, idle)
        leg.append({"idle_s": idle, "wall_ms": wall * 1e3, "first_ms": float(smp[0]), "second_ms": float(smp[1]),
                    "rest_median_ms": float(np.median(smp[1:]))})
    out["idle"] = leg
    s.close()

    s = fresh(profile_gemv=0)
    leg = []
    for steps in (1, 2, 5, 20, 20, 40, 100, 500):
        wall, smp, res = window(s, n, 5, steps)
        leg.append({"steps": steps, "wall_ms": wall * 1e3, "ms_per_step": wall * 1e3 / steps})
    ks = np.array([r["steps"] for r in leg if r["steps"] >= 20], dtype=float)
    ws = np.array([r["wall_ms"] for r in leg if r["steps"] >= 20])
    b, a = np.polyfit(ks, ws, 1)
    out["fixed"] = {"windows": leg, "fit_ms_per_step": b, "fit_fixed_ms": a}
    s.close()

    # what bench.py does by default on a short window: every launch but the first is a sample.  Two forms of the event
    # pair: bound to the dispatch (default) and as marker packets around it (round 1's form).
    for name, markers in (("bench_like_dispatch_events", False), ("bench_like_marker_events", True)):
        s = fresh(profile_gemv=1, profile_markers=markers)
        leg = []
        for rep in range(3):
            wall, smp, res = window(s, n, 5, 20)
            leg.append({"wall_ms": wall * 1e3, "ms_per_step": wall * 1e3 / 20, "median_ms": res["gemv_ms_median"],
                        "avg_ms": res["gemv_ms_avg"], "min_ms": res["gemv_ms_min"], "max_ms": res["gemv_ms_max"],
                        "samples": res["gemv_launches"], "discarded": res["gemv_discarded"]})
        out[name] = leg
        s.close()

    txt = json.dumps(out, indent=1)
    print(txt)
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        open(os.path.join(d, "window_probe.json"), "w").write(txt)


if __name__ == "__main__":
    main()
