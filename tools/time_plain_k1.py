import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768"))
variants = [int(v) for v in os.environ.get("VARIANTS", "10821,10441,10281,20821,20441").split(",")]
pads = [int(v) for v in os.environ.get("PADS", "0,16").split(",")]
for rnd in range(2):
    for pad in pads:
        for v in variants:
            with pkg.CGSolver(gemv_variant=v, lda_pad=pad) as s:
                s.generate_lap2d_matrix(n)
                ms = s.probe_time_gemv(30)
            print(json.dumps(dict(n=n, variant=v, pad=pad, plain_k1_ms=ms, GBs=8.0*(n*n+2*n)/ms/1e6)), flush=True)
