"""How much does K1 time depend on where the 8 GiB matrix landed?  Same shape, several live allocations."""
import os, sys, json, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); v = int(os.environ.get("VARIANT", "10821")); pad = int(os.environ.get("PAD", "0"))
cnt = int(os.environ.get("COUNT", "6"))
ss = []
for i in range(cnt):
    s = pkg.CGSolver(gemv_variant=v, lda_pad=pad); s.generate_lap2d_matrix(n); ss.append(s)
t0 = time.time()
while time.time() - t0 < 1.5:
    for s in ss: s.probe_time_gemv(10)
for rnd in range(3):
    print(rnd, ["%.4f" % s.probe_time_gemv(20) for s in ss], flush=True)
