"""Plain K1 (no iteration head, no r stream, no p store) on the shard shapes, launched back to back like tools/hbm_rows_bw.hip
launches its load-only twin: mean time per launch INCLUDING the kernel boundary (dev tool).
SHARDS=8 VARIANTS=10821,10442 python tools/shard_plain.py"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); P = int(os.environ.get("SHARDS", "8"))
for v in [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]:
    with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=v) as s:
        s.generate_lap2d_matrix(n)
        s.probe_time_gemv(10)
        t = sorted(s.probe_time_gemv(40) for _ in range(5))
        print(json.dumps({"n": n, "shards": P, "variant": v, "plain_ms_per_launch_incl_boundary": t, "median": t[2]}), flush=True)
