import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
for tagged in (False, True):
    with pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1, p2p_tagged=tagged) as s:
        lo, hi = 1, 1 << 16
        # the refusal message names the bound the runtime reported
        s._set_resident_limit(0)
        try:
            s.generate_lap2d_matrix(262144 + 256)
        except pkg.CgxError as e:
            print("tagged" if tagged else "flags", str(e)[:300])
