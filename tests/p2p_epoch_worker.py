"""Worker for tests/test_gpu_p2p.py::test_p2p_epoch_boundaries: `world` OS processes on ONE GPU over the CGX_COMM_P2P
mailboxes.  For each form of the fused exchange (flag words, tagged words) ONE context solves the same small problem again
and again while the test hook cgx_probe_set_p2p_epoch moves the 64-bit epoch counter of the exchange channels forward to just
below a boundary, so that the solve runs ACROSS it -- after the previous solve left its set-up / verification all-gathers
(plain doubles: x, the initial Ap) wherever the mailbox layout puts them:
  2^19, 0xFFF80000       round 3's tag = epoch ^ 0xFFF80000 stopped being a NaN pattern / became 0 there
  2^32 - 1, 2 (2^32 - 1) the 32-bit tag of the tagged form wraps (tag = 1 + epoch mod (2^32 - 1))
  2^32                   the epoch's low 32 bits wrap (crossed by the same solve as 2^32 - 1)
Every solve must give the bits of the first one, on every rank, in both forms.  argv: n iters out.json"""
import hashlib
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

M = 2 ** 32 - 1
BOUNDARIES = [2 ** 19, 0xFFF80000, M, 2 * M]   # the solve across M = 2^32 - 1 also crosses 2^32


def main():
    n, iters, out_path = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert torch.cuda.is_available()
    pkg = g.load_package()
    record = {"world": world, "n": n, "iters": iters, "forms": {}}
    digests = {}
    for form, tagged in (("flag", False), ("tagged", True)):
        s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=world, rank=rank, device=0, p2p_timeout_ms=20000, p2p_tagged=tagged)
        mine = torch.tensor(list(s.p2p_export()), dtype=torch.uint8)
        allh = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allh, mine)
        s.p2p_import(b"".join(bytes(t.tolist()) for t in allh))
        dist.barrier()
        ok = s.p2p_selftest(8)
        dist.barrier()                          # launchers agree on `ok` before the next layout (include/cgx.h)
        s.generate_lap2d_matrix(n)
        s.set_max_iter(iters)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)

        def solve():
            x = np.zeros(n)
            r = s.solve(x)
            return hashlib.sha256(x.tobytes() + np.float64(r["residual_prev"]).tobytes() + np.float64(r["x_norm"]).tobytes()).hexdigest(), r

        base, r0 = solve()
        runs = [{"start": s._p2p_epoch(1) - iters, "digest": base, "k": r0["iterations"]}]
        for b in BOUNDARIES:
            start = b - 4                       # the solve's exchanges are start+1 ... start+iters: it crosses b
            dist.barrier()                      # a quiet point on every rank
            s._set_p2p_epoch(1, start)
            if tagged:
                s._set_p2p_epoch(0, start + 1)  # the plain all-gathers of a tagged context count on channel 0
            d, r = solve()
            runs.append({"start": start, "end": s._p2p_epoch(1), "digest": d, "k": r["iterations"]})
        try:                                    # backwards is refused
            s._set_p2p_epoch(1, 5)
            refused = False
        except pkg.CgxError:
            refused = True
        digests[form] = [q["digest"] for q in runs]
        record["forms"][form] = {"selftest_ok": bool(ok), "runs": runs, "backwards_refused": refused}
        s.close()
        dist.barrier()
    mine = json.dumps(digests, sort_keys=True)
    every = [None] * world
    dist.all_gather_object(every, mine)
    record["ranks_agree"] = all(e == every[0] for e in every)
    if rank == 0:
        json.dump(record, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
