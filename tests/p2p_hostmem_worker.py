"""Worker for tests/test_gpu_p2p.py::test_p2p_processes_over_shared_host_memory: `world` OS processes on ONE GPU, the same
CGX_COMM_P2P solve twice -- over the usual device mailboxes (hipIpc: on one GPU a "peer's" mailbox is this GPU's own HBM) and
over mailboxes in POSIX shared HOST memory (test hook cgx_probe_p2p_host_mailboxes): every store of every rank then leaves
the GPU over PCIe and every poll and load comes back over it, between separate processes.  Both must pass the self-test and
give bit-identical solves on every rank.  argv: n iters out.json tagged(0|1)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    n, iters, out_path, tagged = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1"
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert torch.cuda.is_available()
    pkg = g.load_package()
    prefix = "cgx_test_%d_%d" % (int(os.environ.get("MASTER_PORT", "0")), os.getppid())
    runs = {}
    for where in ("device", "host"):
        s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=world, rank=rank, device=0, p2p_timeout_ms=20000, p2p_tagged=tagged)
        if where == "device":
            mine = torch.tensor(list(s.p2p_export()), dtype=torch.uint8)
            allh = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allh, mine)
            s.p2p_import(b"".join(bytes(t.tolist()) for t in allh))
        else:
            s._host_mailboxes(prefix, 0)
            dist.barrier()                      # every segment exists
            s._host_mailboxes(prefix, 1)
        dist.barrier()
        ok = s.p2p_selftest(16)
        dist.barrier()                          # launchers agree on `ok` before the next layout (include/cgx.h)
        s.generate_lap2d_matrix(n)
        s.set_max_iter(iters)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        x = np.zeros(n)
        dist.barrier()
        res = s.solve(x)
        x2 = np.zeros(n)
        res2 = s.solve(x2)                      # a second solve on the same context (epochs go on, slots are reused)
        runs[where] = (x, res, bool(ok), bool(np.array_equal(x, x2) and res["residual_prev"] == res2["residual_prev"]))
        s.close()
        dist.barrier()
    (xd, rd, okd, repd), (xh, rh, okh, reph) = runs["device"], runs["host"]
    same = bool(np.array_equal(xd, xh) and rd["residual_prev"] == rh["residual_prev"] and rd["x_norm"] == rh["x_norm"])
    every = [None] * world
    dist.all_gather_object(every, (xh.tobytes(), rh["residual_prev"], same, okd, okh, repd, reph))
    if rank == 0:
        O = g.load_oracle()
        xo, ro = O.solve_lap2d(n, iters, 0.0, world)
        json.dump({"world": world, "n": n, "tagged": tagged, "selftest_device": all(e[3] for e in every), "selftest_host": all(e[4] for e in every),
                   "host_equals_device_on_every_rank": all(e[2] for e in every), "ranks_agree": all(e[0] == every[0][0] and e[1] == every[0][1] for e in every),
                   "second_solve_same_bits": all(e[5] and e[6] for e in every),
                   "dx_oracle": float(np.linalg.norm(xh - xo) / np.linalg.norm(xo)), "k": rh["iterations"],
                   "loop_s_device": rd["seconds_loop"], "loop_s_host": rh["seconds_loop"]}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
