"""Worker for the bounded-wait test: rank 1 wires up its mailbox and then never takes part in the solve; rank 0
must come back with CGX_ERR_P2P after the configured timeout instead of hanging.  argv: out.json"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    out_path = sys.argv[1]
    tagged = len(sys.argv) > 2 and sys.argv[2] == "1"
    late = len(sys.argv) > 3 and sys.argv[3] == "1"      # the peer disappears AFTER the set-up phase: the fused exchange itself must time out
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pkg = g.load_package()
    s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=world, rank=rank, device=0, p2p_timeout_ms=400, p2p_tagged=tagged)
    mine = torch.tensor(list(s.p2p_export()), dtype=torch.uint8)
    allh = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allh, mine)
    s.p2p_import(b"".join(bytes(t.tolist()) for t in allh))
    dist.barrier()
    n = 1024
    s.generate_lap2d_matrix(n)
    s.init_source_term(1.0 / n)
    verdict = {}
    if late:
        # both ranks run the set-up phase (its exchanges go through the mailbox all-gather kernel) and two iterations; then
        # rank 1 stops stepping: rank 0's next fused update waits for chunks that never come
        s.set_max_iter(1000)
        s.tolerance(0.0)
        s.solve_begin(np.zeros(n))
        s.solve_steps(2)
        dist.barrier()
    if rank == 0:
        t0 = time.time()
        try:
            if late:
                s.solve_steps(50)
            else:
                s.solve(np.zeros(n))
            verdict = {"raised": False}
        except pkg.CgxError as e:
            verdict = {"raised": True, "status": e.status, "seconds": time.time() - t0, "msg": str(e)}
        json.dump(verdict, open(out_path, "w"))
    else:
        time.sleep(3.0)          # alive (mailbox stays mapped) but silent
    dist.barrier()
    s.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
