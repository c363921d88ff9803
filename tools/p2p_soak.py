"""Soak of the direct peer exchange: WORLD processes on one GPU, tol = 0, many thousands of exchanges back to back; after
every chunk all ranks must hold bit-identical x and scalars.  (dev tool; launch with torch.distributed.run)
argv: n total_iterations chunk [separate 0|1] [tagged 0|1] [hostmem 0|1] [start_epoch]
hostmem 1: every rank's mailbox in POSIX shared HOST memory (all traffic of all ranks over PCIe, cgx_probe_p2p_host_mailboxes);
start_epoch: the exchange channels' epoch counters are moved there first (e.g. 4294966295 = 1000 below the wrap of the tag)."""
import os, sys, time
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g

n, total, chunk = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
separate = len(sys.argv) > 4 and sys.argv[4] == "1"
tagged = len(sys.argv) > 5 and sys.argv[5] == "1"
hostmem = len(sys.argv) > 6 and sys.argv[6] == "1"
start_epoch = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
pkg = g.load_package()
s = pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=world, rank=rank, device=0, p2p_timeout_ms=20000, p2p_separate_exchange=separate, p2p_tagged=tagged)
if hostmem:
    prefix = "cgx_soak_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    s._host_mailboxes(prefix, 0)
    dist.barrier()
    s._host_mailboxes(prefix, 1)
else:
    mine = torch.tensor(list(s.p2p_export()), dtype=torch.uint8)
    allh = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allh, mine)
    s.p2p_import(b"".join(bytes(t.tolist()) for t in allh))
dist.barrier()
assert s.p2p_selftest(16)
dist.barrier()
if start_epoch:
    for chan in (0, 1):
        s._set_p2p_epoch(chan, max(start_epoch, s._p2p_epoch(chan)))
    dist.barrier()
t0 = time.time(); done = 0; rounds = 0
while done < total:
    # a fresh solve per chunk (rounding noise would otherwise blow up with tol = 0): `chunk` exchanges each
    s.generate_lap2d_matrix(n); s.init_source_term(1.0 / n); s.set_max_iter(chunk); s.tolerance(0.0)
    x = np.zeros(n)
    r = s.solve(x)
    v = torch.tensor(list(x[:: max(1, n // 64)]) + [r["iterations"], r["residual_prev"], r["x_norm"]], dtype=torch.float64)
    vs = [torch.zeros_like(v) for _ in range(world)]
    dist.all_gather(vs, v)
    if not all(torch.equal(vs[0].view(torch.int64), t.view(torch.int64)) for t in vs) or r["iterations"] != chunk:
        print("rank %d: DISAGREEMENT after %d exchanges (round %d)" % (rank, done, rounds), flush=True)
        sys.exit(1)
    done += chunk; rounds += 1
    if rank == 0 and rounds % 20 == 0:
        print("%d exchanges ok, %.0f s" % (done, time.time() - t0), flush=True)
if rank == 0:
    print("soak done: %d exchanges in %d solves, %d ranks, %.0f s, all ranks bit-identical (separate=%d tagged=%d hostmem=%d, epochs %d .. %d)" % (
        done, rounds, world, time.time() - t0, separate, tagged, hostmem, start_epoch, s._p2p_epoch(1)))
s.close()
dist.barrier()
