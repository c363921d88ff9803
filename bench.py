#!/usr/bin/env python3
"""bench.py -- CG iterations/s and K1 HBM-roofline fraction on MI355X (BASELINE.json metric).

    python bench.py                       # 1 GPU, generate_lap2d N=32768, 500 timed iterations (configs[2])
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # row-block over N GPUs (configs[3], strong scaling)

A "step" is one body of the CG loop (code/MPI/cg.cc:96-137 of the reference): one A.p GEMV over this
rank's row block, two dot products, the x/r/p updates and the exchanges (here: ONE per iteration).  Inputs are synthetic and
HBM-resident before the timed region: A = generate_lap2d_matrix(N) built on the device, b = init_source_term(1/N).
W warmup steps, then exactly K steps between barrier + torch.cuda.synchronize(); MAX over ranks; rank 0
prints ONE JSON line.  value = K / t for the whole job (every rank advances the same K iterations).

Extra objects in the line:
  roofline     -- K1 (the GEMV) against the 8 TB/s HBM peak: algorithmic bytes 8*(rows*N + N + rows) per launch
                  divided by the mean launch duration measured with HIP events on the library's own stream.
  cpu_baseline -- the CPU oracle (oracle/cg_oracle.c, a port of the reference's serial path) run for a few
                  iterations of the same workload on one host core (rank 0, 1-GPU runs only).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)      # BASELINE.json configs[2]: fixed 500 iterations
    ap.add_argument("--warmup", type=int, default=100)   # ~0.12 s: lets the clocks settle before the timed region
    ap.add_argument("--matrix-size", dest="n", type=int, default=0, help="matrix size (default 32768; weak mode: floor(16384*sqrt(P)))")
    ap.add_argument("--mode", choices=["strong", "weak"], default="strong")
    ap.add_argument("--variant", type=int, default=0, help="K1 shape override (DESIGN.md)")
    ap.add_argument("--transport", choices=["auto", "p2p", "p2p-sep", "rccl"], default="auto",
                    help="multi-GPU exchange: direct xGMI mailboxes with the exchange folded into K3 (p2p) or as its own "
                         "kernel (p2p-sep), RCCL, or whichever of those works and calibrates fastest on this node (auto)")
    ap.add_argument("--lda-pad", type=int, default=-1)
    ap.add_argument("--cpu-baseline-iters", type=int, default=20)   # 20-50 loop bodies at N=32768, not 500
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile-gemv", action="store_true", help="do not bracket K1 with HIP events")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="event-time every n-th K1 launch (default: 4 on one GPU, 8 on several: the event pair "
                         "costs ~5 us per timed launch, measured)")
    return ap.parse_args()


def pmc_traffic(n, nranks):
    """HBM bytes per K1 launch from a committed rocprofv3 --pmc pass (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "k1_hbm_traffic.json")
    try:
        rows = json.load(open(path))["rows"]
    except Exception:
        return None
    for r in rows:
        if r.get("n") == n and r.get("nranks") == nranks:
            return r.get("hbm_bytes_per_launch")
    return None


def broadcast_bytes(dist, payload, nbytes, device):
    """Rank 0's `payload` (bytes of length nbytes) to every rank of the default process group."""
    import torch
    if dist.get_rank() == 0:
        assert len(payload) == nbytes
        t = torch.tensor(list(payload), dtype=torch.uint8, device=device)
    else:
        t = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    return bytes(t.cpu().tolist())


def cpu_baseline(n, iters):
    import __graft_entry__ as g
    O = g.load_oracle()
    t0 = time.time()
    _, r = O.solve_lap2d(n, iters, 0.0, 1)     # tol 0: never converges, exactly `iters` loop bodies
    wall = time.time() - t0
    out = {
        "value": iters / r["seconds_loop"], "unit": "iterations/s", "cores": 1, "kind": "port",
        "sample": "oracle/cg_oracle.c serial CG, generate_lap2d N=%d, %d of the loop bodies timed (loop only, "
                  "%.1f s; %.1f s incl. building the 8*N^2-byte matrix)" % (n, iters, r["seconds_loop"], wall),
        "gemv_GBs": 8.0 * n * n * iters / r["seconds_loop"] / 1e9,
    }
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    out["host_cpu"] = model
    out["host_cores_available"] = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # context row: all host cores of this box's share, row blocks = threads = the
    # reference's MPI ranks; same arithmetic, bit-identical result
    try:
        cores = min(len(os.sched_getaffinity(0)), 16)
    except AttributeError:
        cores = min(os.cpu_count() or 1, 16)
    if cores > 1:
        O.set_threads(cores)
        _, ra = O.solve_lap2d(n, iters, 0.0, cores)
        O.set_threads(1)
        out["all_cores"] = {"value": iters / ra["seconds_loop"], "unit": "iterations/s", "cores": cores, "kind": "port",
                            "sample": "same, %d row blocks on %d threads" % (cores, cores),
                            "gemv_GBs": 8.0 * n * n * iters / ra["seconds_loop"] / 1e9}
    return out


def main():
    args = parse_args()
    # stdout carries the ONE JSON line and nothing else: whatever the libraries print on fd 1 while the run is
    # going (RCCL's version banner, for one) is sent to stderr instead.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch   # before libcgx: the library then shares the HIP/RCCL instances torch loaded
    import __graft_entry__ as g

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path to time"
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        # more ranks than GPUs is only ever a rehearsal on a one-GPU box (RCCL itself refuses duplicate devices)
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    pkg = g.load_package()

    dist = None
    uid = None
    # Under torch.distributed.run (RANK set) the RCCL path is used even for one rank, so that the launcher
    # plumbing and the collectives can be rehearsed on a one-GPU box; plain `python bench.py` has no comm.
    use_comm = world > 1 or ("RANK" in os.environ and os.environ.get("CGX_BENCH_FORCE_SELF") != "1")
    ctl = "cuda"   # device of the small control-plane tensors
    if use_comm:
        import torch.distributed as dist
        backend = os.environ.get("CGX_BENCH_BACKEND", "nccl")   # "gloo": rehearsal of world > 1 on a one-GPU box
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            ctl = "cpu"

    if args.n:
        n = args.n
    elif args.mode == "weak":
        n = int(math.floor(16384 * math.sqrt(world)))   # code/MPI/cg.run:22-44 rounding rule, N^2/P constant
    else:
        n = 32768
    profile_every = 0 if args.no_profile_gemv else (args.profile_every or (4 if world == 1 else 8))

    def all_ok(flag):
        """True only if `flag` is true on every rank (so that all ranks take the same branch)."""
        if dist is None:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def make_solver(transport):
        """transport: 'self' | 'rccl' | 'p2p' | 'p2p-sep'.  Returns a ready solver or None (same answer on every rank).
        The wire-up is cut into stages; after each one all ranks agree (all_ok) whether to go on, so that a failure on
        one rank can never leave the others inside a different torch.distributed call."""
        common = dict(nranks=world, rank=rank, device=local_rank, gemv_variant=args.variant, lda_pad=args.lda_pad,
                      profile_gemv=profile_every)
        box = {"s": None, "uid": None, "handle": None, "all_handles": None}

        def stage(what, fn):
            ok = True
            try:
                ok = fn() is not False
            except Exception as e:                   # noqa: BLE001 -- any failure means "do not use this transport"
                print("bench.py rank %d: transport %s unavailable (%s): %s" % (rank, transport, what, e),
                      file=sys.stderr, flush=True)
                ok = False
            return all_ok(ok)

        def give_up():
            if box["s"] is not None:
                box["s"].close()
            return None

        if transport == "self":
            def create():
                box["s"] = pkg.CGSolver(comm_mode=pkg.COMM_SELF, **common)
            if not stage("create", create):
                return give_up()
        elif transport == "rccl":
            # replaces mpirun's wire-up (MPI_Init, cg_main.cc:15-20): every rank gets rank 0's RCCL id
            def make_id():
                box["uid"] = pkg.comm_unique_id() if rank == 0 else None
            if not stage("unique id", make_id):
                return give_up()
            uid = broadcast_bytes(dist, box["uid"], pkg.cgx.UNIQUE_ID_BYTES, ctl)

            def create():
                box["s"] = pkg.CGSolver(comm_mode=pkg.COMM_RCCL, unique_id=uid, **common)
            if not stage("ncclCommInitRank", create):
                return give_up()
        else:
            def create():
                box["s"] = pkg.CGSolver(comm_mode=pkg.COMM_P2P, p2p_separate_exchange=(transport == "p2p-sep"), **common)
                box["handle"] = box["s"].p2p_export()
            if not stage("mailbox allocation", create):
                return give_up()
            mine = torch.tensor(list(box["handle"]), dtype=torch.uint8, device=ctl)
            allh = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allh, mine)              # every rank's mailbox handle to every rank

            def attach():
                box["s"].p2p_import(b"".join(bytes(t.cpu().tolist()) for t in allh))
            if not stage("opening the peers' mailboxes", attach):   # also the barrier: every rank has attached
                return give_up()

            def selftest():
                return bool(box["s"].p2p_selftest(32))   # pattern all-gathers, verified, every wait bounded
            if not stage("self-test", selftest):
                return give_up()

        # The problem is set up ONCE per solver: re-allocating the 8 GiB matrix after a free can land on
        # fragmented memory and cost ~3 % of K1 (measured), so calibration and timed run share one allocation.
        def problem():
            box["s"].generate_lap2d_matrix(n)
            box["s"].init_source_term(1.0 / n)
        if not stage("problem set-up", problem):
            return give_up()
        return box["s"]

    def run(s, warmup, steps):
        """W warm-up + K timed loop bodies on solver s.  Returns (elapsed, result) or None.  Every rank makes
        the same sequence of torch.distributed calls whatever fails locally, so a failure cannot desynchronise."""
        ok = True
        x = np.zeros(n)
        try:
            s.set_max_iter(warmup + steps)
            s.tolerance(0.0)               # fixed-iteration run: the break of cg.cc:120 is never taken
            s.solve_begin(x)
            s.solve_steps(warmup)
        except Exception as e:             # noqa: BLE001
            print("bench.py rank %d: warm-up failed: %s" % (rank, e), file=sys.stderr, flush=True)
            ok = False
        if not all_ok(ok):
            return None
        sync()
        t0 = time.perf_counter()
        try:
            s.solve_steps(steps)           # enqueues K loop bodies and synchronises the library's stream
        except Exception as e:             # noqa: BLE001
            print("bench.py rank %d: timed steps failed: %s" % (rank, e), file=sys.stderr, flush=True)
            ok = False
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        res = None
        try:
            res = s.solve_end(x)
        except Exception as e:             # noqa: BLE001
            print("bench.py rank %d: solve_end failed: %s" % (rank, e), file=sys.stderr, flush=True)
            ok = False
        if ok and dist is not None:
            # every rank must have reached bit-identical scalars (a stale or torn exchange would break this)
            mine = torch.tensor([res["iterations"], res["residual_prev"], res["x_norm"]], dtype=torch.float64, device=ctl)
            allv = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allv, mine)
            if not all(torch.equal(allv[0], v) for v in allv) or not math.isfinite(res["residual_prev"]):
                print("bench.py rank %d: ranks disagree on the result" % rank, file=sys.stderr, flush=True)
                ok = False
        elif dist is not None:
            dummy = torch.zeros(3, dtype=torch.float64, device=ctl)
            dist.all_gather([torch.zeros_like(dummy) for _ in range(world)], dummy)
        if not all_ok(ok):
            return None
        return elapsed, res

    def prewarm(s):
        """Untimed: about half a second of the same iteration, so that the timed region runs at settled clocks
        (the first ~0.2 s after start-up run 5-10 % slower; with 8 GPUs the whole default run is < 0.2 s).
        The step count is computed, not measured, so every rank does the same number of exchanges."""
        est = 8.0 * n * n / max(world, 1) / 6.5e12 + 25e-6
        iters = max(50, min(20000, int(0.5 / est)))
        return run(s, 0, iters) is not None

    def max_over_ranks(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    if not use_comm:
        order = ["self"]
    elif args.transport == "auto":
        order = ["p2p", "p2p-sep", "rccl"]
    else:
        order = [args.transport]
    # Build every candidate transport that works on this node; with more than one, a short calibration run
    # (same workload, 60 iterations) decides which one carries the timed run.  All decisions are taken on
    # rank-reduced values, so every rank takes the same branch.
    solvers = {}
    for tname in order:
        cand = make_solver(tname)
        if cand is not None:
            solvers[tname] = cand
    for tname, cand in list(solvers.items()):
        if not prewarm(cand):
            cand.close()
            del solvers[tname]
    calib = {}
    if len(solvers) > 1:
        for tname, cand in list(solvers.items()):
            c = run(cand, 30, 60)
            if c is None:
                cand.close()
                del solvers[tname]
            else:
                calib[tname] = max_over_ranks(c[0]) / 60 * 1e3
    out, transport, solver = None, None, None
    for tname in sorted(solvers, key=lambda t: calib.get(t, 0.0)):
        out = run(solvers[tname], args.warmup, args.steps)
        if out is not None:
            transport, solver = tname, solvers[tname]
            break
    if out is None:
        sys.exit("bench.py: no transport produced a result")
    for tname, cand in solvers.items():
        if cand is not solver:
            cand.close()
    elapsed, res = out

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g_ms = torch.tensor([res["gemv_ms_avg"]], dtype=torch.float64, device=ctl)
        dist.all_reduce(g_ms, op=dist.ReduceOp.MAX)
        gemv_ms = float(g_ms.item())
    else:
        gemv_ms = res["gemv_ms_avg"]

    if rank == 0:
        rows0 = pkg.partition(n, world)[1][0]
        gemv_bytes = 8.0 * (rows0 * n + n + rows0)            # A rows once + p + Ap
        ach = gemv_bytes / (gemv_ms * 1e-3) / 1e9 if gemv_ms > 0 else None
        line = {
            "metric": "cg_iterations_per_sec",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.mode,
            "vs_baseline": None,           # the reference publishes seconds only, nothing at this N
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "generate_lap2d_matrix N=%d, init_source_term(1/N), fixed-iteration dense fp64 CG "
                            "(BASELINE.json configs[%d])" % (n, 4 if args.mode == "weak" else (2 if world == 1 else 3)),
                "n": n, "rows_per_gpu": rows0, "parallelism": "rowblock%d" % world,
                "collectives": {"self": "none",
                                "rccl": "1 x ncclAllGather per iteration ([Ap slice | p.Ap partials])",
                                "p2p": "1 exchange per iteration over IPC/xGMI mailboxes, folded into K3 ([Ap slice | p.Ap])",
                                "p2p-sep": "1 mailbox all-gather kernel per iteration over IPC/xGMI ([Ap slice | p.Ap])"}[transport],
                "transport": transport,
                "transport_calibration_ms_per_iteration": calib or None,
                "k1_variant": args.variant,
            },
            "roofline": {
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (ach / HBM_PEAK_GBS) if ach else None,
                "traffic": pmc_traffic(n, world),
                "kernel": "k_gemv (K1, A.p of this rank's row block)", "bytes_per_launch": gemv_bytes,
                "avg_launch_ms": gemv_ms, "launches_timed": res["gemv_launches"],
            },
            "aggregate_gemv_GBs": (ach * world) if ach else None,
            "whole_iteration_GBs_per_gpu": 8.0 * (rows0 * n + n + 14 * rows0) / (elapsed / args.steps) / 1e9,
            "residual_after_run": res["residual_prev"],
            "iterations_done": res["iterations"],
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(n, args.cpu_baseline_iters)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    solver.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
