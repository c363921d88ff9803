"""The direct peer exchange (CGX_COMM_P2P) with REAL processes: world_size ranks share
the one MI355X of the test box and talk through hipIpc-mapped mailboxes (RCCL cannot be rehearsed that way: it
refuses two ranks on one device).  Everything but the physical xGMI hop is the production path."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(world, n, max_iter, tmp_path, port, variant=0, separate=0, tagged=0, banded=0):
    out = tmp_path / ("p2p_%d_%d.json" % (world, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "p2p_worker.py"), str(n), str(max_iter), str(out), str(variant), str(separate), str(tagged), str(banded)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420,
                       env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return json.load(open(out))


@pytest.mark.parametrize("world,n,max_iter,port,separate", [
    (2, 2048, 200, 29701, 0), (4, 1000, 150, 29702, 0), (3, 1024, 4000, 29703, 0),   # exchange folded into K3 (default)
    (3, 1000, 150, 29704, 1), (2, 1024, 4000, 29705, 1),                             # exchange as its own kernel
    (3, 5, 3, 29708, 0), (4, 3, 2, 29709, 1),                                         # fewer rows than ranks: empty shards
])
def test_p2p_processes_on_one_gpu(tmp_path, world, n, max_iter, port, separate):
    v = run(world, n, max_iter, tmp_path, port, separate=separate)
    _check_p2p(v, n)


@pytest.mark.parametrize("world,n,max_iter,port,variant", [
    (2, 2048, 200, 29731, 0), (4, 1000, 150, 29732, 0), (3, 1024, 4000, 29733, 0), (3, 5, 3, 29734, 0),
    (3, 1000, 150, 29735, 10823), (4, 4096, 100, 29736, 10445), (2, 6000, 60, 29737, 0),   # column pieces; six chunks per rank
])
def test_p2p_tagged_words_processes_on_one_gpu(tmp_path, world, n, max_iter, port, variant):
    """The fused exchange with its bytes handed over as tagged 8-byte words (no flags, no fences: cgx_config.p2p_tagged) --
    the same chunks, the same arithmetic, the same bits as the flag form; real processes over IPC, incl. a second solve of
    another size on the same contexts, empty shards, and a K1 whose Ap arrives as column pieces."""
    v = run(world, n, max_iter, tmp_path, port, variant=variant, tagged=1)
    _check_p2p(v, n)


@pytest.mark.parametrize("world,n,max_iter,port,separate,variant", [
    (2, 2048, 200, 29711, 0, 10444), (3, 1000, 150, 29712, 0, 10823), (4, 4096, 100, 29713, 0, 10445),   # pushers and own rows add the pieces
    (3, 1000, 150, 29714, 1, 10444),                                                                  # separate exchange: combine kernel
    (3, 9, 3, 29715, 0, 10444),
])
def test_p2p_processes_with_a_column_split_k1(tmp_path, world, n, max_iter, port, separate, variant):
    """K1 with the columns of a row group split over several workgroups: the rank's Ap slice exists only as partial
    vectors, which the fused update adds up on the fly (pushers and own rows, same order) -- real processes over IPC."""
    v = run(world, n, max_iter, tmp_path, port, variant=variant, separate=separate)
    _check_p2p(v, n)


@pytest.mark.parametrize("tagged,port", [(0, 29741), (1, 29742)])
def test_p2p_more_chunk_pairs_than_threads(tmp_path, tagged, port):
    """200 000 rows on two ranks (banded storage: no dense block of that size exists): 196 chunks per rank, 392 (peer, chunk)
    pairs -- more than the 256 threads of a workgroup, so every thread of the fused update polls and folds more than one
    partial, and the pushers take several pairs each.  Both forms of the exchange, against the oracle's on-the-fly twin."""
    v = run(2, 200000, 30, tmp_path, port, tagged=tagged, banded=1)
    _check_p2p(v, 200000)


@pytest.mark.parametrize("world,n,iters,port", [(2, 3000, 12, 29751), (3, 1500, 12, 29752)])
def test_p2p_epoch_boundaries(tmp_path, world, n, iters, port):
    """The 64-bit epoch counter of the exchange moved (test hook) to just below 2^19, 0xFFF80000, 2^32 - 1, 2^32 and
    2 (2^32 - 1): a solve across each boundary, after a previous solve left its plain-double all-gathers in the mailbox, gives
    the bits of the first solve -- in both forms of the fused exchange, on every rank (real processes on one GPU)."""
    out = tmp_path / "epochs.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "p2p_epoch_worker.py"),
           str(n), str(iters), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420,
                       env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = json.load(open(out))
    assert v["ranks_agree"], v
    base = v["forms"]["flag"]["runs"][0]["digest"]
    for form in ("flag", "tagged"):
        f = v["forms"][form]
        assert f["selftest_ok"] and f["backwards_refused"], v
        assert len(f["runs"]) == 5
        assert f["runs"][3]["start"] < 2 ** 32 - 1 < 2 ** 32 < f["runs"][3]["end"]
        for q in f["runs"]:
            assert q["digest"] == base and q["k"] == iters, (form, q, base)
        # tagged words: the fused exchange alone counts on channel 1 (the plain all-gathers of begin / end have channel 0);
        # flag words: begin's and end's all-gather of segments count there as well
        for q in f["runs"][1:]:
            assert q["end"] == q["start"] + iters + (0 if form == "tagged" else 2), (form, q)


@pytest.mark.parametrize("world,n,iters,tagged,port", [(2, 3000, 60, 0, 29761), (2, 3000, 60, 1, 29762), (3, 2048, 80, 0, 29763), (3, 2048, 80, 1, 29764)])
def test_p2p_processes_over_shared_host_memory(tmp_path, world, n, iters, tagged, port):
    """The multi-process exchange with every rank's mailbox in POSIX shared HOST memory: all stores, polls and loads of all
    ranks cross PCIe between separate processes -- memory that is remote for every party, unlike the device mailboxes of a
    one-GPU rehearsal, which are this GPU's own HBM.  Self-test green, solves bit-identical to the device-mailbox run on every
    rank, both forms of the fused exchange; no segment is left behind in /dev/shm."""
    out = tmp_path / "hostmem.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "p2p_hostmem_worker.py"),
           str(n), str(iters), str(out), str(tagged)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420,
                       env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = json.load(open(out))
    assert v["selftest_device"] and v["selftest_host"], v
    assert v["host_equals_device_on_every_rank"] and v["ranks_agree"] and v["second_solve_same_bits"], v
    assert v["k"] == iters and v["dx_oracle"] < 1e-12, v
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("cgx_test_%d_" % port)]
    print("loop seconds: device mailboxes %.4f, shared host memory %.4f (world %d, tagged %d)" % (v["loop_s_device"], v["loop_s_host"], world, tagged))


def _check_p2p(v, n):
    assert v["selftest_ok"], v
    assert v["ranks_agree"], v
    assert v["k"] == v["k_oracle"] or (v["converged"] and abs(v["k"] - v["k_oracle"]) <= 0.15 * v["k_oracle"]), v
    assert v["dx"] < 1e-12, v
    if "second_dx" in v:                     # second solve of another size on the same contexts
        assert v["second_dx"] < 1e-12 and v["second_k"][0] == v["second_k"][1], v
    if not v["converged"] and n >= 16:       # on a 3x3 system the residual after 2 iterations is rounding noise
        assert v["residual_rel"] < 1e-6, v


@pytest.mark.parametrize("transport", ["p2p", "p2p-tag", "auto"])
def test_cgsolver_cli_forked_ranks_over_mailboxes(tmp_path, transport):
    """`cgsolver N OUT MAXITER --gpus 3`: the CLI forks one process per rank before touching the GPU and wires
    the mailboxes over pipes.  --same-device puts every rank on device 0 (one-GPU rehearsal).  auto = the flag form if its
    self-test passes on every rank, else RCCL; the tagged-word form runs on request (--transport p2p-tag)."""
    exe = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")
    out = tmp_path / "strong.txt"
    r = subprocess.run([exe, "2048", str(out), "200", "--gpus", "3", "--same-device", "--transport", transport, "--stats"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[STEP 200] residual = 1.331819e-05, ||x|| = 8.808702e+07" in r.stdout      # reference's own numbers
    assert r.stdout.count("[STEP") == 1                                                 # only rank 0 prints (cg.cc:144)
    assert out.read_text().strip().startswith("2048,3,")
    assert "gpus=3" in r.stderr and "unavailable" not in r.stderr


@pytest.mark.parametrize("tagged,port", [(0, 29706), (1, 29738)])
def test_p2p_config4_n32768_500_iterations_4_processes(tmp_path, tagged, port):
    """BASELINE.json configs[3] shape (N=32768, 500 iterations, row blocks) with 4 real processes exchanging over
    the mailboxes (both forms of the fused exchange), against the reference's recorded residual, ||x|| and sampled x."""
    v = run(4, 32768, 500, tmp_path, port, tagged=tagged)
    assert v["selftest_ok"] and v["ranks_agree"], v
    assert v["k"] == 500 and not v["converged"], v
    assert v["residual_rel"] < 1e-6 and v["x_norm_rel"] < 1e-12 and v["dx"] < 1e-12, v


@pytest.mark.parametrize("tagged,late,port", [(0, 0, 29707), (0, 1, 29739), (1, 1, 29740)])
def test_p2p_wait_is_bounded(tmp_path, tagged, late, port):
    """A peer that never answers must produce CGX_ERR_P2P after the timeout, not a hang: in the set-up phase (mailbox
    all-gather kernel), and -- `late` -- in the middle of the iteration loop, where the wait sits inside the fused update
    kernel (flag words, or tagged words polled by every thread)."""
    out = tmp_path / "stall.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "p2p_stall_worker.py"), str(out),
           str(tagged), str(late)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    v = json.load(open(out))
    assert v["raised"] and v["status"] == 8, v
    assert 0.3 < v["seconds"] < 2.5, v          # ~0.4 s timeout once; later exchanges return at once


def test_fused_update_refuses_a_grid_the_device_cannot_keep_resident(oracle):
    """The workgroups of the fused update wait for each other inside the kernel, so its grid must fit the device at once:
    the bound comes from the runtime (occupancy x CUs) when a problem is set; with the test hook it is forced below the
    grid and the problem is refused with CGX_ERR_UNSUPPORTED instead of risking a dead wait."""
    import numpy as np
    import torch  # noqa: F401 -- before libcgx
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    n = 2048                                       # 8 workgroups of 256 rows
    with pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1) as s:
        s._set_resident_limit(7)
        with pytest.raises(pkg.CgxError) as e:
            s.generate_lap2d_matrix(n)
        assert e.value.status == 7 and "must be resident at once" in str(e.value) and "8 workgroups" in str(e.value)
        s._set_resident_limit(8)
        s.generate_lap2d_matrix(n)                 # exactly fits
        s._set_resident_limit(0)                   # the runtime's own bound: thousands
        s.generate_lap2d_matrix(n + 2)
        s.set_max_iter(60)
        s.init_source_term(1.0 / (n + 2))
        x = np.zeros(n + 2)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n + 2, 60, 1e-10, 1)
    assert r["iterations"] == 60 and np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)


@pytest.mark.parametrize("tagged", [False, True])
def test_exchange_through_host_memory_gives_the_same_bits(oracle, tagged):
    """The mailbox in pinned, coherent HOST memory (test hook): every store of the fused exchange leaves the GPU over PCIe,
    every poll and load comes back over it -- memory that is neither this GPU's HBM nor behind its L2, at several times the
    latency.  The self-test must pass and the solve must give bit-identical results to the same solve over the device mailbox:
    the protocol does not rest on how local memory happens to behave.  Both forms of the exchange."""
    import numpy as np
    import torch  # noqa: F401 -- before libcgx
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    n, iters = 6000, 80                         # 12 chunks; 24 workgroups
    out = []
    for on_host in (False, True):
        with pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1, p2p_tagged=tagged, p2p_timeout_ms=20000) as s:
            if on_host:
                s._mailbox_to_host()
            assert s.p2p_selftest(8)
            s.generate_lap2d_matrix(n)
            s.set_max_iter(iters)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            x = np.zeros(n)
            r = s.solve(x)
            out.append((x, r))
    (xd, rd), (xh, rh) = out
    assert rd["iterations"] == rh["iterations"] == iters
    assert np.array_equal(xd, xh) and rd["residual_prev"] == rh["residual_prev"] and rd["x_norm"] == rh["x_norm"]
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    assert np.linalg.norm(xh - xo) <= 1e-12 * np.linalg.norm(xo)
    print("loop seconds: device mailbox %.4f, host mailbox %.4f" % (rd["seconds_loop"], rh["seconds_loop"]))


def test_both_forms_of_the_fused_exchange_give_the_same_bits():
    """Flag words and tagged words hand over the same chunks, reduced by the same device functions in the same order: the solves
    are bit-identical (one rank over its own mailbox; the multi-process tests compare each form with the oracle)."""
    import numpy as np
    import torch  # noqa: F401 -- before libcgx
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    pkg = g.load_package()
    n, iters = 5000, 60
    out = []
    for tagged in (False, True):
        with pkg.CGSolver(comm_mode=pkg.COMM_P2P, nranks=1, p2p_tagged=tagged) as s:
            s.generate_lap2d_matrix(n)
            s.set_max_iter(iters)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            x = np.zeros(n)
            out.append((x, s.solve(x)))
    (xf, rf), (xt, rt) = out
    assert np.array_equal(xf, xt) and rf["residual_prev"] == rt["residual_prev"] and rf["x_norm"] == rt["x_norm"]
