"""The streaming persistent kernel (csrc/cgx_stream.hip: the loop code/MPI/cg.cc:95-137 as ONE persistent kernel that streams
every row of A; 4096 < n <= 16384 on one GPU, the library's default up to n = 10000) against the oracle and against the
per-launch path.  All marked gpu.

gemv_variant 50000 asks for this kernel whatever the size (1024 <= n <= 16384; an expired wait is then an error); 0 is the
library's choice (with the environment's CGX_RESIDENT=0 of the rest of the suite removed).

Tolerances (fp64): fixed-iteration solves ||dx||/||x|| <= 1e-12 and residual rel. 1e-10 against the oracle (the same bars as
tests/test_gpu_parity.py); converged solves sqrt(rsnew) < tol, ||Ax-b||/||b|| <= 1e-11, k within 15 % of the oracle's.
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAM = 50000        # cgx_config.gemv_variant: ask for the streaming persistent kernel, fail if it cannot be had
LAUNCHES = -1         # the per-launch path with its default K1 shape


def rel(a, b):
    return abs(a - b) / abs(b)


def lap(pkg, n, variant, max_iter=None, tol=None, **kw):
    s = pkg.CGSolver(gemv_variant=variant, **kw)
    s.generate_lap2d_matrix(n)
    if max_iter is not None:
        s.set_max_iter(max_iter)
    if tol is not None:
        s.tolerance(tol)
    s.init_source_term(1.0 / n)
    return s


# one size per number of column steps S = ceil(n / 1024) the kernel is instantiated for, ragged sizes, and the sizes the
# round's brief names (5000, 8192, 10000, 12288); below 4097 only on request
SIZES = [1024, 1500, 2049, 3100, 4097, 5000, 5120, 6144, 6145, 7000, 8191, 8192, 8193, 9300, 10000, 10241, 11500, 12288, 13000, 14400, 15500, 16384]


@pytest.mark.parametrize("n", SIZES)
def test_fixed_iteration_solve_matches_oracle(gpu_pkg, oracle, n):
    iters = 40 if n <= 8192 else 24     # not yet converged: a residual at rounding level has no digits to compare
    with lap(gpu_pkg, n, STREAM, iters, 0.0) as s:
        plan = s.gemv_plan()
        x = np.zeros(n)
        r = s.solve(x)
        rec = s.resident_record()
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    assert plan["variant"] == 5 and plan["grid"] <= 256 and plan["R"] * plan["grid"] >= n and plan["U"] == (n + 1023) // 1024, plan
    assert r["iterations"] == ro["iterations"] == iters and not r["converged"]
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
    assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10
    assert rel(r["x_norm"], ro["x_norm"]) <= 1e-12
    assert rec["iterations"] == iters and rec["launches"] == 1 and rec["fallbacks"] == 0


@pytest.mark.parametrize("n", [1024, 4097, 5000])
def test_converged_solve(gpu_pkg, oracle, n):
    with lap(gpu_pkg, n, STREAM) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, None, 1e-10, 1)
    assert r["converged"] and ro["converged"]
    assert r["residual_last"] < 1e-10 <= r["residual_prev"]           # the break of cg.cc:120-121, taken at the first such k
    assert abs(r["iterations"] - ro["iterations"]) <= 0.15 * ro["iterations"] + 1
    assert r["rel_residual"] <= 1e-11
    assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)


def test_baseline_config_2_to_convergence(gpu_pkg, monkeypatch):
    """BASELINE.json configs[1]: generate_lap_2d_matrix N = 10000 on one GPU, run to convergence -- through the streaming kernel
    (the library's default at this size: per iteration the two are within 2 % of each other, the whole solve is 5 % shorter through it) and through the per-launch path: the reference's
    own k = 607 (tests/golden/reference_probe.json) within its run-to-run spread, the same x to rounding."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    n = 10000
    out = {}
    for name, v in (("default", 0), ("launches", LAUNCHES)):
        with lap(gpu_pkg, n, v) as s:
            out[name + "_plan"] = s.gemv_plan()["variant"]
            x = np.zeros(n)
            out[name] = (s.solve(x), x)
    assert out["default_plan"] == 5 and out["launches_plan"] == 1
    (ra, xa), (rb, xb) = out["default"], out["launches"]
    assert ra["converged"] and rb["converged"] and abs(ra["iterations"] - 607) <= 60 and abs(ra["iterations"] - rb["iterations"]) <= 30
    assert ra["rel_residual"] <= 1e-11 and rb["rel_residual"] <= 1e-11
    assert np.linalg.norm(xa - xb) <= 1e-9 * np.linalg.norm(xb)


def test_the_library_default(gpu_pkg, oracle, monkeypatch):
    """gemv_variant 0: the streaming kernel from n = 4097 to n = 10000 (where it measures faster than K1 + K3), the per-launch path
    above; CGX_STREAM_MAX moves that end; CGX_RESIDENT=0 and -1 keep the per-launch path."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    for n, want in ((4097, 5), (6000, 5), (8192, 5), (8193, 5), (9216, 5), (10000, 5), (10001, 1), (10240, 1), (12000, 1)):
        with lap(gpu_pkg, n, 0) as s:
            assert s.gemv_plan()["variant"] == want, n
    # the shape the plan reports: rows per workgroup, column steps of 1024, rows kept on the chip ("split"), one-row ring ("light")
    for n, rows, steps, chip in ((5120, 20, 5, 9), (8192, 32, 8, 4), (9216, 36, 9, 3)):
        with lap(gpu_pkg, n, 0) as s:
            pl = s.gemv_plan()
            assert (pl["R"], pl["U"], pl["split"], pl["light"], pl["grid"]) == (rows, steps, chip, 1, 256), pl
    monkeypatch.setenv("CGX_STREAM_MAX", "12288")
    with lap(gpu_pkg, 12000, 0) as s:
        assert s.gemv_plan()["variant"] == 5 and s.gemv_plan()["split"] == 0
    monkeypatch.setenv("CGX_STREAM_MAX", "4096")
    with lap(gpu_pkg, 6000, 0) as s:
        assert s.gemv_plan()["variant"] == 1
    monkeypatch.delenv("CGX_STREAM_MAX")
    with lap(gpu_pkg, 6000, LAUNCHES) as s:
        assert s.gemv_plan()["variant"] == 1
    monkeypatch.setenv("CGX_RESIDENT", "0")
    with lap(gpu_pkg, 6000, 0) as s:
        assert s.gemv_plan()["variant"] == 1
    monkeypatch.delenv("CGX_RESIDENT")
    with pytest.raises(gpu_pkg.CgxError):
        lap(gpu_pkg, 1000, STREAM)                 # below 1024 the streaming kernel is not built
    with pytest.raises(gpu_pkg.CgxError):
        lap(gpu_pkg, 16385, STREAM)
    # the default at n = 8192 against the oracle
    with lap(gpu_pkg, 8192, 0, 30, 0.0) as s:
        x = np.zeros(8192)
        s.solve(x)
    xo, _ = oracle.solve_lap2d(8192, 30, 0.0, 1)
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)


@pytest.mark.parametrize("n", [5000, 8192, 10000, 12288])
def test_fixed_iterations_in_pieces_are_bit_identical(gpu_pkg, n):
    """The loop cut into launches of any length gives the same bits as one launch: the state that crosses a launch boundary (x, r,
    p, rsold -- one state block read, the other written) is complete."""
    runs = []
    for pieces in ([60], [1] * 5 + [55], [30, 30], [7] * 9):
        with lap(gpu_pkg, n, STREAM, max_iter=60, tol=0.0) as s:
            s.solve_begin(np.zeros(n))
            for k in pieces:
                s.solve_steps(k)
            x = np.zeros(n)
            runs.append((s.solve_end(x), x))
    for r, x in runs[1:]:
        assert r["iterations"] == 60 and r["residual_prev"] == runs[0][0]["residual_prev"]
        assert np.array_equal(x, runs[0][1])


def test_break_semantics_and_resuming(gpu_pkg):
    """After the break nothing is updated any more (cg.cc:120-121), wherever the launches are cut."""
    n = 4500
    runs = []
    for pieces in ([600], [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 457], [37] * 17):
        with lap(gpu_pkg, n, STREAM, max_iter=600, tol=1e-10) as s:
            s.solve_begin(np.zeros(n))
            for k in pieces:
                s.solve_steps(k)
            x = np.zeros(n)
            runs.append((s.solve_end(x), x))
    r0, x0 = runs[0]
    assert r0["converged"] and r0["iterations"] < 600
    for r, x in runs[1:]:
        assert r["iterations"] == r0["iterations"] and r["converged"]
        assert r["residual_prev"] == r0["residual_prev"] and r["residual_last"] == r0["residual_last"]
        assert np.array_equal(x, x0)


@pytest.mark.parametrize("n,lda_pad", [(4500, -1), (5000, 0), (8192, -1), (8192, 0), (10000, 2), (12288, -1)])
def test_dense_hash_matrix_and_row_pitches(gpu_pkg, oracle, n, lda_pad):
    """Every element of the matrix a different number (the generator leaves five non-zeros per row): a lane that read the wrong
    column, a row that was swept twice or not at all would show.  lda_pad = 0: the pitch is roundup(n, 16) with no pad columns
    behind it, so the last column step of a row lands on real entries of the next row: they must not count."""
    seed, it = 777 + n, 12
    diag = 1.03 * 2.0 * (n / 3.0) ** 0.5
    with gpu_pkg.CGSolver(gemv_variant=STREAM, lda_pad=lda_pad) as s:
        s.generate_lap2d_matrix(n)
        s.probe_fill_matrix_hash(seed, symmetric=True, diag=diag)
        s.set_max_iter(it)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve(oracle.hash_rows(n, 0, n, seed, True, diag), oracle.init_source_term(n), max_iter=it, tol=0.0)
    assert r["iterations"] == ro["iterations"] == it
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
    assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_caller_matrix_and_initial_guess(gpu_pkg, oracle):
    rng = np.random.default_rng(23)
    n = 4200
    M = rng.standard_normal((n, 64))
    A = M @ M.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    with gpu_pkg.CGSolver(gemv_variant=STREAM) as s:
        s.set_matrix_dense(A)
        s.set_source_term(b)
        s.set_max_iter(25)
        s.tolerance(0.0)
        assert s.gemv_plan()["variant"] == 5
        x = x0.copy()
        r = s.solve(x)
    xo, ro = oracle.solve(A, b, x0, 25, 0.0, 1)
    # (x starts at x0, ||x0|| = 65, and ends near the solution, ||x|| = 0.015: its rounding errors are those of the larger of the two)
    assert np.linalg.norm(x - xo) <= 1e-12 * max(np.linalg.norm(xo), np.linalg.norm(x0)) and rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_matrix_market_config_1(gpu_pkg, oracle):
    """BASELINE.json configs[0]'s matrix (lap2D_5pt_n100.mtx, N = 10000) through the streaming kernel, 30 iterations against the
    oracle's reader + solve (the reference's CUDA benchmark runs this file: code/CUDA/cg.run)."""
    path = os.path.join(ROOT, "tests", "golden", "lap2D_5pt_n100.mtx")
    A = oracle.read_mtx_dense(path)[0]
    n = A.shape[0]
    with gpu_pkg.CGSolver(gemv_variant=STREAM) as s:
        s.read_matrix(path)
        assert s.gemv_plan()["variant"] == 5
        s.set_max_iter(30)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve(A, oracle.init_source_term(n), None, 30, 0.0, 1)
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo) and rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_context_reuse_across_sizes_and_kernels(gpu_pkg, monkeypatch):
    """One context through problems the resident kernel takes, the streaming kernel takes and the per-launch path takes, back and
    forth: every solve gives the bits of a fresh context (exchange buffer laid out anew per geometry, epochs only grow, the two
    state blocks rebound per problem)."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    sizes = (2048, 5000, 11500, 8192, 1024, 6144)
    fresh = {}
    for n in sizes:
        with lap(gpu_pkg, n, 0, 40, 0.0) as s:
            x = np.zeros(n)
            s.solve(x)
            fresh[n] = (x, s.gemv_plan()["variant"])
    assert [fresh[n][1] for n in sizes] == [4, 5, 1, 5, 4, 5]
    with gpu_pkg.CGSolver(gemv_variant=0) as s:
        for n in (5000, 2048, 8192, 11500, 5000, 1024, 6144, 8192, 2048):
            s.generate_lap2d_matrix(n)
            s.set_max_iter(40)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            assert s.gemv_plan()["variant"] == fresh[n][1]
            for _ in range(2):
                x = np.zeros(n)
                s.solve(x)
                assert np.array_equal(x, fresh[n][0]), n


def test_epoch_wrap_of_the_tag(gpu_pkg):
    """Solves across the wrap of the 32-bit tag give the bits of a fresh context (as tests/test_gpu_resident.py)."""
    n, iters = 5000, 40
    with lap(gpu_pkg, n, STREAM, iters, 0.0) as s:
        x0 = np.zeros(n)
        r0 = s.solve(x0)
    for start in (2**32 - 1 - 15, 2**32 - 15, 2**40):
        with lap(gpu_pkg, n, STREAM, iters, 0.0) as s:
            s.solve(np.zeros(n))                          # leaves tagged words of small epochs in the buffer
            s._resident_test(epoch=start)
            for _ in range(2):
                x = np.zeros(n)
                r = s.solve(x)
                assert r["iterations"] == iters and r["residual_prev"] == r0["residual_prev"], start
                assert np.array_equal(x, x0), start


@pytest.mark.parametrize("n", [5000, 8192])
def test_expired_waits(gpu_pkg, oracle, monkeypatch, n):
    """A workgroup that never publishes (test hook): on request (50000) the call returns an error within the bound and the context
    stays usable; under the default choice the launch is redone on the per-launch path and the call returns the oracle's result."""
    import time
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    iters = 30
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    with lap(gpu_pkg, n, STREAM, iters, 0.0, p2p_timeout_ms=200) as s:
        x_good = np.zeros(n)
        s.solve(x_good)
        s._resident_test(mute_workgroup=100)
        t0 = time.perf_counter()
        with pytest.raises(gpu_pkg.CgxError) as e:
            s.solve(np.zeros(n))
        assert "expired" in str(e.value) and time.perf_counter() - t0 < 5.0
        x = np.zeros(n)
        s.solve(x)
        assert np.array_equal(x, x_good)
    with lap(gpu_pkg, n, 0, iters, 0.0, p2p_timeout_ms=200) as s:
        assert s.gemv_plan()["variant"] == 5
        s.solve_begin(np.zeros(n))
        s.solve_steps(11)
        s._resident_test(mute_workgroup=3)
        s.solve_steps(iters)                              # falls back in the middle of the solve
        x = np.zeros(n)
        r = s.solve_end(x)
        assert s.gemv_plan()["variant"] == 1 and s.resident_record()["fallbacks"] == 1
    assert r["iterations"] == iters and np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
    assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_error_paths_of_a_streamed_solve(gpu_pkg):
    """Every HIP call of a whole solve through the streaming kernel is made to fail in turn (cgx_probe_set_fault_after): the call
    reports an error, the device's free memory is back where it was, and the same context then solves to the same bits."""
    import torch

    def free_mb():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] / 2**20

    n = 4500
    with lap(gpu_pkg, n, STREAM, 40, 0.0) as s:
        x_good = np.zeros(n)
        s.solve(x_good)
        base, failures = free_mb(), 0
        for k in range(200):
            s._set_fault_after(k)
            try:
                x = np.zeros(n)
                s.solve(x)
                s._set_fault_after(-1)
                break
            except gpu_pkg.CgxError as e:
                assert e.status in (3, 5), e
                failures += 1
            s._set_fault_after(-1)
            assert abs(free_mb() - base) < 2, k
        assert failures >= 5 and np.array_equal(x, x_good)      # (a solve from a zero initial guess is four launches and two synchronisations)
        x = np.zeros(n)
        s.solve(x)
        assert np.array_equal(x, x_good)


def test_cgsolver_cli(gpu_pkg, oracle, tmp_path):
    """`cgsolver 5000 out` (code/MPI/cg_main.cc:13-69) takes the streaming kernel by default and prints the reference's line; the
    same k as with CGX_RESIDENT=0 to the run-to-run spread, and --stats names the kernel and what its waits cost."""
    import re
    exe = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")
    outs = {}
    for name, val in (("stream", None), ("launches", "0")):
        env = dict(os.environ)
        env.pop("CGX_RESIDENT", None)
        if val is not None:
            env["CGX_RESIDENT"] = val
        p = subprocess.run([exe, "5000", str(tmp_path / (name + ".txt")), "--stats"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        outs[name] = p.stdout + p.stderr
    assert "loop=streaming-persistent-kernel" in outs["stream"] and "first_wait_us=" in outs["stream"]
    assert "loop=per-launch" in outs["launches"]
    _, ro = oracle.solve_lap2d(5000, None, 1e-10, 1)
    ks = {name: int(re.search(r"\[STEP (\d+)\]", text).group(1)) for name, text in outs.items()}
    assert abs(ks["stream"] - ro["iterations"]) <= 0.15 * ro["iterations"] + 1 and abs(ks["stream"] - ks["launches"]) <= 30


def test_where_a_workgroup_begins_its_sweep_changes_no_bit():
    """Every workgroup begins its sweep at a batch of its own (CGX_STREAM_STAGGER, default 1; 0 = all at their first rows): the rows
    are independent, so x must not differ in a single bit.  The switch is read once per process: two child processes."""
    code = ("import sys, hashlib, numpy as np; sys.path.insert(0, %r); import torch, __graft_entry__ as g; pkg = g.load_package()\n"
            "for n in (4500, 5120, 7000, 8192, 10000):\n"
            "    s = pkg.CGSolver(gemv_variant=50000); s.generate_lap2d_matrix(n); s.set_max_iter(60); s.tolerance(0.0)\n"
            "    s.init_source_term(1.0 / n); x = np.zeros(n); s.solve(x); s.close()\n"
            "    print(n, hashlib.sha256(x.tobytes()).hexdigest())\n" % ROOT)
    out = []
    for v in ("0", "1"):
        env = dict(os.environ, CGX_STREAM_STAGGER=v)
        env.pop("CGX_RESIDENT", None)
        p = subprocess.run(["python3", "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        out.append([l for l in p.stdout.splitlines() if l and l[0].isdigit()])
    assert len(out[0]) == 5 and out[0] == out[1], out
