"""The resident solver (csrc/cgx_resident.hip: the loop code/MPI/cg.cc:95-137 as ONE persistent kernel; n <= 2048: the row groups of A
in the CUs' LDS, 2048 < n <= 4096: in LDS + registers + a streamed rest; one GPU) against the oracle and against the per-launch
path.  All marked gpu.

The rest of the GPU suite runs with CGX_RESIDENT=0 (tests/conftest.py), so that its small cases keep exercising K1 / K3;
here the resident kernel is asked for explicitly (gemv_variant 40000) or chosen by the library's default (environment
variable removed).

Tolerances (fp64): fixed-iteration solves ||dx||/||x|| <= 1e-12 and residual rel. 1e-10 against the oracle (the same bars as
tests/test_gpu_parity.py); converged solves sqrt(rsnew) < tol, ||Ax-b||/||b|| <= 1e-11, k within 15 % of the oracle's.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESIDENT = 40000      # cgx_config.gemv_variant: ask for the resident kernel, fail if it cannot be had
LAUNCHES = -1         # the per-launch path with its default K1 shape


def rel(a, b):
    return abs(a - b) / abs(b)


def lap(pkg, n, variant, max_iter=None, tol=None):
    s = pkg.CGSolver(gemv_variant=variant)
    s.generate_lap2d_matrix(n)
    if max_iter is not None:
        s.set_max_iter(max_iter)
    if tol is not None:
        s.tolerance(tol)
    s.init_source_term(1.0 / n)
    return s


SIZES = [3, 7, 64, 255, 256, 257, 511, 512, 513, 1000, 1024, 1025, 1448, 1536, 1537, 2047, 2048,
         # above 2048: 16 rows per workgroup in LDS + registers + (n > 3072) a rest streamed every iteration
         2049, 2560, 2561, 2896, 3072, 3073, 3500, 3584, 3585, 4000, 4095, 4096]


@pytest.mark.parametrize("n", SIZES)
def test_fixed_iteration_solve_matches_oracle(gpu_pkg, oracle, n):
    iters = max(1, min(n // 6, 40))     # not yet converged: a residual at rounding level has no digits to compare
    with lap(gpu_pkg, n, RESIDENT, iters, 0.0) as s:
        plan = s.gemv_plan()
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    assert plan["variant"] == 4 and plan["grid"] <= 256 and plan["R"] * plan["grid"] >= n, plan
    assert (plan["light"] > 0) == (n > 2048) and (plan["R"] == 16) == (n > 2048), plan
    assert r["iterations"] == ro["iterations"] == iters and not r["converged"]
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
    assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10
    assert rel(r["x_norm"], ro["x_norm"]) <= 1e-12


@pytest.mark.parametrize("n", [1, 7, 300, 1000, 1024, 2048, 2896, 4096])
def test_converged_solve(gpu_pkg, oracle, n):
    with lap(gpu_pkg, n, RESIDENT) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, None, 1e-10, 1)
    if n == 1:      # A = [4], b = [0]: the reference's own 0/0 (cg.cc:107 with rsold = 0); NaN on both sides, no break
        assert r["iterations"] == ro["iterations"] == 1 and np.isnan(x[0]) and np.isnan(xo[0])
        return
    assert r["converged"] and ro["converged"]
    assert r["residual_last"] < 1e-10 <= r["residual_prev"]           # the break of cg.cc:120-121, taken at the first such k
    assert abs(r["iterations"] - ro["iterations"]) <= 0.15 * ro["iterations"] + 1
    assert r["rel_residual"] <= 1e-11
    assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)


def test_same_results_as_the_per_launch_path(gpu_pkg):
    """Two different summation orders of the same recurrence: agreement to rounding, the same k on a converging run."""
    n = 1448
    out = {}
    for name, v in (("resident", RESIDENT), ("launches", LAUNCHES)):
        with lap(gpu_pkg, n, v) as s:
            assert (s.gemv_plan()["variant"] == 4) == (name == "resident")
            x = np.zeros(n)
            out[name] = (s.solve(x), x)
    (ra, xa), (rb, xb) = out["resident"], out["launches"]
    assert ra["converged"] and rb["converged"] and abs(ra["iterations"] - rb["iterations"]) <= 2
    assert np.linalg.norm(xa - xb) <= 1e-9 * np.linalg.norm(xb)


def test_break_semantics_and_resuming(gpu_pkg):
    """The loop cut into launches of any length gives the same bits as one launch: the state that crosses a launch boundary
    (x, r, p, rsold) is complete, and after the break nothing is updated any more (cg.cc:120-121)."""
    n = 1000
    runs = []
    for pieces in ([400], [1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 257], [7] * 60):
        with lap(gpu_pkg, n, RESIDENT, max_iter=400, tol=1e-10) as s:
            s.solve_begin(np.zeros(n))
            for k in pieces:
                s.solve_steps(k)
            x = np.zeros(n)
            runs.append((s.solve_end(x), x))
    r0, x0 = runs[0]
    assert r0["converged"] and r0["iterations"] < 400
    for r, x in runs[1:]:
        assert r["iterations"] == r0["iterations"] and r["converged"]
        assert r["residual_prev"] == r0["residual_prev"] and r["residual_last"] == r0["residual_last"]
        assert np.array_equal(x, x0)


@pytest.mark.parametrize("n", [2048, 2896, 3584, 4096])
def test_fixed_iterations_in_pieces_are_bit_identical(gpu_pkg, n):
    runs = []
    for pieces in ([120], [1] * 5 + [115], [60, 60]):
        with lap(gpu_pkg, n, RESIDENT, max_iter=120, tol=0.0) as s:
            s.solve_begin(np.zeros(n))
            for k in pieces:
                s.solve_steps(k)
            x = np.zeros(n)
            runs.append((s.solve_end(x), x))
    for r, x in runs[1:]:
        assert r["iterations"] == 120 and r["residual_prev"] == runs[0][0]["residual_prev"]
        assert np.array_equal(x, runs[0][1])


def test_max_iter_zero_and_one(gpu_pkg, oracle):
    n = 513
    for iters in (0, 1):
        with lap(gpu_pkg, n, RESIDENT, iters, 0.0) as s:
            x = np.zeros(n)
            r = s.solve(x)
        xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
        assert r["iterations"] == ro["iterations"] == iters
        assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-12
        assert np.linalg.norm(x - xo) <= 1e-13 * max(np.linalg.norm(xo), 1e-300)


@pytest.mark.parametrize("n", [777, 1536, 2048, 2500, 3000, 3600, 4096])
def test_dense_hash_matrix(gpu_pkg, oracle, n):
    """Every element of the matrix a different number (the generator leaves five non-zeros per row): a lane that read the
    wrong LDS word would show."""
    seed, it = 4242 + n, 30
    diag = 1.03 * 2.0 * (n / 3.0) ** 0.5
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
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


def test_caller_matrix_initial_guess_and_alpha_safeguard(gpu_pkg, oracle):
    rng = np.random.default_rng(11)
    n = 333
    M = rng.standard_normal((n, n))
    A = M @ M.T + n * np.eye(n)
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        s.set_matrix_dense(A)
        s.set_source_term(b)
        s.set_max_iter(40)
        s.tolerance(0.0)
        x = x0.copy()
        r = s.solve(x)
    xo, ro = oracle.solve(A, b, x0, 40, 0.0, 1)
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo) and rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10
    # A = -I: p.Ap < rsold * NEARZERO in every iteration, the second operand of std::max (cg.cc:107)
    n = 96
    A = -np.eye(n)
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        s.set_matrix_dense(A)
        s.set_source_term(b[:n].copy())
        s.set_max_iter(3)
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve(A, b[:n].copy(), None, 3, 1e-10, 1)
    assert r["iterations"] == ro["iterations"] == 3 and np.linalg.norm(x) > 1e40
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)


def test_context_reuse_across_sizes(gpu_pkg):
    """One context, problems of changing size and repeated solves: the exchange buffer is laid out anew (zero-filled) for every
    geometry and the epoch counter only grows, so no solve can read a tagged word of an earlier one."""
    fresh = {}
    for n in (2048, 1024, 600, 3000, 4096):
        with lap(gpu_pkg, n, RESIDENT, 60, 0.0) as s:
            x = np.zeros(n)
            s.solve(x)
            fresh[n] = x
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        for n in (2048, 1024, 4096, 2048, 600, 3000, 2048, 4096, 1024):
            s.generate_lap2d_matrix(n)
            s.set_max_iter(60)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            for _ in range(2):
                x = np.zeros(n)
                s.solve(x)
                assert np.array_equal(x, fresh[n]), n


def test_selection(gpu_pkg, monkeypatch):
    """Default = resident where it fits; an explicit per-launch shape, -1, CGX_RESIDENT=0, a larger matrix, several row blocks or
    banded storage keep the per-launch path; asking for it where it cannot be had is an error, not a silent fallback."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    with lap(gpu_pkg, 1024, 0) as s:
        assert s.gemv_plan()["variant"] == 4
    with lap(gpu_pkg, 4096, 0) as s:
        assert s.gemv_plan()["variant"] == 4
    with lap(gpu_pkg, 4097, 0) as s:               # above 4096 the matrix no longer fits the chip: the STREAMING persistent kernel
        assert s.gemv_plan()["variant"] == 5       # (csrc/cgx_stream.hip, tests/test_gpu_stream.py), by default up to n = 10000
    with lap(gpu_pkg, 10001, 0) as s:
        assert s.gemv_plan()["variant"] == 1
    with lap(gpu_pkg, 1024, 10421) as s:
        assert s.gemv_plan()["variant"] == 1
    with lap(gpu_pkg, 1024, LAUNCHES) as s:
        assert s.gemv_plan()["variant"] == 1
    monkeypatch.setenv("CGX_RESIDENT", "0")
    with lap(gpu_pkg, 1024, 0) as s:
        assert s.gemv_plan()["variant"] == 1
    with lap(gpu_pkg, 1024, RESIDENT) as s:       # the explicit request is not overridden by the environment
        assert s.gemv_plan()["variant"] == 4
    monkeypatch.delenv("CGX_RESIDENT")
    with lap(gpu_pkg, 4097, RESIDENT) as s:        # asked for: a persistent kernel up to n = 16384
        assert s.gemv_plan()["variant"] == 5
    with pytest.raises(gpu_pkg.CgxError) as e:
        lap(gpu_pkg, 16385, RESIDENT)
    assert "does not fit" in str(e.value)
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK, nranks=2, gemv_variant=0) as s:
        s.generate_lap2d_matrix(1024)
        assert s.gemv_plan()["variant"] == 1
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK, nranks=2, gemv_variant=RESIDENT) as s:
        with pytest.raises(gpu_pkg.CgxError):
            s.generate_lap2d_matrix(1024)
    with gpu_pkg.CGSolver(matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.generate_lap2d_matrix(1024)
        assert s.gemv_plan()["variant"] == 3


def test_grid_must_be_resident_at_once(gpu_pkg, monkeypatch):
    """The workgroups wait for each other inside the kernel: when the device would not keep all of them resident the default
    falls back to the per-launch path and the explicit request is refused."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    with gpu_pkg.CGSolver(gemv_variant=0) as s:
        s._set_resident_limit(100)
        s.generate_lap2d_matrix(1024)          # 256 workgroups
        assert s.gemv_plan()["variant"] == 1
        s.generate_lap2d_matrix(300)           # 150 workgroups of 2 rows
        assert s.gemv_plan()["variant"] == 1
        s.generate_lap2d_matrix(100)           # 100 workgroups
        assert s.gemv_plan()["variant"] == 4
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        s._set_resident_limit(100)
        with pytest.raises(gpu_pkg.CgxError) as e:
            s.generate_lap2d_matrix(1024)
        assert "resident at once" in str(e.value)


def test_cgsolver_cli_takes_the_resident_path_by_default(gpu_pkg, oracle, tmp_path):
    """`cgsolver 1024 out` (code/MPI/cg_main.cc:13-69): same lines as with CGX_RESIDENT=0, to rounding."""
    exe = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")
    outs = {}
    for name, val in (("resident", None), ("launches", "0")):
        env = dict(os.environ)
        env.pop("CGX_RESIDENT", None)
        if val is not None:
            env["CGX_RESIDENT"] = val
        out = tmp_path / (name + ".txt")
        p = subprocess.run([exe, "1024", str(out), "--stats"], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        outs[name] = p.stdout + p.stderr
    _, ro = oracle.solve_lap2d(1024, None, 1e-10, 1)
    import re
    ks = {}
    for name, text in outs.items():
        m = re.search(r"\[STEP (\d+)\]", text)
        assert m, text
        ks[name] = int(m.group(1))
    assert abs(ks["resident"] - ro["iterations"]) <= 0.15 * ro["iterations"] + 1
    assert abs(ks["resident"] - ks["launches"]) <= 2


@pytest.mark.parametrize("n", [1448, 3584])
def test_epoch_wrap_of_the_tag(gpu_pkg, n):
    """The 32-bit tag of the exchange is 1 + epoch mod (2^32 - 1): solves that run across the wrap of the tag, of the epoch's
    low 32 bits and of twice the period (cgx_probe_resident_test moves the counter there) give the bits of a fresh context.
    n = 1448: all rows in LDS; n = 3584: LDS + registers + streamed rows."""
    iters = 80
    with lap(gpu_pkg, n, RESIDENT, iters, 0.0) as s:
        x0 = np.zeros(n)
        r0 = s.solve(x0)
    for start in (2**32 - 1 - 30, 2**32 - 30, 2 * (2**32 - 1) - 41, 2**40):
        with lap(gpu_pkg, n, RESIDENT, iters, 0.0) as s:
            x = np.zeros(n)
            s.solve(x)                                   # leaves tagged words of small epochs in the buffer
            s._resident_test(epoch=start)
            for _ in range(2):                           # the first crosses the boundary, the second starts behind it
                x = np.zeros(n)
                r = s.solve(x)
                assert r["iterations"] == iters and r["residual_prev"] == r0["residual_prev"], start
                assert np.array_equal(x, x0), start
            with pytest.raises(gpu_pkg.CgxError):
                s._resident_test(epoch=start - 1)        # the counter only moves forward


@pytest.mark.parametrize("n", [1024, 4096])
def test_a_wait_that_expires_is_reported(gpu_pkg, n):
    """Every wait inside the kernel is bounded: a workgroup that never publishes (test hook) makes the waits for it expire after
    p2p_timeout_ms.  Asked for explicitly (gemv_variant 40000) the call returns an error instead of hanging -- and the context
    is NOT dead: the next solve starts with the error word down and gives the bits of a fresh context."""
    import time
    x_good = None
    for mute in (0, 37, 255):
        with gpu_pkg.CGSolver(gemv_variant=RESIDENT, p2p_timeout_ms=200) as s:
            s.generate_lap2d_matrix(n)
            s.set_max_iter(50)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            x_good = np.zeros(n)
            s.solve(x_good)
            s._resident_test(mute_workgroup=mute)
            t0 = time.perf_counter()
            with pytest.raises(gpu_pkg.CgxError) as e:
                s.solve(np.zeros(n))
            assert "expired" in str(e.value) and time.perf_counter() - t0 < 5.0
            assert s.resident_record()["fallbacks"] == 0
            x = np.zeros(n)
            s.solve(x)                                   # same context, same problem
            assert np.array_equal(x, x_good)
            s.generate_lap2d_matrix(n)                   # same context, the problem set again
            s.set_max_iter(50)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            x = np.zeros(n)
            s.solve(x)
            assert np.array_equal(x, x_good)
    with lap(gpu_pkg, n, RESIDENT, 50, 0.0) as s:
        x = np.zeros(n)
        s.solve(x)
    assert np.array_equal(x, x_good)


@pytest.mark.parametrize("n", [1024, 2048, 4096])
def test_the_default_choice_falls_back_to_the_per_launch_path(gpu_pkg, oracle, monkeypatch, n):
    """The same expired wait under the library's DEFAULT choice (gemv_variant 0): the launch has written nothing of the solver's
    state, the library redoes it on the per-launch path and the call returns CGX_OK with the oracle's result; the plan then
    reports the per-launch shape, the record counts the event, and the context goes on working -- the same problem on the
    per-launch path, the next problem on the persistent kernel again."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    import time
    iters = 50
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    with gpu_pkg.CGSolver(gemv_variant=0, p2p_timeout_ms=200) as s:
        s.generate_lap2d_matrix(n)
        s.set_max_iter(iters)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        assert s.gemv_plan()["variant"] == 4
        s._resident_test(mute_workgroup=37 % s.gemv_plan()["grid"])
        t0 = time.perf_counter()
        x = np.zeros(n)
        r = s.solve(x)
        assert time.perf_counter() - t0 < 5.0
        assert r["iterations"] == iters and np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
        assert rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10
        rec = s.resident_record()
        assert s.gemv_plan()["variant"] == 1 and rec["fallbacks"] == 1 and rec["persistent"] == 0
        x2 = np.zeros(n)
        s.solve(x2)                                      # the same problem again: stays on the per-launch path
        assert np.array_equal(x2, x) and s.resident_record()["fallbacks"] == 1
        s.generate_lap2d_matrix(n)                       # the next problem: the persistent kernel gets its chance again
        s.set_max_iter(iters)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        assert s.gemv_plan()["variant"] == 4
        x3 = np.zeros(n)
        s.solve(x3)
        assert np.linalg.norm(x3 - xo) <= 1e-12 * np.linalg.norm(xo) and s.resident_record()["persistent"] == 1


@pytest.mark.parametrize("k0", [1, 20, 21])
def test_fallback_in_the_middle_of_a_solve(gpu_pkg, oracle, monkeypatch, k0):
    """The wait expires in a LATER launch of a solve (the loop cut into pieces by the caller): the persistent kernels leave p
    already formed between launches, where the per-launch K1 forms it itself -- the iteration in between runs as the plain K1
    on the formed p followed by K3, and the solve goes on to the oracle's result."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    n, iters = 2048, 60
    xo, ro = oracle.solve_lap2d(n, iters, 0.0, 1)
    with gpu_pkg.CGSolver(gemv_variant=0, p2p_timeout_ms=200) as s:
        s.generate_lap2d_matrix(n)
        s.set_max_iter(iters)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)
        s.solve_begin(np.zeros(n))
        s.solve_steps(k0)
        assert s.gemv_plan()["variant"] == 4
        s._resident_test(mute_workgroup=5)
        s.solve_steps(7)                                  # falls back inside this call
        assert s.gemv_plan()["variant"] == 1
        s.solve_steps(iters)
        x = np.zeros(n)
        r = s.solve_end(x)
        assert s.resident_record()["fallbacks"] == 1
    assert r["iterations"] == iters
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo) and rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_the_record_of_the_waits(gpu_pkg):
    """cgx_get_resident_record: what the waits of the persistent launches cost comes back with {done, k_final} (no extra
    synchronisation) and is what `cgsolver --stats` prints."""
    with lap(gpu_pkg, 2048, RESIDENT, 300, 0.0) as s:
        s.solve(np.zeros(2048))
        rec = s.resident_record()
    assert rec["iterations"] == 300 and rec["launches"] == 1 and rec["persistent"] == 1 and rec["fallbacks"] == 0
    assert 0 < rec["wg0_first_wait_ticks"] <= rec["max_first_wait_ticks"] < 100000      # ticks of 10 ns: below 1 ms
    assert rec["wg0_longest_wait_ticks"] <= rec["max_longest_wait_ticks"] < 100000


_TENANT = r"""
import sys, time
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as g
pkg = g.load_package()
n, seconds = int(sys.argv[2]), float(sys.argv[3])
t0 = time.time()
done = errors = 0
with pkg.CGSolver(gemv_variant=40000, p2p_timeout_ms=500) as s:
    s.generate_lap2d_matrix(n); s.set_max_iter(20000); s.tolerance(0.0); s.init_source_term(1.0 / n)
    print("READY", flush=True)
    while time.time() - t0 < seconds:
        try:
            s.solve(np.zeros(n)); done += 1
        except pkg.CgxError:
            errors += 1
print("TENANT", done, errors)
"""


def test_another_tenant_of_the_gpu(gpu_pkg, oracle, monkeypatch):
    """A second process keeps LDS-heavy persistent grids on the GPU (the resident kernel itself at n = 2048, one workgroup per CU,
    20 000 iterations per launch, WITHOUT the advisory lock: CGX_RESIDENT_NOLOCK) while this process solves n = 2048 under the
    default choice with a short bound on the waits.  Every solve here ends with CGX_OK and the oracle's result within the bound,
    by either path (the record says how many went through the fallback)."""
    import time
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    monkeypatch.setenv("CGX_RESIDENT_NOLOCK", "1")
    n, iters = 2048, 100
    xo, _ = oracle.solve_lap2d(n, iters, 0.0, 1)
    env = dict(os.environ)
    env.pop("CGX_RESIDENT", None)
    tenant = subprocess.Popen([sys.executable, "-c", _TENANT, ROOT, str(n), "12"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env)
    try:
        assert tenant.stdout.readline().startswith("READY")
        fell, solves, worst = 0, 0, 0.0
        t_end = time.time() + 8.0
        while time.time() < t_end:
            with gpu_pkg.CGSolver(gemv_variant=0, p2p_timeout_ms=300) as s:
                s.generate_lap2d_matrix(n)
                s.set_max_iter(iters)
                s.tolerance(0.0)
                s.init_source_term(1.0 / n)
                for _ in range(5):
                    t0 = time.perf_counter()
                    x = np.zeros(n)
                    r = s.solve(x)
                    worst = max(worst, time.perf_counter() - t0)
                    assert r["iterations"] == iters and np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
                    solves += 1
                fell += s.resident_record()["fallbacks"]
        so, se = tenant.communicate(timeout=120)
    finally:
        if tenant.poll() is None:
            tenant.kill()
    assert tenant.returncode == 0, so + se
    assert solves >= 5 and worst < 5.0, (solves, worst)
    print("solves %d, of which through the fallback %d, slowest %.3f s; %s" % (solves, fell, worst, so.strip().splitlines()[-1]))


_WORKER = r"""
import sys, hashlib
import numpy as np, torch
sys.path.insert(0, sys.argv[1])
import __graft_entry__ as g
pkg = g.load_package()
n, reps = int(sys.argv[2]), int(sys.argv[3])
h = hashlib.sha256()
with pkg.CGSolver(gemv_variant=40000, p2p_timeout_ms=3000) as s:
    s.generate_lap2d_matrix(n); s.set_max_iter(150); s.tolerance(0.0); s.init_source_term(1.0 / n)
    for _ in range(reps):
        x = np.zeros(n); s.solve(x); h.update(x.tobytes())
print("DIGEST", h.hexdigest())
"""


def test_two_processes_and_two_threads_at_once(gpu_pkg):
    """A resident grid of n = 2048 takes the LDS of every CU; two of them dispatched at the same moment could each be given
    half of the CUs and wait for the rest for ever (until the bounded waits expire).  The per-device advisory lock lets one
    run at a time: concurrent solves from two processes, and from two threads of one process, all finish with the same bits."""
    import threading
    n, reps = 2048, 60
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, str(n), str(reps)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for _ in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    digests = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
        digests.append([l for l in so.splitlines() if l.startswith("DIGEST")][0])
    assert digests[0] == digests[1]

    results, errors = [None, None], []

    def work(i):
        try:
            with gpu_pkg.CGSolver(gemv_variant=RESIDENT, p2p_timeout_ms=3000) as s:
                s.generate_lap2d_matrix(n)
                s.set_max_iter(150)
                s.tolerance(0.0)
                s.init_source_term(1.0 / n)
                for _ in range(reps):
                    x = np.zeros(n)
                    s.solve(x)
                results[i] = x
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert np.array_equal(results[0], results[1])


@pytest.mark.parametrize("n", [1024, 3584])
def test_error_paths_of_a_resident_solve(gpu_pkg, n):
    """Every HIP call of setting a problem and of a whole solve through the resident kernel is made to fail in turn
    (cgx_probe_set_fault_after): the call reports an error, the device's free memory is back where it was, and the same
    context then solves to the same bits."""
    import torch

    def free_mb():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] / 2**20

    def problem(s):
        s.generate_lap2d_matrix(n)
        s.set_max_iter(60)
        s.tolerance(0.0)
        s.init_source_term(1.0 / n)

    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        problem(s)
        assert s.gemv_plan()["variant"] == 4
        x_good = np.zeros(n)
        s.solve(x_good)
        base, failures = free_mb(), 0
        for k in range(200):
            s._set_fault_after(k)
            try:
                x = np.zeros(n)
                s.solve(x)
                s._set_fault_after(-1)
                break                                        # k is past the last HIP call of a solve
            except gpu_pkg.CgxError as e:
                assert e.status in (3, 5), e
                failures += 1
            s._set_fault_after(-1)
            assert abs(free_mb() - base) < 2, k
        assert failures >= 5 and np.array_equal(x, x_good)      # (a solve from a zero initial guess is four launches and two synchronisations)
        x = np.zeros(n)
        s.solve(x)
        assert np.array_equal(x, x_good)
    # the same over the calls that set the problem (a new geometry every time: n - 16, then n again)
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
        failures = 0
        for k in range(200):
            s.generate_lap2d_matrix(n - 16)
            s._set_fault_after(k)
            try:
                problem(s)
                s._set_fault_after(-1)
                break
            except gpu_pkg.CgxError:
                failures += 1
            s._set_fault_after(-1)
        assert failures >= 5
        x = np.zeros(n)
        s.solve(x)
        assert np.array_equal(x, x_good)


@pytest.mark.parametrize("n,lda_pad", [(2896, 0), (2896, 2), (3584, 0), (4096, 0), (1448, 0), (3000, 6)])
def test_other_row_pitches(gpu_pkg, oracle, n, lda_pad):
    """cgx_config.lda_pad = 0: the pitch is roundup(n, 16) with no pad columns behind it, so a column step of the streamed /
    register / LDS rows that reaches behind the pitch lands on real entries of a dense matrix: they must not count."""
    seed, it = 99 + n, 20
    diag = 1.03 * 2.0 * (n / 3.0) ** 0.5
    with gpu_pkg.CGSolver(gemv_variant=RESIDENT, lda_pad=lda_pad) as s:
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


def test_matrix_market_input(gpu_pkg, oracle, tmp_path):
    """MatrixCOO::read + Matrix::read (matrix_coo.cc:7-60, matrix.cc:6-22) into a problem the resident kernel takes: a
    symmetric coordinate file of n = 300 and a general one of n = 2500, against the oracle's reader + solve."""
    rng = np.random.default_rng(8)
    for n, sym in ((300, True), (2500, False)):
        path = tmp_path / ("m%d.mtx" % n)
        entries = []
        for i in range(n):
            entries.append((i, i, 4.0 + rng.random()))
            for j in (i - 1, i - 17):
                if j >= 0:
                    v = -rng.random()
                    entries.append((i, j, v))
                    if not sym:
                        entries.append((j, i, v))
        with open(path, "w") as f:
            f.write("%%%%MatrixMarket matrix coordinate real %s\n%d %d %d\n" % ("symmetric" if sym else "general", n, n, len(entries)))
            for i, j, v in entries:
                f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
        A = oracle.read_mtx_dense(str(path))[0]
        b = oracle.init_source_term(n)
        with gpu_pkg.CGSolver(gemv_variant=RESIDENT) as s:
            s.read_matrix(str(path))
            assert s.gemv_plan()["variant"] == 4
            s.set_max_iter(30)
            s.tolerance(0.0)
            s.init_source_term(1.0 / n)
            x = np.zeros(n)
            r = s.solve(x)
        xo, ro = oracle.solve(A, b, None, 30, 0.0, 1)
        assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo) and rel(r["residual_prev"], ro["residual_prev"]) <= 1e-10


def test_context_life_cycles_leak_nothing(gpu_pkg):
    """200 contexts set a resident problem, solve and go away: the device's free memory and the process's open files (the lock
    file of the device is opened per context) are back where they were."""
    import torch

    def state():
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0] / 2**20, len(os.listdir("/proc/self/fd"))

    with lap(gpu_pkg, 1024, RESIDENT, 20, 0.0) as s:      # first use: the runtime's own one-time allocations
        s.solve(np.zeros(1024))
    mb0, fd0 = state()
    for i in range(200):
        n = (1024, 300, 2896)[i % 3]
        with lap(gpu_pkg, n, RESIDENT, 5, 0.0) as s:
            s.solve(np.zeros(n))
    mb1, fd1 = state()
    assert abs(mb1 - mb0) < 8 and fd1 == fd0, (mb0, mb1, fd0, fd1)


def test_one_call_and_three_calls_give_the_same(gpu_pkg, monkeypatch):
    """cgx_solve from a zero initial guess enqueues the verification GEMV and the end kernel behind the persistent launch and
    synchronises once; begin / steps / end does it in three steps with a synchronisation each: the same bits in x and in every
    field of the result, for the resident and the streaming kernel, converged and cut off, and with a non-zero initial guess (where
    neither path is the lean one)."""
    monkeypatch.delenv("CGX_RESIDENT", raising=False)
    for n, iters, tol in ((1024, 400, 1e-10), (1024, 60, 0.0), (3000, 80, 0.0), (5000, 50, 0.0), (5000, 600, 1e-10)):
        for x0 in (np.zeros(n), np.linspace(-1.0, 1.0, n)):
            with lap(gpu_pkg, n, 0, iters, tol) as s:
                assert s.gemv_plan()["variant"] in (4, 5)
                xa = x0.copy()
                ra = s.solve(xa)
                s.solve_begin(x0)
                s.solve_steps(iters)
                xb = np.zeros(n)
                rb = s.solve_end(xb)
                xc = x0.copy()
                rc = s.solve(xc)                         # and once more in one call, on the same context
                assert s.resident_record()["fallbacks"] == 0
            assert np.array_equal(xa, xb) and np.array_equal(xa, xc), (n, iters, tol)
            for key in ("iterations", "converged", "residual_prev", "residual_last", "x_norm", "rel_residual"):
                assert ra[key] == rb[key] == rc[key], (n, iters, tol, key)
