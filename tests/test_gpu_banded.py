"""Opt-in banded storage (cgx_config.matrix_format = CGX_MATRIX_BANDED) against the same
checkers as the dense path: the CPU oracle run on the DENSE matrix, the reference's recorded outputs, and, where an
n x n block cannot exist, the oracle's on-the-fly twin (validated against the dense oracle in test_oracle.py).

The banded path is not in the reference; what is claimed is that it returns what the reference's dense solve returns
on the same input, to the same tolerances as the dense HIP path (tests/test_gpu_parity.py header).
"""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    return abs(a - b) / abs(b)


def make(pkg, n=None, mode=None, nranks=1, max_iter=None, mtx=None, tol=None, dense=None, **kw):
    s = pkg.CGSolver(comm_mode=pkg.COMM_SELF if mode is None else mode, nranks=nranks,
                     matrix_format=pkg.MATRIX_BANDED, **kw)
    if mtx:
        s.read_matrix(mtx)
    elif dense is not None:
        s.set_matrix_dense(dense)
    else:
        s.generate_lap2d_matrix(n)
    if max_iter is not None:
        s.set_max_iter(max_iter)
    if tol is not None:
        s.tolerance(tol)
    s.init_source_term(1.0 / s.n())
    return s


# ---- storage: the diagonals hold exactly the generator's / the file's / the caller's matrix -----------------------
@pytest.mark.parametrize("n,mode,p", [(1, None, 1), (2, None, 1), (3, None, 1), (17, None, 1), (1000, None, 1),
                                      (1001, 1, 3), (4096, 1, 8), (5, 1, 8)])
def test_generator_bit_exact_and_offsets(gpu_pkg, oracle, n, mode, p):
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_SELF if mode is None else gpu_pkg.COMM_LOOPBACK, nranks=p,
                          matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.generate_lap2d_matrix(n)
        blocks = [s.probe_matrix_rows(i) for i in range(p)]
        fmt, offs, nbytes = s.matrix_format(0)
    inc = int(np.floor(np.sqrt(n)))
    assert fmt == gpu_pkg.MATRIX_BANDED
    assert offs == [o for o in (-(inc + 1), -1, 0, 1, inc + 1) if -n < o < n]
    assert nbytes <= 8 * 5 * (n // p + p + 2)
    assert np.array_equal(np.vstack([b[0] for b in blocks]), oracle.generate_lap2d(n))


def test_mtx_into_banded_storage_matches_oracle_dense(gpu_pkg, oracle, mtx_path):
    A, nz, sym = oracle.read_mtx_dense(mtx_path)
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK, nranks=3, matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.read_matrix(mtx_path)
        got = np.vstack([s.probe_matrix_rows(i)[0] for i in range(3)])
        offs = s.matrix_format(1)[1]
    assert offs == [-100, -1, 0, 1, 100]          # the true 5-point Laplacian of a 100 x 100 grid
    assert np.array_equal(got, A)


def test_dense_input_is_scanned_and_packed(gpu_pkg):
    rng = np.random.default_rng(5)
    n = 777
    A = np.zeros((n, n))
    for o in (-300, -7, -1, 0, 2, 5, 776):
        idx = np.arange(max(0, -o), min(n, n - o))
        A[idx, idx + o] = rng.standard_normal(idx.size)
    A[10, 10] = 0.0                                 # a zero inside a stored diagonal
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK, nranks=4, matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.set_matrix_dense(A)
        got = np.vstack([s.probe_matrix_rows(i)[0] for i in range(4)])
        all_offs = sorted(set(o for i in range(4) for o in s.matrix_format(i)[1]))
        p = rng.standard_normal(n)
        y, pap = s.probe_gemv(p)
    assert np.array_equal(got, A)
    assert all_offs == [-300, -7, -1, 0, 2, 5, 776]
    yo = A @ p
    assert np.max(np.abs(y - yo)) <= 2e-14 * np.max(np.abs(yo))
    assert abs(pap - p @ yo) <= 1e-12 * np.sum(np.abs(p * yo))


def test_too_many_diagonals_is_refused_loudly(gpu_pkg, tmp_path):
    """A matrix that is not banded must be refused, never silently densified or truncated."""
    rng = np.random.default_rng(1)
    A = rng.standard_normal((200, 200))
    with gpu_pkg.CGSolver(matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        with pytest.raises(gpu_pkg.CgxError) as e:
            s.set_matrix_dense(A)
        assert e.value.status == 7 and "diagonals" in str(e.value)
        with pytest.raises(gpu_pkg.CgxError):
            s.solve(np.zeros(200))                  # no matrix is set after the refusal
        f = tmp_path / "wide.mtx"
        rows = ["%d %d 1.0" % (1, j + 1) for j in range(100)]
        f.write_text("%%MatrixMarket matrix coordinate real general\n100 100 100\n" + "\n".join(rows) + "\n")
        with pytest.raises(gpu_pkg.CgxError) as e:
            s.read_matrix(str(f))
        assert e.value.status == 7
        s.generate_lap2d_matrix(64)                 # the context stays usable
        s.init_source_term(1.0 / 64)
        assert s.solve(np.zeros(64))["iterations"] > 0
    with pytest.raises(gpu_pkg.CgxError):
        gpu_pkg.CGSolver(matrix_format=7)


# ---- K1 on the diagonals ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,p", [(1, 1), (2, 1), (63, 1), (257, 1), (1000, 1), (2049, 1), (1000, 3), (2048, 8), (5, 8), (70000, 1), (600000, 2)])
def test_banded_matvec_generated(gpu_pkg, n, p):
    rng = np.random.default_rng(n)
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK if p > 1 else gpu_pkg.COMM_SELF, nranks=p,
                          matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.generate_lap2d_matrix(n)
        v = rng.standard_normal(n)
        y, pap = s.probe_gemv(v)
    inc = int(np.floor(np.sqrt(n)))
    yo = 4.0 * v                                     # cg.cc:181-185 applied to v
    yo[1:] -= v[:-1]                                 # i > 0          : -v[i-1]
    yo[:-1] -= v[1:]                                 # i < n-1        : -v[i+1]
    if n > inc + 1:
        yo[inc + 1:] -= v[:n - inc - 1]              # i > inc        : -v[i-1-inc]
        yo[:n - inc - 1] -= v[inc + 1:]              # i < n-1-inc    : -v[i+1+inc]
    assert np.max(np.abs(y - yo)) <= 1e-14 * max(np.max(np.abs(yo)), 1e-300)
    assert abs(pap - v @ yo) <= 1e-12 * np.sum(np.abs(v * yo))


K1B_DIRECT, K1B_WINDOWS = 30001, 30002      # cgx_config.gemv_variant: the two forms of K1b (default: by size)


@pytest.mark.parametrize("n,p", [(1, 1), (2, 1), (3, 1), (63, 1), (511, 1), (512, 1), (513, 1), (1000, 3), (2049, 1), (2048, 8), (5, 8),
                                 (70001, 1), (600000, 2), (1000003, 3)])
def test_banded_matvec_lds_windows_equal_the_direct_form_bit_for_bit(gpu_pkg, n, p):
    """The LDS-window form of K1b fetches each window of p once per 512-row tile; the arithmetic and its order are the
    direct form's, so Ap and p.Ap must be identical bits -- on odd sizes, odd row offsets of the shards (n=1000, P=3:
    rows start at 333), first and last tiles, and tiles behind the block."""
    rng = np.random.default_rng(n + p)
    v = rng.standard_normal(n)
    out = []
    for variant in (K1B_DIRECT, K1B_WINDOWS):
        with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK if p > 1 else gpu_pkg.COMM_SELF, nranks=p,
                              matrix_format=gpu_pkg.MATRIX_BANDED, gemv_variant=variant) as s:
            s.generate_lap2d_matrix(n)
            out.append(s.probe_gemv(v))
    assert np.array_equal(out[0][0], out[1][0]) and out[0][1] == out[1][1]


@pytest.mark.parametrize("offsets", [[0], [-1, 0, 1], [-700, -300, -299, 0, 1, 2, 300, 999], list(range(-20, 21)),
                                     [-2500 + 600 * i for i in range(9)]])
def test_banded_lds_windows_arbitrary_diagonals(gpu_pkg, oracle, offsets):
    """Caller matrices with other diagonal sets: one window, several windows with gaps, 41 adjacent diagonals in one
    window, and nine far-apart diagonals (more windows than the kernel takes: the direct form must run instead)."""
    n = 3001
    rng = np.random.default_rng(len(offsets))
    A = np.zeros((n, n))
    for o in offsets:
        idx = np.arange(max(0, -o), min(n, n - o))
        A[idx, idx + o] = rng.standard_normal(idx.size)
    v = rng.standard_normal(n)
    yo = oracle.gemv(A, v)
    for variant in (K1B_DIRECT, K1B_WINDOWS):
        with gpu_pkg.CGSolver(matrix_format=gpu_pkg.MATRIX_BANDED, gemv_variant=variant) as s:
            s.set_matrix_dense(A)
            assert s.matrix_format()[1] == sorted(offsets)
            y, pap = s.probe_gemv(v)
        assert np.max(np.abs(y - yo)) <= 1e-13 * np.max(np.abs(yo))
        assert abs(pap - oracle.dot(v, yo)) <= 1e-11 * np.sum(np.abs(v * yo))


@pytest.mark.parametrize("n,max_iter,mode,p", [(1000, 150, 1, 3), (2048, 200, None, 1), (70001, 120, None, 1), (300000, 100, 1, 2)])
def test_banded_solve_lds_windows_equal_the_direct_form(gpu_pkg, n, max_iter, mode, p):
    """Whole solves (fused K1b: head, p = r + beta p formed on the way into LDS, p_new stored) with both forms."""
    res = []
    for variant in (K1B_DIRECT, K1B_WINDOWS):
        with make(gpu_pkg, n, mode, p, max_iter, gemv_variant=variant) as s:
            x = np.zeros(n)
            r = s.solve(x)
            res.append((x, r))
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1]["residual_prev"] == res[1][1]["residual_prev"] and res[0][1]["iterations"] == res[1][1]["iterations"] == max_iter


def test_banded_matvec_equals_dense_matvec_on_the_same_matrix(gpu_pkg):
    n = 3000
    rng = np.random.default_rng(3)
    v = rng.standard_normal(n)
    out = []
    for fmt in (gpu_pkg.MATRIX_DENSE, gpu_pkg.MATRIX_BANDED):
        with gpu_pkg.CGSolver(matrix_format=fmt) as s:
            s.generate_lap2d_matrix(n)
            out.append(s.probe_gemv(v))
    assert np.max(np.abs(out[0][0] - out[1][0])) <= 4e-15 * np.max(np.abs(out[0][0]))
    assert rel(out[0][1], out[1][1]) < 1e-12


# ---- whole solves against the dense oracle ------------------------------------------------------------------------------
@pytest.mark.parametrize("n,max_iter,mode,p", [
    (64, 10, None, 1), (1000, 100, None, 1), (2048, 200, None, 1), (2048, 200, 1, 2), (2048, 200, 1, 8), (1000, 150, 1, 3),
    (1000, 150, 1, 7), (4096, 200, 1, 8),
])
def test_fixed_iteration_solve_matches_dense_oracle(gpu_pkg, oracle, n, max_iter, mode, p):
    with make(gpu_pkg, n, mode, p, max_iter) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, max_iter, 1e-10, p)
    assert r["iterations"] == ro["iterations"] == max_iter and not r["converged"]
    assert rel(r["residual_prev"], ro["residual_prev"]) < 1e-6
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-12
    assert rel(r["x_norm"], ro["x_norm"]) < 1e-12
    tol = 1e-5 if ro["rel_residual"] > 1e-9 else 1e-2
    assert rel(r["rel_residual"], ro["rel_residual"]) < tol


@pytest.mark.parametrize("n,max_iter,p", [(3, 2, 4), (5, 3, 8), (7, 4, 7), (2, 1, 3)])
def test_fewer_rows_than_ranks(gpu_pkg, oracle, n, max_iter, p):
    with make(gpu_pkg, n, gpu_pkg.COMM_LOOPBACK, p, max_iter, tol=0.0) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, max_iter, 0.0, p)
    assert r["iterations"] == ro["iterations"] == max_iter
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)


@pytest.mark.parametrize("n,mode,p", [(1024, None, 1), (1024, 1, 4), (1000, 1, 3), (4096, None, 1)])
def test_converged_solve(gpu_pkg, oracle, reference_probe, n, mode, p):
    with make(gpu_pkg, n, mode, p) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, None, 1e-10, p)
    assert r["converged"] and r["residual_last"] < 1e-10 <= r["residual_prev"]
    assert r["rel_residual"] <= 1e-11
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-12
    assert abs(r["iterations"] - ro["iterations"]) <= 0.15 * ro["iterations"]
    for q in [q for q in reference_probe["generated"] if q["n"] == n and q["max_iter"] is None]:
        assert abs(r["iterations"] - q["k"]) <= 0.15 * q["k"]
        assert rel(r["x_norm"], q["x_norm"]) < 1e-6


def test_initial_guess_and_caller_matrix(gpu_pkg, oracle):
    """set_matrix_dense + a non-zero x0 + user b, three row blocks: everything the dense path's test covers."""
    n = 600
    rng = np.random.default_rng(11)
    A = oracle.generate_lap2d(n) + np.diag(rng.uniform(0.0, 1.0, n))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_LOOPBACK, nranks=3, matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.set_matrix_dense(A)
        s.set_source_term(b)
        s.set_max_iter(40)
        x = x0.copy()
        r = s.solve(x)
    xo, ro = oracle.solve(A, b, x0, 40, 1e-10, 3)
    assert r["iterations"] == ro["iterations"]
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-12
    assert rel(r["residual_prev"], ro["residual_prev"]) < 1e-6


# ---- the reference's own recorded outputs -------------------------------------------------------------------------------
def test_mtx_solve_matches_reference_golden(gpu_pkg, mtx_path, reference_probe):
    row = reference_probe["mtx_lap2D_5pt_n100"][0]
    with make(gpu_pkg, mtx=mtx_path) as s:
        x = np.zeros(10000)
        r = s.solve(x)
    assert r["converged"] and r["residual_last"] < 1e-10
    assert abs(r["iterations"] - row["k"]) <= 0.15 * row["k"]
    assert rel(r["x_norm"], row["x_norm"]) < 1e-6 and r["rel_residual"] <= 1e-11
    for i, v in row["x_samples"].items():
        assert rel(x[int(i)], v) < 1e-10, i


@pytest.mark.parametrize("mode,p", [(None, 1), (1, 8)])
def test_config3_n32768_500_iterations_matches_reference(gpu_pkg, reference_probe, mode, p):
    row = [q for q in reference_probe["generated_large"] if q["n"] == 32768][0]
    with make(gpu_pkg, 32768, mode, p, max_iter=500) as s:
        x = np.zeros(32768)
        r = s.solve(x)
        fmt, offs, nbytes = s.matrix_format(0)
    assert offs == [-182, -1, 0, 1, 182] and nbytes < 2.0e6      # 8 GiB as a dense block
    assert r["iterations"] == row["k"]
    assert rel(r["residual_prev"], row["residual"]) < 1e-6
    assert rel(r["x_norm"], row["x_norm"]) < 1e-12
    assert rel(r["rel_residual"], row["rel_residual"]) < 1e-5
    for i, v in row["x_samples"].items():
        assert rel(x[int(i)], v) < 1e-12, i


def test_config5_n46340_matches_reference(gpu_pkg, reference_probe):
    row = [q for q in reference_probe["generated_large"] if q["n"] == 46340][0]
    with make(gpu_pkg, 46340, gpu_pkg.COMM_LOOPBACK, 8, max_iter=200) as s:
        x = np.zeros(46340)
        r = s.solve(x)
    assert r["iterations"] == row["k"] and rel(r["residual_prev"], row["residual"]) < 1e-6
    assert rel(r["x_norm"], row["x_norm"]) < 1e-12


# ---- sizes where no dense block can exist: the oracle's on-the-fly twin --------------------------------------------------
@pytest.mark.parametrize("n,max_iter,mode,p", [(300000, 300, None, 1), (1000003, 200, None, 1), (1000003, 100, 1, 3), (4194304, 60, None, 1)])
def test_large_n_matches_banded_oracle(gpu_pkg, oracle, n, max_iter, mode, p):
    """n > 262144 also exercises the strided update kernel and the capped partial counts."""
    with make(gpu_pkg, n, mode, p, max_iter) as s:
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d_banded(n, max_iter, 1e-10, p)
    assert r["iterations"] == ro["iterations"] == max_iter
    assert rel(r["residual_prev"], ro["residual_prev"]) < 1e-6
    assert np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-11
    assert rel(r["x_norm"], ro["x_norm"]) < 1e-11


def test_large_n_p2p_is_refused(gpu_pkg):
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_P2P, nranks=1, matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        with pytest.raises(gpu_pkg.CgxError) as e:
            s.generate_lap2d_matrix(300000)
        assert e.value.status == 7
        s.generate_lap2d_matrix(30000)
        s.init_source_term(1.0 / 30000)
        s.set_max_iter(50)
        assert s.solve(np.zeros(30000))["iterations"] == 50


def test_rccl_one_rank_banded(gpu_pkg, oracle):
    """The RCCL transport carries the same segments in banded mode (one rank: the collective plumbing only)."""
    n = 5000
    with gpu_pkg.CGSolver(comm_mode=gpu_pkg.COMM_RCCL, nranks=1, rank=0, unique_id=gpu_pkg.comm_unique_id(),
                          matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.generate_lap2d_matrix(n)
        s.init_source_term(1.0 / n)
        s.set_max_iter(120)
        x = np.zeros(n)
        r = s.solve(x)
    xo, ro = oracle.solve_lap2d(n, 120, 1e-10, 1)
    assert r["iterations"] == 120 and np.linalg.norm(x - xo) / np.linalg.norm(xo) < 1e-12


def test_dense_and_banded_agree_and_banded_is_the_lighter_one(gpu_pkg):
    n = 8192
    res = {}
    for fmt in (gpu_pkg.MATRIX_DENSE, gpu_pkg.MATRIX_BANDED):
        with gpu_pkg.CGSolver(matrix_format=fmt) as s:
            s.generate_lap2d_matrix(n)
            s.init_source_term(1.0 / n)
            x = np.zeros(n)
            r = s.solve(x)
            res[fmt] = (x, r, s.matrix_format(0)[2])
    xd, rd, bd = res[gpu_pkg.MATRIX_DENSE]
    xb, rb, bb = res[gpu_pkg.MATRIX_BANDED]
    assert rd["converged"] and rb["converged"] and abs(rd["iterations"] - rb["iterations"]) <= 2
    assert np.linalg.norm(xd - xb) / np.linalg.norm(xd) < 1e-12
    assert bb * 1000 < bd


def test_cgsolver_cli_banded(gpu_pkg, mtx_path, tmp_path):
    exe = os.path.join(ROOT, "conjugate-gradient_amd", "cgsolver")
    out = tmp_path / "out.txt"
    r = subprocess.run([exe, "10000", str(out), "--banded"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    step = [ln for ln in r.stdout.splitlines() if "[STEP" in ln][0]
    k = int(step.split("[STEP")[1].split("]")[0])
    assert abs(k - 607) <= 61                                      # the reference's 607 +- 10 %
    r = subprocess.run([exe, mtx_path, str(out), "--banded", "--loopback", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    k = int([ln for ln in r.stdout.splitlines() if "[STEP" in ln][0].split("[STEP")[1].split("]")[0])
    assert abs(k - 488) <= 49
    lines = out.read_text().strip().splitlines()
    assert lines[0].startswith("10000,1,") and lines[1].startswith("10000,2,")
