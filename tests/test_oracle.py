"""CPU tests of the parity oracle (oracle/cg_oracle.c) against the reference outputs recorded by the
survey (tests/golden/reference_probe.json) and against structural truths of the reference's code.
These pin the oracle; the GPU tests then compare libcgx with the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

REF = "/root/reference/code/MPI"


def rel(a, b):
    return abs(a - b) / abs(b)


# ---- partition_matrix, code/MPI/cg.cc:236-268 ---------------------------------------------------
@pytest.mark.parametrize("n,p,start,num", [
    (10, 1, [0], [10]),
    (10, 2, [0, 5], [5, 5]),
    (10, 3, [0, 3, 6], [3, 3, 4]),
    (1000, 3, [0, 333, 666], [333, 333, 334]),
    (7, 8, [0] * 8, [0] * 7 + [7]),
    (46340, 8, [5792 * i for i in range(8)], [5792] * 7 + [5796]),
    (32768, 8, [4096 * i for i in range(8)], [4096] * 8),
])
def test_partition_truth_table(oracle, n, p, start, num):
    s, c = oracle.partition(n, p)
    assert s == start and c == num
    assert sum(c) == n


# ---- generate_lap2d_matrix, code/MPI/cg.cc:159-188 ---------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9, 10, 16, 17, 100, 1000])
def test_generator_structure(oracle, n):
    A = oracle.generate_lap2d(n)
    inc = int(np.floor(np.sqrt(n)))
    E = np.zeros((n, n))
    for i in range(n):   # literal restatement of the five assignments
        if i > inc: E[i, i - 1 - inc] = -1
        if i > 0: E[i, i - 1] = -1
        E[i, i] = 4
        if i < n - 1: E[i, i + 1] = -1
        if i < n - 1 - inc: E[i, i + 1 + inc] = -1
    assert np.array_equal(A, E)
    assert np.array_equal(A, A.T)                     # symmetric penta-diagonal Toeplitz
    if n >= 4:
        assert np.count_nonzero(A[n // 2]) <= 5


def test_generator_row_blocks_match_full(oracle):
    n = 1000
    A = oracle.generate_lap2d(n)
    s, c = oracle.partition(n, 3)
    for r0, nr in zip(s, c):
        assert np.array_equal(oracle.generate_lap2d(n, r0, nr), A[r0:r0 + nr])


# ---- init_source_term, code/MPI/cg.cc:218-234 ---------------------------------------------------------
def test_source_term_expression(oracle):
    n = 1024
    b = oracle.init_source_term(n)
    h = 1.0 / n
    import math
    for i in (0, 1, 7, 511, 1023):
        e = -2. * i * math.pi * math.pi * math.sin(10. * math.pi * i * h) * math.sin(10. * math.pi * i * h)
        assert b[i] == e
    # ||b|| recorded by the survey for the reference
    assert rel(np.linalg.norm(b), 2.2847e5) < 1e-4
    assert rel(np.linalg.norm(oracle.init_source_term(10000)), 6.9722e6) < 1e-4


# ---- BLAS restatements vs numpy ---------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(1, 1), (3, 5), (4, 4), (7, 129), (64, 1000), (333, 1001)])
def test_gemv_dot_against_numpy(oracle, m, n):
    rng = np.random.default_rng(m * 1000 + n)
    A = rng.standard_normal((m, n))
    x = rng.standard_normal(n)
    y = oracle.gemv(A, x)
    assert np.allclose(y, A @ x, rtol=1e-13, atol=1e-13)
    assert rel(oracle.dot(x, x), float(x @ x)) < 1e-14


# ---- solve vs the reference's own outputs -------------------------------------------------------------
def _fixed_rows(probe):
    return [r for r in probe["generated"] if r["max_iter"] is not None and r["n"] <= 4096]


def test_fixed_iteration_runs_match_reference(oracle, reference_probe):
    """At a fixed iteration count the reference's printed numbers are reproducible to ~7 digits whatever
    the BLAS / rank count: the oracle must land on them."""
    seen = 0
    for row in _fixed_rows(reference_probe):
        x, r = oracle.solve_lap2d(row["n"], row["max_iter"], 1e-10, row["ranks"])
        assert r["iterations"] == row["k"]
        assert rel(r["residual_prev"], row["residual"]) < 1e-6, row
        assert rel(r["x_norm"], row["x_norm"]) < 1e-6, row          # printed with 7 digits
        # ||Ax-b||/||b|| near its rounding floor (~1e-11) moves with the summation order: 2e-4 observed
        # between MKL and this oracle at n=2048; away from the floor it is reproducible to 6 digits.
        tol = 1e-5 if row["rel_residual"] > 1e-9 else 1e-2
        assert rel(r["rel_residual"], row["rel_residual"]) < tol, row
        seen += 1
    assert seen >= 8


@pytest.mark.parametrize("n,ranks", [(1024, 1), (1024, 2), (1024, 8), (1000, 3), (2048, 1)])
def test_converged_runs_match_reference(oracle, reference_probe, n, ranks):
    row = [r for r in reference_probe["generated"] if r["n"] == n and r["ranks"] == ranks and r["max_iter"] is None][0]
    x, r = oracle.solve_lap2d(n, None, 1e-10, ranks)
    assert r["converged"] == 1
    assert abs(r["iterations"] - row["k"]) <= 0.15 * row["k"]        # k is not a stable observable
    assert r["residual_last"] < 1e-10
    assert rel(r["x_norm"], row["x_norm"]) < 1e-6
    assert r["rel_residual"] < 1e-11
    # break happens before rsold is updated: the printed residual is the previous iteration's (cg.cc:120,152)
    assert r["residual_prev"] >= 1e-10


def test_rank_count_does_not_change_solution(oracle):
    xs = [oracle.solve_lap2d(2048, 200, 1e-10, p)[0] for p in (1, 2, 3, 8)]
    for x in xs[1:]:
        assert np.linalg.norm(x - xs[0]) / np.linalg.norm(xs[0]) < 5e-15   # reference: <= 9e-16


def test_large_oracle_runs_pinned_to_reference(reference_probe, oracle_large):
    """tests/golden/oracle_large.json (oracle at the BASELINE.json sizes, made by make_oracle_large.py)
    against the reference's recorded outputs incl. sampled solution entries at full precision."""
    ref = {(r["n"], r["max_iter"]): r for r in reference_probe["generated_large"]}
    checked = 0
    for c in oracle_large["cases"]:
        r = ref.get((c["n"], c["max_iter"]))
        if r is None:
            continue
        assert c["k"] == r["k"]
        assert rel(c["residual"], r["residual"]) < 1e-6
        assert rel(c["x_norm"], r["x_norm"]) < 1e-12
        assert rel(c["rel_residual"], r["rel_residual"]) < 1e-5
        for i, v in r["x_samples"].items():
            assert rel(c["x_samples"][i], v) < 1e-12, (c["n"], i)
        checked += 1
    assert checked >= 2


def test_weak_scaling_series_is_fully_pinned(reference_probe, oracle_large):
    """Every point of BASELINE.json configs[4] -- (16384,1), (23170,2), (32768,4), (46340,8) at 200 iterations -- has a
    committed row.  Three are the reference's own outputs; (32768, P=4, 200) was not captured by the survey and is the
    oracle's (on-the-fly twin, itself pinned to the reference at 32768/500 and 46340/200 above)."""
    ref = {r["n"] for r in reference_probe["generated_large"] if r["max_iter"] == 200}
    ours = {(c["n"], c["psize"]): c for c in oracle_large["cases"] if c["max_iter"] == 200}
    assert ref == {16384, 23170, 46340}
    for n, p in ((16384, 1), (23170, 2), (32768, 4), (46340, 8)):
        assert (n, p) in ours, (n, p)
    c = ours[(32768, 4)]
    assert c["k"] == 200 and len(c["x_samples"]) >= 16 and c["residual"] > 1.0
    # the partition does not change the recurrence beyond rounding: the P=4 row and a fresh P=1 run of the twin agree
    # (reference: 1.331819e-05 for 1/2/4/8 ranks at N=2048)


def test_weak_scaling_p4_row_reproduces(oracle, oracle_large):
    """The committed (32768, P=4, 200 iterations) row is what the oracle produces today (the twin takes seconds)."""
    c = [q for q in oracle_large["cases"] if q["n"] == 32768 and q["max_iter"] == 200 and q["psize"] == 4][0]
    x, r = oracle.solve_lap2d_banded(32768, 200, 1e-10, 4)
    assert r["iterations"] == c["k"] and r["residual_prev"] == c["residual"] and r["x_norm"] == c["x_norm"]
    for i, v in c["x_samples"].items():
        assert x[int(i)] == v, i


def test_mtx_oracle_run_to_convergence_is_pinned(reference_probe, oracle_large):
    """BASELINE.json configs[0]: the committed oracle run of lap2D_5pt_n100.mtx to convergence (make_oracle_large.py mtx)
    against the reference's recorded run: k is not a stable observable (reference: 488 on one rank, 462 on four), x is."""
    row = oracle_large["mtx"][0]
    ref = reference_probe["mtx_lap2D_5pt_n100"][0]
    assert row["converged"] and row["residual_last"] < 1e-10 <= row["residual"]
    assert abs(row["k"] - ref["k"]) <= 0.15 * ref["k"]
    assert rel(row["x_norm"], ref["x_norm"]) < 1e-6 and row["rel_residual"] <= 1e-11
    assert len(row["x_samples"]) >= 16
    for i, v in ref["x_samples"].items():
        assert rel(row["x_samples"][i], v) < 1e-12, i


def test_config2_oracle_run_to_convergence_is_pinned(reference_probe, oracle_large):
    """BASELINE.json configs[1]: generate_lap2d N = 10000 to convergence (make_oracle_large.py c10000:1, the dense
    restatement; cg_main.cc:31-55) against the reference's recorded run: the same k = 607 here, the printed 7 digits of
    ||x|| and ||Ax-b||/||b||; the last recurrence residual itself sits at rounding level (1.01e-10 vs 1.02e-10)."""
    row = [q for q in oracle_large["converged"] if q["n"] == 10000 and q["psize"] == 1][0]
    ref = [q for q in reference_probe["generated"] if q["n"] == 10000][0]
    assert row["converged"] and row["residual_last"] < 1e-10 <= row["residual"]
    assert abs(row["k"] - ref["k"]) <= 0.10 * ref["k"]
    assert rel(row["x_norm"], ref["x_norm"]) < 1e-6 and rel(row["rel_residual"], ref["rel_residual"]) < 1e-2
    assert rel(row["residual"], ref["residual"]) < 0.05
    assert len(row["x_samples"]) >= 20


# ---- Matrix-Market input surface -----------------------------------------------------------------------
def test_mtx_fixture_is_the_reference_file(mtx_path):
    ref = os.path.join(REF, "lap2D_5pt_n100.mtx")
    if not os.path.exists(ref):
        pytest.skip("reference tree not present on this machine")
    assert open(mtx_path, "rb").read() == open(ref, "rb").read()


def test_mtx_reader(oracle, mtx_path):
    A, nz, sym = oracle.read_mtx_dense(mtx_path)
    assert A.shape == (10000, 10000) and nz == 29800 and sym
    assert np.count_nonzero(A) == 49600                       # after mirroring
    assert np.array_equal(A, A.T)
    assert np.all(np.diag(A) == 4.0)
    rs = A.sum(axis=1)
    assert rs.min() == 0.0 and rs.max() == 2.0                # interior rows sum to 0, corners to 2
    # true 5-point stencil: the +-1 band is broken at grid-row boundaries, unlike generate_lap2d_matrix
    assert A[99, 100] == 0.0 and A[100, 99] == 0.0 and A[0, 100] == -1.0


def test_mtx_solve_matches_reference(oracle, mtx_path, reference_probe):
    A, _, _ = oracle.read_mtx_dense(mtx_path)
    n = A.shape[0]
    b = oracle.init_source_term(n)
    row = reference_probe["mtx_lap2D_5pt_n100"][0]
    x, r = oracle.solve(A, b, None, 60, 1e-10, 1)             # bounded: 60 iterations ~ 4 s on one core
    assert r["iterations"] == 60
    # converged-run anchors need the full ~488 iterations (30 s single core): checked in the GPU suite
    # against the same golden row; here only sanity of the early iterations.
    assert r["residual_prev"] > 1.0 and np.isfinite(r["x_norm"])
    assert row["k"] == 488


# ---- oracle/_ref: the reference's own reader, compiled from its sources ------------------------------------
MTX_CASES = {
    "general_dups.mtx": "%%MatrixMarket matrix coordinate real general\n% comment\n%another\n4 4 7\n1 1 2.5\n2 2 1\n3 3 4e0\n4 4 -3\n1 3 -1\n1 3 -7\n4 1 0.125\n",
    "symmetric.mtx": "%%MatrixMarket matrix coordinate real symmetric\n5 5 6\n1 1 4\n2 1 -1\n3 3 4\n5 2 7.5\n5 5 1\n4 4 2\n",
    "integer_caps.mtx": "%%MatrixMarket MATRIX COORDINATE INTEGER GENERAL\n3 3 3\n1 1 1\n2 2 2\n3 3 3\n",
}


def test_oracle_reader_matches_the_reference_reader(oracle, mtx_path, tmp_path):
    if not oracle.ref_available():
        pytest.skip("oracle/_ref not built (needs /root/reference: `make -C oracle ref`)")
    A_ref = oracle.ref_read_mtx_dense(mtx_path)
    A, _, _ = oracle.read_mtx_dense(mtx_path)
    assert np.array_equal(A, A_ref)
    for name, text in MTX_CASES.items():
        f = tmp_path / name
        f.write_text(text)
        assert np.array_equal(oracle.read_mtx_dense(str(f))[0], oracle.ref_read_mtx_dense(str(f))), name


# ---- the on-the-fly twin used as checker for libcgx's opt-in banded storage at large n -------------------------
@pytest.mark.parametrize("n,max_iter,p", [(2, None, 1), (3, None, 2), (100, None, 1), (1000, None, 3), (2048, 150, 4), (4096, 100, 1)])
def test_banded_twin_matches_the_dense_oracle(oracle, n, max_iter, p):
    """Same matrix (generator rule cg.cc:181-185), never materialised: must reproduce the dense restatement, which
    is the one pinned to the reference.  Differences are summation-order noise of a 5-term row sum."""
    xd, rd = oracle.solve_lap2d(n, max_iter, 1e-10, p)
    xb, rb = oracle.solve_lap2d_banded(n, max_iter, 1e-10, p)
    assert rd["iterations"] == rb["iterations"] and rd["converged"] == rb["converged"]
    assert np.linalg.norm(xd - xb) <= 1e-13 * np.linalg.norm(xd)
    if rd["residual_prev"] > 1e-6:
        assert abs(rd["residual_prev"] - rb["residual_prev"]) <= 1e-9 * rd["residual_prev"]
    assert abs(rd["x_norm"] - rb["x_norm"]) <= 1e-13 * rd["x_norm"]


def test_banded_twin_at_the_pinned_large_sizes(oracle, reference_probe):
    """N = 32768 / 500 iterations and N = 46340 / 200 iterations against the REFERENCE's recorded outputs
    (tests/golden/reference_probe.json): the twin is pinned by the reference directly, not only through the dense port."""
    for n, k in ((32768, 500), (46340, 200)):
        row = [q for q in reference_probe["generated_large"] if q["n"] == n][0]
        x, r = oracle.solve_lap2d_banded(n, k, 1e-10, 1)
        assert r["iterations"] == row["k"] == k
        assert abs(r["residual_prev"] - row["residual"]) <= 1e-6 * row["residual"]
        assert abs(r["x_norm"] - row["x_norm"]) <= 1e-12 * row["x_norm"]
        for i, v in row["x_samples"].items():
            assert abs(x[int(i)] - v) <= 1e-12 * abs(v), i


def test_oracle_solves_do_not_depend_on_the_callers_floating_point_environment(oracle):
    """The checker must give the same bits whatever the host process has done to the calling thread's MXCSR (a library
    loaded with fast-math start-up code sets flush-to-zero, anything may switch the rounding mode): the solve entry points
    run under round-to-nearest without FTZ / DAZ and put the caller's environment back."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    FE_TONEAREST, FE_TOWARDZERO = 0x000, 0xc00
    n, it, p = 600, 40, 3
    x_ref, r_ref = oracle.solve_lap2d(n, it, 0.0, p)
    assert oracle.fp_state() == 0x1f80
    try:
        assert libm.fesetround(FE_TOWARDZERO) == 0
        assert oracle.fp_state() != 0x1f80                      # the thread really computes toward zero now
        x, r = oracle.solve_lap2d(n, it, 0.0, p)
        xb, rb = oracle.solve_lap2d_banded(n, it, 0.0, p)
        assert libm.fegetround() == FE_TOWARDZERO               # the caller's environment is put back
    finally:
        libm.fesetround(FE_TONEAREST)
    assert np.array_equal(x, x_ref) and r["residual_prev"] == r_ref["residual_prev"] and r["x_norm"] == r_ref["x_norm"]
    xb_ref, rb_ref = oracle.solve_lap2d_banded(n, it, 0.0, p)
    assert np.array_equal(xb, xb_ref) and rb["residual_prev"] == rb_ref["residual_prev"]


# ---- the hash matrix of the dense-data probes (test data, not a reference function) ------------------------------------
HASH_KNOWN = [   # (seed, i, j) -> value, from an independent big-integer evaluation of the definition (tests/golden: none needed)
    ((0, 0, 0), "0x1.3836e97a68cbcp-2"),
    ((1, 0, 1), "0x1.a7f58127596bcp-1"),
    ((0x5EEDC0DE, 16376, 32767), "0x1.ae8f8e8b1527cp-2"),
    ((0x5EEDC0DE, 46339, 0), "0x1.20f991012f210p-2"),
    ((2 ** 63 + 5, 131071, 131070), "0x1.90d517aa00f98p-1"),
]


def _hash_entry_bigint(seed, i, j, sym=False, diag=0.0):
    """cgx_kernels.h hash_entry once more, in Python integers: mix64 = the splitmix64 finaliser."""
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    if i == j and diag != 0.0:
        return diag
    a, b = (min(i, j), max(i, j)) if sym else (i, j)
    return (mix(mix(seed) ^ ((a << 32) | b)) >> 11) * 2.0 ** -52 - 1.0


def test_hash_matrix_known_answers(oracle):
    for (seed, i, j), want in HASH_KNOWN:
        assert _hash_entry_bigint(seed, i, j) == float.fromhex(want)
        assert oracle.hash_rows(max(i, j) + 1, i, 1, seed, False, 0.0)[0, j] == float.fromhex(want)
        assert oracle.hash_rows_numpy(max(i, j) + 1, [i], seed, False, 0.0)[0, j] == float.fromhex(want)


@pytest.mark.parametrize("sym,diag", [(False, 0.0), (True, 0.0), (True, 215.5), (False, -3.0)])
def test_hash_matrix_restatements_agree(oracle, sym, diag):
    n, seed = 411, 0xABCDEF
    A = oracle.hash_rows(n, 0, n, seed, sym, diag)
    assert np.array_equal(A, oracle.hash_rows_numpy(n, range(n), seed, sym, diag))
    assert np.array_equal(A[37:40], oracle.hash_rows(n, 37, 3, seed, sym, diag))
    rng = np.random.default_rng(3)
    for i, j in rng.integers(0, n, size=(50, 2)):
        assert A[i, j] == _hash_entry_bigint(seed, int(i), int(j), sym, diag)
    assert A.min() >= -1.0 and A.max() < 1.0 or diag != 0.0
    if sym:
        assert np.array_equal(A, A.T)
    oracle.set_threads(4)
    try:
        assert np.array_equal(A, oracle.hash_rows(n, 0, n, seed, sym, diag))   # thread count does not change the fill
    finally:
        oracle.set_threads(1)


def test_hash_matrix_with_dominant_diagonal_is_spd(oracle):
    """What the rate runs and the fused-K1 test rely on: symmetric + diag just above 2 sqrt(n/3) is positive definite."""
    n = 1500
    diag = 1.03 * 2.0 * np.sqrt(n / 3.0)
    A = oracle.hash_rows(n, 0, n, 99, True, diag)
    w = np.linalg.eigvalsh(A)
    assert w[0] > 0.0 and w[-1] / w[0] < 400.0
    x, r = oracle.solve(A, oracle.init_source_term(n), max_iter=n, tol=1e-10)
    assert r["converged"] and r["rel_residual"] < 1e-11


# ---- the third-party arithmetic: the same recurrence through a real OpenBLAS -----------------------------------------------
def _solve_through_openblas(A, b, max_iter, tol, psize=1):
    """code/MPI/cg.cc:38-156 once more, with the three BLAS calls of the reference (cblas_dgemv cg.cc:80,101,146; cblas_ddot
    :91,105,116; cblas_daxpy :82,110,113,128) made through the OpenBLAS that numpy / scipy bundle (scipy.linalg.blas: the
    Fortran entry points of the same kernels the reference's cblas_* wrappers call; the reference used OpenBLAS 0.3.10,
    figures/gprof.png) -- row blocks as partition_matrix cuts them, the Allreduce sums in rank order."""
    from scipy.linalg import blas
    n = b.size
    starts, counts = [0], [n]
    if psize > 1:
        n_loc = n // psize
        starts = [q * n_loc for q in range(psize)]
        counts = [n_loc] * (psize - 1) + [n - (psize - 1) * n_loc]
    blocks = [np.asfortranarray(A[s:s + c].T) for s, c in zip(starts, counts)]      # A_block^T column-major = A_block row-major
    gemv = lambda v: np.concatenate([blas.dgemv(1.0, At, v, trans=1) for At in blocks])   # noqa: E731
    dot = lambda u, v: sum(blas.ddot(u[s:s + c], v[s:s + c]) for s, c in zip(starts, counts))   # noqa: E731
    x = np.zeros(n)
    r = b - gemv(x)                                              # cg.cc:79-82
    p = r.copy()                                                 # cg.cc:85
    rsold = dot(r, r)                                            # cg.cc:91-92
    k = 0
    while k < max_iter:
        Ap = gemv(p)                                             # cg.cc:100-102
        conj = dot(p, Ap)                                        # cg.cc:105-106
        alpha = rsold / max(conj, rsold * 1e-14)                 # cg.cc:107
        x = blas.daxpy(p, x, a=alpha)                            # cg.cc:110
        r = blas.daxpy(Ap, r, a=-alpha)                          # cg.cc:113
        rsnew = dot(r, r)                                        # cg.cc:116-117
        if np.sqrt(rsnew) < tol:                                 # cg.cc:120-121
            break
        beta = rsnew / rsold                                     # cg.cc:124
        p = blas.daxpy(p, r.copy(), a=beta)                      # cg.cc:127-129: p = r + beta p
        rsold = rsnew                                            # cg.cc:132
        k += 1
    return x, k, float(np.sqrt(rsold))


@pytest.mark.parametrize("n,max_iter,psize", [(2048, 200, 1), (2048, 200, 4), (4096, 200, 1), (4096, 50, 8), (3000, 150, 3), (1024, None, 1),
                                              (1000, None, 3)])
def test_oracle_agrees_with_the_recurrence_through_a_real_openblas(oracle, reference_probe, n, max_iter, psize):
    """The reference's GEMV / dot / axpy live in OpenBLAS (unpinned; 0.3.10 in figures/gprof.png), whose summation order is not
    the oracle's.  The same recurrence driven through the OpenBLAS kernels numpy / scipy bundle must land on the oracle's
    result to the tolerance DESIGN.md states for parity (x to 1e-12, residual to 1e-10 relative at a fixed iteration count),
    and on the reference's own recorded numbers where the survey recorded this run."""
    A = oracle.generate_lap2d(n)
    b = oracle.init_source_term(n)
    it = n if max_iter is None else max_iter
    xb, kb, resb = _solve_through_openblas(A, b, it, 1e-10, psize)
    xo, ro = oracle.solve(A, b, max_iter=it, tol=1e-10, psize=psize)
    if max_iter is None:
        assert ro["converged"] and abs(kb - ro["iterations"]) <= 0.15 * ro["iterations"]       # k is not a stable observable
        assert np.linalg.norm(xb - xo) <= 1e-10 * np.linalg.norm(xo)
    else:
        assert kb == ro["iterations"] == max_iter
        assert np.linalg.norm(xb - xo) <= 1e-12 * np.linalg.norm(xo)
        assert abs(resb - ro["residual_prev"]) <= 1e-10 * ro["residual_prev"]
    rows = [q for q in reference_probe["generated"] if q["n"] == n and q["max_iter"] == max_iter and q["ranks"] == psize]
    if n != 3000:
        assert rows, "a run the survey recorded from the compiled reference"
        assert abs(np.linalg.norm(xb) - rows[0]["x_norm"]) <= 1e-6 * rows[0]["x_norm"]
        if max_iter is not None:
            assert kb == rows[0]["k"] and abs(resb - rows[0]["residual"]) <= 1e-6 * rows[0]["residual"]
        else:
            assert abs(kb - rows[0]["k"]) <= 0.15 * rows[0]["k"]
