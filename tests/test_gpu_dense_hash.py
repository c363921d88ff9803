"""K1 on dense, incompressible data at the BASELINE sizes ("missing" 3).

The reference's generator leaves five non-zeros per row (code/MPI/cg.cc:178-186): at N = 32768 the matrix every other
full-size test streams is 99.985 % zeros, and a lane that read the wrong element inside the zero region of a row could not
be seen by row sums, linearity or symmetry.  The reference's GEMV is a general dense dgemv (cg.cc:101-102), so here the row
blocks are overwritten on the device with a counter-based hash matrix (cgx_probe_fill_matrix_hash: every element a
different number in [-1, 1), a pure function of (seed, i, j)) and the checker rebuilds rows on the host from the same
definition (oracle.hash_rows; a second restatement in numpy and known-answer values pin the definition in
tests/test_oracle.py).

Tolerance of a row of the GEMV: 4e-16 * sqrt(n) * (|A| . |p|) -- the summation-order bound used for every dense-random
GEMV test of this suite; a wrong, missing or doubled element is off by ~|a_ij p_j| ~ 0.5, twelve orders above it.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EEDC0DE


def _crossing_rows(n, lda, rows_in_block):
    """Local rows of a block whose byte offset row * lda * 8 lies next to a multiple of 4 GiB (32-bit offset wrap)."""
    out = []
    k = 1
    while True:
        r = (k << 32) // (lda * 8)
        if r >= rows_in_block:
            break
        out += [x for x in range(r - 3, r + 4) if 0 <= x < rows_in_block]
        k += 1
    return out


@pytest.mark.parametrize("n,p,sym,diag", [(1, 1, 0, 0.0), (17, 1, 1, 3.5), (1000, 1, 0, 0.0), (1001, 3, 1, 0.0), (1001, 3, 0, -2.0),
                                          (4096, 8, 1, 80.0)])
def test_hash_fill_is_bit_exact(gpu_pkg, oracle, n, p, sym, diag):
    mode = gpu_pkg.COMM_SELF if p == 1 else gpu_pkg.COMM_LOOPBACK
    with gpu_pkg.CGSolver(comm_mode=mode, nranks=p) as s:
        s.generate_lap2d_matrix(n)
        s.probe_fill_matrix_hash(SEED + n, symmetric=bool(sym), diag=diag)
        blocks = [s.probe_matrix_rows(i) for i in range(p)]
    A = np.vstack([b[0] for b in blocks])
    assert np.array_equal(A, oracle.hash_rows(n, 0, n, SEED + n, sym, diag))
    if sym:
        assert np.array_equal(A, A.T)
    if n >= 1000:   # dense and incompressible: no zeros, no repeated values to speak of
        assert np.count_nonzero(A) == A.size
        assert np.unique(A).size > 0.999 * (A.size / 2 if sym else A.size)


def test_fill_needs_a_dense_problem(gpu_pkg):
    with gpu_pkg.CGSolver() as s:
        with pytest.raises(gpu_pkg.CgxError):
            s.probe_fill_matrix_hash(1)
    with gpu_pkg.CGSolver(matrix_format=gpu_pkg.MATRIX_BANDED) as s:
        s.generate_lap2d_matrix(100)
        with pytest.raises(gpu_pkg.CgxError):
            s.probe_fill_matrix_hash(1)


@pytest.mark.parametrize("n,p", [(32768, 1), (32768, 8), (46340, 8), (46340, 1), (32768, 4), (32768, 2), (23170, 2), (16384, 1)])
def test_gemv_dense_hash_sampled_rows_at_baseline_sizes(gpu_pkg, oracle, n, p):
    """cgx_probe_gemv on the hash matrix at N = 32768 (one block of 8 GiB; 8 row blocks) and N = 46340 (8 uneven row blocks;
    one block of 17 GB): >= 256 sampled rows against oracle.gemv of the rebuilt rows -- first and last row of every block,
    the rows next to every 4 GiB boundary of a block's byte offsets, random others."""
    mode = gpu_pkg.COMM_SELF if p == 1 else gpu_pkg.COMM_LOOPBACK
    rng = np.random.default_rng(n + p)
    pv = rng.standard_normal(n)
    with gpu_pkg.CGSolver(comm_mode=mode, nranks=p) as s:
        s.generate_lap2d_matrix(n)
        s.probe_fill_matrix_hash(SEED, symmetric=False)
        plan = s.gemv_plan(0)
        y, pap = s.probe_gemv(pv)
    starts, counts = oracle.partition(n, p)
    lda = (n + 15) // 16 * 16 + 16
    rows = set()
    for s0, c in zip(starts, counts):
        rows |= {s0, s0 + 1, s0 + c - 2, s0 + c - 1}
        rows |= {s0 + r for r in _crossing_rows(n, lda, c)}
    rows |= set(int(v) for v in rng.integers(0, n, size=260))
    rows = sorted(r for r in rows if 0 <= r < n)
    assert len(rows) >= 256
    worst = 0.0
    for i0 in range(0, len(rows), 64):
        idx = rows[i0:i0 + 64]
        A = np.vstack([oracle.hash_rows(n, r, 1, SEED, False, 0.0) for r in idx])
        yo = oracle.gemv(A, pv)
        bound = 4e-16 * np.sqrt(n) * (np.abs(A) @ np.abs(pv))
        err = np.abs(y[idx] - yo)
        worst = max(worst, float(np.max(err / bound)))
        assert np.all(err <= bound), (plan, [(r, e, b) for r, e, b in zip(idx, err, bound) if e > b][:4])
    # the fused p.Ap of the same launch (cg.cc:105): all rows enter, so a wrong row anywhere would show here as well
    assert np.all(np.isfinite(y))
    assert abs(pap - float(np.dot(pv, y))) <= 1e-12 * float(np.sum(np.abs(pv * y)))
    print("n=%d P=%d plan=%s rows=%d worst err/bound=%.3f" % (n, p, plan, len(rows), worst))


@pytest.mark.parametrize("p", [1, 8])
def test_fused_k1_on_dense_hash_matrix_n32768(gpu_pkg, oracle, p):
    """The FUSED K1 (iteration head + p = r + beta p on the fly + GEMV; cg.cc:100-132) on dense data at N = 32768: three CG
    iterations on the symmetric hash matrix with a dominant diagonal (SPD), one block and eight, against oracle.solve on
    the same 8 GiB matrix rebuilt on the host."""
    n, iters = 32768, 3
    diag = 1.03 * 2.0 * np.sqrt(n / 3.0)   # just above the spectral radius of the off-diagonal part: condition ~ 70
    mode = gpu_pkg.COMM_SELF if p == 1 else gpu_pkg.COMM_LOOPBACK
    with gpu_pkg.CGSolver(comm_mode=mode, nranks=p) as s:
        s.generate_lap2d_matrix(n)
        s.probe_fill_matrix_hash(SEED + 1, symmetric=True, diag=diag)
        s.init_source_term(1.0 / n)
        s.set_max_iter(iters)
        s.tolerance(0.0)
        x = np.zeros(n)
        res = s.solve(x)
    oracle.set_threads(16)
    try:
        A = oracle.hash_rows(n, 0, n, SEED + 1, True, diag)
        xo, ro = oracle.solve(A, oracle.init_source_term(n), max_iter=iters, tol=0.0, psize=p)
    finally:
        oracle.set_threads(1)
    del A
    assert res["iterations"] == ro["iterations"] == iters
    assert np.linalg.norm(x - xo) <= 1e-12 * np.linalg.norm(xo)
    assert abs(res["residual_prev"] - ro["residual_prev"]) <= 1e-11 * ro["residual_prev"]
    assert abs(res["rel_residual"] - ro["rel_residual"]) <= 1e-9 * ro["rel_residual"]
