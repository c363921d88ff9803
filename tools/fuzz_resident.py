"""Randomised parity sweep of the LDS-resident solver (dev tool): random n <= 4096, generated or hash matrix, random b / x0,
random number of iterations, the loop cut into random pieces; against the CPU oracle (||dx||/||x||) and, bit for bit,
against the same solve in one launch.  One context is reused for a run of cases (changing sizes: the exchange buffer is
laid out anew every time).  python tools/fuzz_resident.py SECONDS [SEED]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0; worst = 0.0; bad = 0
s = pkg.CGSolver(gemv_variant=40000)
while time.time() - t0 < budget:
    if cases % 50 == 49:
        s.close(); s = pkg.CGSolver(gemv_variant=40000)
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 520), rng.integers(500, 1100), rng.integers(1000, 2049), 2048, 1024,
                        rng.integers(2049, 4097), rng.integers(2049, 4097), 4096, 3072]))
    iters = max(1, min(int(rng.integers(1, 60)), n // 2))
    hashed = n >= 8 and rng.integers(0, 2) == 0
    hseed, hdiag = int(rng.integers(1, 2 ** 62)), 1.03 * 2.0 * (n / 3.0) ** 0.5 + 1.0
    A = O.hash_rows(n, 0, n, hseed, True, hdiag) if hashed else O.generate_lap2d(n)
    b = rng.standard_normal(n) if rng.integers(0, 2) else O.init_source_term(n)
    x0 = rng.standard_normal(n) if rng.integers(0, 2) else np.zeros(n)
    tol = 0.0 if rng.integers(0, 3) else 10.0 ** float(rng.integers(-9, -2))
    s.generate_lap2d_matrix(n)
    if hashed:
        s.probe_fill_matrix_hash(hseed, symmetric=True, diag=hdiag)
    s.set_source_term(b); s.set_max_iter(iters); s.tolerance(tol)
    assert s.gemv_plan()["variant"] == 4
    x1 = x0.copy(); r1 = s.solve(x1)
    # the same loop in pieces
    s.solve_begin(x0)
    left = iters
    while left > 0:
        k = int(rng.integers(1, left + 1)); s.solve_steps(k); left -= k
    x2 = np.zeros(n); r2 = s.solve_end(x2)
    xo, ro = O.solve(A, b, x0, iters, tol, 1)
    nx = np.linalg.norm(xo)
    err = np.linalg.norm(x1 - xo) / nx if nx > 0 else np.linalg.norm(x1 - xo)
    if np.isnan(xo).any():    # n = 1 with b = 0: the reference's own 0/0 (cg.cc:107, rsold = 0): NaN in the same places on both sides
        err = 0.0 if np.array_equal(np.isnan(x1), np.isnan(xo)) else np.inf
    same = np.array_equal(x1, x2, equal_nan=True) and r1["iterations"] == r2["iterations"] and (
        r1["residual_prev"] == r2["residual_prev"] or (np.isnan(r1["residual_prev"]) and np.isnan(r2["residual_prev"])))
    # a converging run may break one iteration apart from the oracle's (rsnew against tol at rounding level): then x differs
    k_ok = r1["iterations"] == ro["iterations"] and r1["converged"] == ro["converged"]
    worst = max(worst, err if k_ok else 0.0)
    cases += 1
    if cases % 500 == 0:
        print("... %d cases, %.0f s, worst %.3e" % (cases, time.time() - t0, worst), flush=True)
    if not same or (k_ok and not err <= 1e-11) or (not k_ok and tol == 0.0):
        bad += 1
        print("MISMATCH n=%d iters=%d hashed=%s tol=%g: err %.3e same=%s k %d/%d conv %d/%d" % (
            n, iters, hashed, tol, err, same, r1["iterations"], ro["iterations"], r1["converged"], ro["converged"]), flush=True)
s.close()
print("fuzz_resident: %d cases in %.0f s, worst ||dx||/||x|| = %.3e, mismatches %d" % (cases, time.time() - t0, worst, bad))
sys.exit(1 if bad else 0)
