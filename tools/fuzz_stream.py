"""Randomised parity sweep of the streaming persistent kernel (dev tool): random 4097 <= n <= 11264 (every number of column steps
that keeps rows of A on the chip, and the sizes around their boundaries), random row pitch, generated or hash matrix, random b / x0,
random number of iterations, the loop cut into random pieces; against the CPU oracle (||dx||/||x||) and, bit for bit, against the
same solve in one launch.  python tools/fuzz_stream.py SECONDS [SEED]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("CGX_RESIDENT", None)
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time(); cases = 0; worst = 0.0; bad = 0
edges = [4097, 4098, 5119, 5120, 5121, 6144, 6145, 7168, 7169, 8191, 8192, 8193, 9216, 9217, 10240, 10241, 11264]
while time.time() - t0 < budget:
    pad = int(rng.choice([-1, -1, 0, 2, 16, 64, int(rng.integers(0, 200))]))
    with pkg.CGSolver(gemv_variant=50000, lda_pad=pad) as s:
        for _ in range(4):
            n = int(rng.choice([rng.integers(4097, 6145), rng.integers(4097, 6145), rng.integers(6145, 9217), rng.integers(9217, 11265), rng.choice(edges)]))
            iters = int(rng.integers(1, 25))
            hashed = rng.integers(0, 2) == 0
            hseed, hdiag = int(rng.integers(1, 2 ** 62)), 1.03 * 2.0 * (n / 3.0) ** 0.5 + 1.0
            A = O.hash_rows(n, 0, n, hseed, True, hdiag) if hashed else O.generate_lap2d(n)
            b = rng.standard_normal(n) if rng.integers(0, 2) else O.init_source_term(n)
            x0 = rng.standard_normal(n) if rng.integers(0, 2) else np.zeros(n)
            tol = 0.0 if rng.integers(0, 3) else 10.0 ** float(rng.integers(-9, -2))
            s.generate_lap2d_matrix(n)
            if hashed:
                s.probe_fill_matrix_hash(hseed, symmetric=True, diag=hdiag)
            s.set_source_term(b); s.set_max_iter(iters); s.tolerance(tol)
            assert s.gemv_plan()["variant"] == 5
            x1 = x0.copy(); r1 = s.solve(x1)
            s.solve_begin(x0)                      # the same loop in pieces
            left = iters
            while left > 0:
                k = int(rng.integers(1, left + 1)); s.solve_steps(k); left -= k
            x2 = np.zeros(n); r2 = s.solve_end(x2)
            xo, ro = O.solve(A, b, x0, iters, tol, 1)
            del A
            err = np.linalg.norm(x1 - xo) / max(np.linalg.norm(xo), np.linalg.norm(x0), 1e-300)
            same = np.array_equal(x1, x2) and r1["iterations"] == r2["iterations"] and r1["residual_prev"] == r2["residual_prev"]
            # a converging run may break one iteration apart from the oracle's (rsnew against tol at rounding level): then x differs
            k_ok = r1["iterations"] == ro["iterations"] and r1["converged"] == ro["converged"]
            worst = max(worst, err if k_ok else 0.0)
            cases += 1
            if not same or (k_ok and not err <= 1e-11) or (not k_ok and tol == 0.0) or s.resident_record()["fallbacks"]:
                bad += 1
                print("MISMATCH n=%d pad=%d iters=%d hashed=%s tol=%g: err %.3e same=%s k %d/%d conv %d/%d" % (
                    n, pad, iters, hashed, tol, err, same, r1["iterations"], ro["iterations"], r1["converged"], ro["converged"]), flush=True)
            if cases % 10 == 0:
                print("... %d cases, %.0f s, worst %.3e" % (cases, time.time() - t0, worst), flush=True)
            if time.time() - t0 > budget:
                break
print("fuzz_stream: %d cases in %.0f s, worst ||dx||/||x|| = %.3e, mismatches %d" % (cases, time.time() - t0, worst, bad))
sys.exit(1 if bad else 0)
