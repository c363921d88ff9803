"""Randomised parity sweep (dev tool, not part of the suite): random size, row-block count, K1 shape and storage
format; fixed number of iterations with tol = 0 on both sides; x, residual and ||x|| against the CPU oracle.
python tools/fuzz_parity.py SECONDS [SEED]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
VARIANTS = [0, 0, 0, 10821, 10811, 10441, 10421, 10241, 10281, 10181, 11611, 20821, 20441, 20241, 20181,
            10822, 10842, 10442, 10282, 11612,   # ..2: the one-round form of the column-split K1
            10823, 10824, 10825, 10444, 10445]   # ..3/4/5: the same with 2 / 4 / 8 XCD-affine column pieces (chunked consumer)
BANDED_VARIANTS = [0, 30001, 30002]        # K1b: default / direct / LDS windows
t0 = time.time(); cases = 0; worst = 0.0
while time.time() - t0 < budget:
    n = int(rng.choice([rng.integers(1, 40), rng.integers(40, 700), rng.integers(700, 3000), rng.integers(3000, 6000)]))
    P = int(rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 11, 16]))
    banded = bool(rng.integers(0, 2))
    variant = int(rng.choice(BANDED_VARIANTS)) if banded else int(rng.choice(VARIANTS))
    iters = int(rng.integers(1, 40))
    iters = max(1, min(iters, n // 2))   # past ~n iterations the recurrence only moves rounding noise
    every = int(rng.choice([0, 1, 3, 16]))
    x0 = rng.standard_normal(n) if rng.integers(0, 2) else np.zeros(n)
    # a third of the dense cases: the hash matrix (every element a different number, symmetric, dominant diagonal: SPD) filled
    # on the device by the test probe instead of the generator's five diagonals
    hashed = (not banded) and n >= 8 and rng.integers(0, 3) == 0
    hseed, hdiag = int(rng.integers(1, 2 ** 62)), 1.03 * 2.0 * (n / 3.0) ** 0.5 + 1.0
    A = O.hash_rows(n, 0, n, hseed, True, hdiag) if hashed else O.generate_lap2d(n)
    b = O.init_source_term(n)
    if rng.integers(0, 3) == 0:
        b = rng.standard_normal(n)
    only = os.environ.get("ONLY")          # "n,P": replay one case of a seed (the draws above are consumed as usual)
    if only and only != "%d,%d" % (n, P):
        cases += 1
        continue
    if only:
        # diagnostics for a replayed case: the same problem through other forms of the path
        xo, ro = O.solve(A, b, x0, iters, 0.0, P)
        for (bd, v, PP) in ((banded, variant, P), (banded, 30001 if banded else 0, P), (False, 0, P), (banded, variant, 1), (banded, variant, 2), (banded, variant, 4)):
            with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if PP > 1 else pkg.COMM_SELF, nranks=PP, gemv_variant=v, check_every=every,
                              matrix_format=pkg.MATRIX_BANDED if bd else pkg.MATRIX_DENSE) as s:
                s.generate_lap2d_matrix(n)
                if hashed and not bd:
                    s.probe_fill_matrix_hash(hseed, symmetric=True, diag=hdiag)
                s.set_source_term(b); s.set_max_iter(iters); s.tolerance(0.0)
                x = x0.copy(); r = s.solve(x)
            xo2, ro2 = O.solve(A, b, x0, iters, 0.0, PP)
            print("banded=%d variant=%d P=%d: err vs oracle(P) %.3e  res %r / %r  x0 zero=%s" % (bd, v, PP, np.linalg.norm(x - xo2) / np.linalg.norm(xo2), r["residual_prev"], ro2["residual_prev"], not x0.any()), flush=True)
        sys.exit(0)
    with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=variant,
                      check_every=every, matrix_format=pkg.MATRIX_BANDED if banded else pkg.MATRIX_DENSE) as s:
        s.generate_lap2d_matrix(n)
        if hashed:
            s.probe_fill_matrix_hash(hseed, symmetric=True, diag=hdiag)
        s.set_source_term(b); s.set_max_iter(iters); s.tolerance(0.0)
        x = x0.copy(); r = s.solve(x)
    xo, ro = O.solve(A, b, x0, iters, 0.0, P)
    nx = np.linalg.norm(xo)
    err = np.linalg.norm(x - xo) / nx if nx > 0 else np.linalg.norm(x - xo)
    ok = r["iterations"] == ro["iterations"] == iters and (err < 1e-11 or not np.isfinite(nx))
    if np.isfinite(ro["residual_prev"]) and ro["residual_prev"] > 1e-9 * np.linalg.norm(b):
        ok = ok and abs(r["residual_prev"] - ro["residual_prev"]) <= 1e-6 * ro["residual_prev"]
    worst = max(worst, err if np.isfinite(err) else 0.0)
    cases += 1
    if not ok:
        # which side moved?  the oracle again (same inputs, same process), its floating-point environment, the GPU again
        xo3, ro3 = O.solve(A, b, x0, iters, 0.0, P)
        print("oracle again: residual %r (first %r), |dx oracle-oracle| %.3e, MXCSR 0x%x, case %d" % (
            ro3["residual_prev"], ro["residual_prev"], np.linalg.norm(xo3 - xo), O.fp_state(), cases), flush=True)
        print("MISMATCH n=%d P=%d banded=%d hashed=%d variant=%d iters=%d every=%d err=%.3e res %r vs %r" %
              (n, P, banded, hashed, variant, iters, every, err, r["residual_prev"], ro["residual_prev"]), flush=True)
        # Everything needed to analyse the case offline goes to a file: the inputs (the matrix by its rule: generate_lap2d(n)),
        # what the GPU returned, what the oracle returned the first and the second time, the scalars of all three and the
        # oracle's floating-point state.  (Round 3's one disagreement, profiles/r03_fuzz_long.txt, could only be replayed, and
        # the replay passed: the oracle's first answer was gone.)
        out_dir = os.environ.get("FUZZ_DUMP_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out"))
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, "fuzz_mismatch_seed%s_case%d_n%d_P%d.npz" % (sys.argv[2] if len(sys.argv) > 2 else "0", cases, n, P))
        with pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, gemv_variant=variant,
                          check_every=every, matrix_format=pkg.MATRIX_BANDED if banded else pkg.MATRIX_DENSE) as s:
            s.generate_lap2d_matrix(n)
            if hashed:
                s.probe_fill_matrix_hash(hseed, symmetric=True, diag=hdiag)
            s.set_source_term(b); s.set_max_iter(iters); s.tolerance(0.0)
            x_again = x0.copy(); r_again = s.solve(x_again)
        np.savez(path, matrix_rule=("hash_rows(%d, seed %d, symmetric, diag %r)" % (n, hseed, hdiag)) if hashed else "generate_lap2d(%d)" % n, n=n, P=P, banded=banded, variant=variant, iters=iters, check_every=every,
                 b=b, x0=x0, x_gpu=x, x_gpu_again=x_again, x_oracle_first=xo, x_oracle_again=xo3,
                 scalars_gpu=np.array([r["residual_prev"], r["x_norm"], r["rel_residual"]]),
                 scalars_gpu_again=np.array([r_again["residual_prev"], r_again["x_norm"], r_again["rel_residual"]]),
                 scalars_oracle_first=np.array([ro["residual_prev"], ro["x_norm"], ro["rel_residual"]]),
                 scalars_oracle_again=np.array([ro3["residual_prev"], ro3["x_norm"], ro3["rel_residual"]]),
                 mxcsr=O.fp_state(), gpu_bit_identical_again=np.array_equal(x, x_again), oracle_bit_identical_again=np.array_equal(xo, xo3))
        print("case dumped to %s (GPU again bit-identical: %s, oracle again bit-identical: %s)" % (
            path, np.array_equal(x, x_again), np.array_equal(xo, xo3)), flush=True)
        sys.exit(1)
    nhashed = globals().get("nhashed", 0) + (1 if hashed else 0)
    if cases % 200 == 0:
        print("%d cases ok (%d on the hash matrix), worst ||dx||/||x|| = %.2e, %.0f s" % (cases, nhashed, worst, time.time() - t0), flush=True)
print("done: %d cases ok (%d on the hash matrix), worst ||dx||/||x|| = %.2e" % (cases, globals().get("nhashed", 0), worst))
