"""A FINITE soak of the persistent kernels' exchange: long-lived contexts, back-to-back solves of at most 300
iterations (tol = 1e-10: the smaller problems end by the break of cg.cc:120-121, the larger ones at the cap), generated and
hash matrices, every x finite and bit-identical to the first solve of its problem.  Every iteration is one epoch of the
tagged-word exchange carrying finite doubles -- unlike tools/soak_resident.sh's single solve of 15 M iterations, whose
recurrence runs into 0/0 after a few hundred iterations and soaks liveness only.

  python tools/soak_finite.py [EPOCHS_PER_PROBLEM]      (default 1 300 000: 8 resident + 4 streaming problems > 15 M epochs)
"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("CGX_RESIDENT", None)
import __graft_entry__ as g
pkg = g.load_package()

per_problem = int(sys.argv[1]) if len(sys.argv) > 1 else 1_300_000
problems = [(n, kind) for n in (1024, 2048, 3584, 4096) for kind in ("lap2d", "hash")] + [(n, kind) for n in (5000, 8192) for kind in ("lap2d", "hash")]
total_epochs = total_solves = 0
t_all = time.time()
for n, kind in problems:
    budget = per_problem if n <= 4096 else per_problem // 8      # the streaming sizes run 10-30 x longer per epoch
    with pkg.CGSolver(gemv_variant=0) as s:
        s.generate_lap2d_matrix(n)
        if kind == "hash":
            s.probe_fill_matrix_hash(1234 + n, symmetric=True, diag=1.03 * 2.0 * (n / 3.0) ** 0.5)
        s.set_max_iter(300)
        s.tolerance(1e-10)
        s.init_source_term(1.0 / n)
        plan = s.gemv_plan()["variant"]
        assert plan in (4, 5), plan
        x_ref = np.zeros(n)
        r_ref = s.solve(x_ref)
        assert np.all(np.isfinite(x_ref)) and np.isfinite(r_ref["residual_prev"]), (n, kind)
        per_solve = r_ref["iterations"] + (1 if r_ref["converged"] else 0)
        epochs, solves, t0, worst = per_solve, 1, time.time(), 0.0
        x = np.empty(n)
        while epochs < budget:
            x[:] = 0.0
            t1 = time.perf_counter()
            r = s.solve(x)
            worst = max(worst, time.perf_counter() - t1)
            if not (np.array_equal(x, x_ref) and r["iterations"] == r_ref["iterations"] and r["residual_prev"] == r_ref["residual_prev"]):
                print(json.dumps({"n": n, "matrix": kind, "FAILED_at_solve": solves, "iterations": r["iterations"]}), flush=True)
                sys.exit(1)
            epochs += per_solve
            solves += 1
        rec = s.resident_record()
        assert rec["fallbacks"] == 0 and s.gemv_plan()["variant"] == plan
    total_epochs += epochs
    total_solves += solves
    print(json.dumps({"n": n, "matrix": kind, "kernel": "resident" if plan == 4 else "streaming", "epochs": epochs, "solves": solves,
                      "iterations_per_solve": r_ref["iterations"], "converged": bool(r_ref["converged"]), "all_x_finite_and_bit_identical": True,
                      "seconds": round(time.time() - t0, 1), "slowest_solve_ms": round(worst * 1e3, 2), "fallbacks": 0}), flush=True)
print(json.dumps({"total_epochs": total_epochs, "total_solves": total_solves, "seconds": round(time.time() - t_all, 1),
                  "every_epoch_carried_finite_doubles": True}), flush=True)
