"""What handing a HOST matrix over the C ABI costs (cgx_set_matrix_dense: the `.mtx` / caller-supplied dense path, cg.cu:307-321
after Matrix::read): seconds and GB/s of the upload of this rank's row block over PCIe from pageable host memory, and the
PCIe-inclusive rate of a whole solve (upload + solve()) beside the resident one.  Dev tool; one JSON line per size."""
import json, os, sys, time
import numpy as np, torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
for n, iters in ((10000, 607), (16384, 200), (32768, 100)):
    A = O.generate_lap2d(n)
    with pkg.CGSolver() as s:
        s.set_matrix_dense(A[:64].copy() if False else A)      # first call: allocation + first-touch of the runtime's staging
        t0 = time.perf_counter(); s.set_matrix_dense(A); t1 = time.perf_counter()
        s.init_source_term(1.0 / n); s.set_max_iter(iters); s.tolerance(0.0)
        x = np.zeros(n); r = s.solve(x)
        x = np.zeros(n); r = s.solve(x)
    up = t1 - t0
    print(json.dumps({"n": n, "matrix_GB": 8.0 * n * n / 1e9, "upload_s": up, "upload_GBs": 8.0 * n * n / up / 1e9, "iterations": iters,
                      "solve_s": r["seconds_solve"], "iterations_per_s_resident": iters / r["seconds_solve"],
                      "iterations_per_s_incl_upload": iters / (r["seconds_solve"] + up)}), flush=True)
    del A
