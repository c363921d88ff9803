"""Dev script: first contact with the MI355X -- correctness probes and a K1 variant sweep."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package(); O = g.load_oracle()
print("torch", torch.__version__, torch.cuda.get_device_name(0), flush=True)
g.smoke()

rng = np.random.default_rng(0)
def check_gemv(n, variant, mode=pkg.COMM_SELF, nranks=1):
    with pkg.CGSolver(comm_mode=mode, nranks=nranks, gemv_variant=variant) as s:
        s.generate_lap2d_matrix(n)
        A = O.generate_lap2d(n)
        rows = []
        for i in range(nranks if mode == pkg.COMM_LOOPBACK else 1):
            a, r0 = s.probe_matrix_rows(i); rows.append(a)
        Ad = np.vstack(rows)
        assert np.array_equal(Ad, A), "generator mismatch"
        p = rng.standard_normal(n)
        y, pap = s.probe_gemv(p)
        yo = O.gemv(A, p)
        err = np.max(np.abs(y - yo)) / np.max(np.abs(yo))
        perr = abs(pap - O.dot(p, yo)) / abs(O.dot(p, yo))
        print("gemv n=%d variant=%d mode=%d P=%d maxrel=%.2e pAp rel=%.2e" % (n, variant, mode, nranks, err, perr), flush=True)
        assert err < 1e-13 and perr < 1e-12

for v in (0, 10821, 10441, 10241, 10181, 20821, 20441, 20241):
    check_gemv(1000, v)
check_gemv(1001, 0); check_gemv(777, 20441)
check_gemv(1000, 0, pkg.COMM_LOOPBACK, 3)
check_gemv(2048, 20441, pkg.COMM_LOOPBACK, 4)

def solve(n, mi, mode=pkg.COMM_SELF, nranks=1, variant=0):
    with pkg.CGSolver(comm_mode=mode, nranks=nranks, gemv_variant=variant) as s:
        s.generate_lap2d_matrix(n)
        if mi: s.set_max_iter(mi)
        s.init_source_term(1.0 / n)
        x = np.zeros(n); r = s.solve(x)
    return x, r
for (n, mi, mode, P) in [(2048, 200, 0, 1), (2048, 200, 1, 4), (1000, None, 1, 3), (1024, None, 0, 1), (4096, 50, 0, 1)]:
    x, r = solve(n, mi, mode, P)
    xo, ro = O.solve_lap2d(n, mi, 1e-10, P)
    print("solve n=%d mi=%s P=%d: k=%d/%d res=%.6e/%.6e xnorm=%.6e relres=%.3e |dx|/|x|=%.2e t=%.3fs" % (
        n, mi, P, r["iterations"], ro["iterations"], r["residual_prev"], ro["residual_prev"], r["x_norm"], r["rel_residual"],
        np.linalg.norm(x - xo) / np.linalg.norm(xo), r["seconds_solve"]), flush=True)

# K1 sweep at the roofline point
N = int(os.environ.get("SWEEP_N", "32768"))
variants = [int(v) for v in os.environ.get("SWEEP_VARIANTS", "10821,10820,10441,10440,10811,10281,20821,20441,20241,20421,20820").split(",")]
pads = [int(v) for v in os.environ.get("SWEEP_PADS", "0,16").split(",")]
out = []
for pad in pads:
    for v in variants:
        with pkg.CGSolver(gemv_variant=v, lda_pad=pad, profile_gemv=True) as s:
            s.generate_lap2d_matrix(N); s.set_max_iter(10**6); s.init_source_term(1.0 / N)
            s.solve_begin(np.zeros(N)); s.solve_steps(5)
            torch.cuda.synchronize(); t0 = time.time(); s.solve_steps(40); t1 = time.time()
            r = s.solve_end()
        gbs = r["gemv_bytes"] / (r["gemv_ms_avg"] * 1e-3) / 1e9
        row = dict(N=N, variant=v, pad=pad, gemv_ms_avg=r["gemv_ms_avg"], gemv_ms_min=r["gemv_ms_min"], gemv_GBs=gbs,
                   frac=gbs / 8000, iter_ms=(t1 - t0) / 40 * 1e3)
        out.append(row); print(json.dumps(row), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/sweep_%d.json" % N, "w"), indent=1)
