"""Rate of the fused K1 on the reference's generated matrix (99.985 % zeros at N = 32768) and on dense incompressible data
(the hash matrix of cgx_probe_fill_matrix_hash, symmetric with a dominant diagonal so that CG keeps running on it), same
process shape, for rocprofv3 / PMC runs and A/B on one box (dev tool).
Env: N (32768), SHARDS (1; > 1 = logical row blocks on one GPU), MATRIX = lap2d | hash, STEPS (400), WARM (100), SMI=1 samples
rocm-smi clocks / power while the timed steps run.  Prints one JSON line."""
import json, os, subprocess, sys, threading, time
import numpy as np, torch  # noqa: F401  (torch first: one HIP runtime in the process)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
n = int(os.environ.get("N", "32768")); P = int(os.environ.get("SHARDS", "1")); matrix = os.environ.get("MATRIX", "lap2d")
steps = int(os.environ.get("STEPS", "400")); warm = int(os.environ.get("WARM", "100"))
s = pkg.CGSolver(comm_mode=pkg.COMM_LOOPBACK if P > 1 else pkg.COMM_SELF, nranks=P, profile_gemv=1)
s.generate_lap2d_matrix(n)
diag = 0.0
if matrix == "hash":
    diag = 1.03 * 2.0 * (n / 3.0) ** 0.5
    s.probe_fill_matrix_hash(0x5EEDC0DE, symmetric=True, diag=diag)
s.set_max_iter(10 ** 6); s.tolerance(0.0); s.init_source_term(1.0 / n)
smi = []
stop = threading.Event()
def sample():
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(out).values()))
            smi.append({k: v for k, v in card.items() if "sclk" in k or "mclk" in k or "fclk" in k or "ower" in k})
        except Exception as e:  # noqa: BLE001
            smi.append({"error": repr(e)[:80]})
        time.sleep(0.2)
s.solve_begin(np.zeros(n)); s.solve_steps(warm)
th = None
if os.environ.get("SMI") == "1":
    th = threading.Thread(target=sample, daemon=True); th.start(); time.sleep(0.3)
t0 = time.perf_counter(); done = s.solve_steps(steps); t1 = time.perf_counter()
stop.set()
if th: th.join(timeout=15)
samples = s.gemv_samples()
r = s.solve_end()
rows = n // P
alg = 8.0 * (rows * n + n + rows)
med = float(np.median(samples))
print(json.dumps({"n": n, "shards": P, "matrix": matrix, "diag": diag, "plan": s.gemv_plan(0), "steps": steps, "done_early": bool(done),
                  "k1_median_ms": med, "k1_mean_ms": float(np.mean(samples)), "k1_min_ms": float(np.min(samples)), "k1_max_ms": float(np.max(samples)),
                  "k1_samples": int(samples.size), "k1_GBs": alg / med / 1e6, "frac_of_8TBs": alg / med / 1e6 / 8000.0,
                  "ms_per_step_wall": (t1 - t0) * 1e3 / steps, "residual_prev": r["residual_prev"], "finite": bool(np.isfinite(r["residual_prev"])),
                  "smi": smi[:: max(1, len(smi) // 6)] if smi else None}))
