"""Run the CPU oracle at the BASELINE.json sizes and store its scalars + sampled x.

Usage (dev container, ~10 min single core, up to 17 GB RAM):
    python tests/golden/make_oracle_large.py [case ...]
      n:max_iter:psize        dense restatement (oracle_solve_lap2d: the n x n block is held in RAM)
      bn:max_iter:psize       the on-the-fly twin (oracle_solve_lap2d_banded: same solve_blocks, the five entries of a
                              row come from the generator's rule, cg.cc:181-185); seconds instead of minutes, any n
      cn:psize                dense restatement run to CONVERGENCE (max_iter = n as cg.cc:172, tol 1e-10): BASELINE.json
                              configs[1] is c10000:1 (~40 s, 800 MB); 20 sampled entries of x
      mtx                     tests/golden/lap2D_5pt_n100.mtx, init_source_term(1/n), run to convergence (tol 1e-10)
Writes/updates tests/golden/oracle_large.json.  The tests compare these oracle outputs with the
reference outputs in reference_probe.json (pinning the oracle) and the GPU path with both.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "oracle_large.json")
DEFAULT = ["16384:200:1", "23170:200:2", "32768:500:1", "b32768:200:4", "b46340:200:8", "mtx", "c10000:1"]


def sample_indices(n, extra=()):
    """16 spread positions + the ends + whatever the reference's probe sampled."""
    return sorted(set(list(extra) + [0, n // 3, n // 2, n - 1] + [(2 * j + 1) * n // 32 for j in range(16)]))


def main(argv):
    cases = argv or DEFAULT
    probe = json.load(open(os.path.join(HERE, "reference_probe.json")))
    want = {r["n"]: sorted(int(i) for i in r["x_samples"]) for r in probe["generated_large"]}
    data = json.load(open(OUT)) if os.path.exists(OUT) else {"_provenance": "outputs of oracle/cg_oracle.c (this repo) produced by tests/golden/make_oracle_large.py", "cases": []}
    data.setdefault("mtx", [])
    for c in cases:
        if c == "mtx":
            path = os.path.join(HERE, "lap2D_5pt_n100.mtx")
            A, nz, sym = O.read_mtx_dense(path)
            n = A.shape[0]
            x, r = O.solve(A, O.init_source_term(n), None, n, 1e-10, 1)     # loop bound m_n as code/CUDA/cg.cu:236
            idx = sample_indices(n, [0, 5000, 9999])
            row = {"file": "lap2D_5pt_n100.mtx", "n": n, "psize": 1, "k": r["iterations"], "converged": bool(r["converged"]),
                   "residual": r["residual_prev"], "residual_last": r["residual_last"], "x_norm": r["x_norm"],
                   "rel_residual": r["rel_residual"], "x_samples": {str(i): float(x[i]) for i in idx},
                   "seconds_loop": r["seconds_loop"]}
            data["mtx"] = [row]
            json.dump(data, open(OUT, "w"), indent=1)
            print(json.dumps(row), flush=True)
            continue
        if c.startswith("c"):
            n, ps = (int(v) for v in c[1:].split(":"))
            x, r = O.solve_lap2d(n, n, 1e-10, ps)                      # m_maxIter = size, cg.cc:172; cg_main.cc:31-55
            row = {"n": n, "psize": ps, "k": r["iterations"], "converged": bool(r["converged"]), "residual": r["residual_prev"],
                   "residual_last": r["residual_last"], "x_norm": r["x_norm"], "rel_residual": r["rel_residual"],
                   "x_samples": {str(i): float(x[i]) for i in sample_indices(n)}, "seconds_loop": r["seconds_loop"]}
            data["converged"] = [q for q in data.get("converged", []) if not (q["n"] == n and q["psize"] == ps)] + [row]
            json.dump(data, open(OUT, "w"), indent=1)
            print(json.dumps(row), flush=True)
            continue
        banded = c.startswith("b")
        n, mi, ps = (int(v) for v in c.lstrip("b").split(":"))
        x, r = (O.solve_lap2d_banded if banded else O.solve_lap2d)(n, mi, 1e-10, ps)
        idx = sample_indices(n, want.get(n, [])) if banded else sorted(set(want.get(n, []) + [0, n // 3, n // 2, n - 1]))
        row = {"n": n, "max_iter": mi, "psize": ps, "k": r["iterations"], "residual": r["residual_prev"],
               "residual_last": r["residual_last"], "x_norm": r["x_norm"], "rel_residual": r["rel_residual"],
               "x_samples": {str(i): float(x[i]) for i in idx}, "seconds_loop": r["seconds_loop"]}
        if banded:
            row["via"] = "oracle_solve_lap2d_banded (on-the-fly twin of the dense restatement)"
        data["cases"] = [q for q in data["cases"] if not (q["n"] == n and q["max_iter"] == mi and q["psize"] == ps)] + [row]
        json.dump(data, open(OUT, "w"), indent=1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
