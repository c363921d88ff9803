"""Run the CPU oracle at the BASELINE.json sizes and store its scalars + sampled x.

Usage (dev container, ~10 min single core, up to 17 GB RAM):
    python tests/golden/make_oracle_large.py [n:max_iter:psize ...]
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
DEFAULT = ["16384:200:1", "23170:200:2", "32768:500:1"]


def main(argv):
    cases = argv or DEFAULT
    probe = json.load(open(os.path.join(HERE, "reference_probe.json")))
    want = {r["n"]: sorted(int(i) for i in r["x_samples"]) for r in probe["generated_large"]}
    data = json.load(open(OUT)) if os.path.exists(OUT) else {"_provenance": "outputs of oracle/cg_oracle.c (this repo) produced by tests/golden/make_oracle_large.py", "cases": []}
    for c in cases:
        n, mi, ps = (int(v) for v in c.split(":"))
        x, r = O.solve_lap2d(n, mi, 1e-10, ps)
        idx = sorted(set(want.get(n, []) + [0, n // 3, n // 2, n - 1]))
        row = {"n": n, "max_iter": mi, "psize": ps, "k": r["iterations"], "residual": r["residual_prev"],
               "residual_last": r["residual_last"], "x_norm": r["x_norm"], "rel_residual": r["rel_residual"],
               "x_samples": {str(i): float(x[i]) for i in idx}, "seconds_loop": r["seconds_loop"]}
        data["cases"] = [q for q in data["cases"] if not (q["n"] == n and q["max_iter"] == mi and q["psize"] == ps)] + [row]
        json.dump(data, open(OUT, "w"), indent=1)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
