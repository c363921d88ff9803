#!/usr/bin/env python3
"""The two MPI figures of the reference's plots.ipynb (strong-scaling speedup S_p, weak-scaling efficiency E_p) for the files
experiments/cg_mi355x.run writes, without the one hand edit the notebook needs: the notebook slices weak_scaling.txt in
blocks of ITEMS = 7 rows (plots.ipynb, show_weak_MPI_results: the reference ran 1..64 processes), here the rows per series
are read off the file (a series starts where psize is 1 again: 4 on an 8-GPU node, 1 on a one-GPU box).

    python3 results/plot_mi355x.py [--dir results] [--png PREFIX]     prints both tables; --png also draws them

Same file formats and the same quantities as the notebook: `n,psize,seconds` per line (code/MPI/cg_main.cc:62);
S_p = t(1)/t(p) beside Amdahl's p / (alpha p + 1 - alpha) for the sizes the notebook lists a serial fraction for;
E_p = t(N0, 1) / t(N0 sqrt(p), p)."""
import argparse
import os

# serial fractions per matrix size as plots.ipynb:5-6 lists them (its key 1148 is a typo for 1448; kept as it is there)
ALPHAS = {1024: 1 - 0.9753, 1148: 1 - 0.9852, 2048: 1 - 0.9910, 2896: 1 - 0.9935, 4096: 1 - 0.9950, 5792: 1 - 0.9966,
          8192: 1 - 0.9981, 11585: 1 - 0.9988, 16384: 1 - 0.9995}


def rows(path):
    """[(n, psize, seconds)] of one result file, file order."""
    out = []
    for line in open(path):
        c = line.rstrip("\n").split(",")
        out.append((int(c[0]), int(c[1]), float(c[2])))
    return out


def strong_series(path):
    """{n: [(psize, seconds, S_p, Amdahl bound or None)]}, sizes in first-seen order (the notebook's dict)."""
    by_n = {}
    for n, p, t in rows(path):
        by_n.setdefault(n, []).append((p, t))
    return {n: [(p, t, v[0][1] / t, (p / (ALPHAS[n] * p + 1 - ALPHAS[n])) if n in ALPHAS else None) for p, t in v]
            for n, v in by_n.items()}


def weak_series(path):
    """[[(n, psize, seconds, E_p)]]: one list per series; ITEMS = its length, taken from the file."""
    series = []
    for n, p, t in rows(path):
        if p == 1 or not series:
            series.append([])
        series[-1].append((n, p, t))
    return [[(n, p, t, s[0][2] / t) for n, p, t in s] for s in series]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--png", default="", help="also draw the two figures to PREFIX_strong.png / PREFIX_weak.png")
    a = ap.parse_args()
    strong = {f: strong_series(os.path.join(a.dir, f)) for f in ("strong_scaling.txt", "strong_scaling_n32768.txt")
              if os.path.exists(os.path.join(a.dir, f))}
    weak = {f: weak_series(os.path.join(a.dir, f)) for f in ("weak_scaling.txt", "weak_scaling_n16384.txt")
            if os.path.exists(os.path.join(a.dir, f))}
    for f, d in strong.items():
        print("%s  (S_p = t(1)/t(p); Amdahl where plots.ipynb has a serial fraction)" % f)
        for n, v in d.items():
            for p, t, s, am in v:
                print("  N=%-6d p=%-2d %10.6f s   S_p=%6.3f   Amdahl=%s" % (n, p, t, s, "%.3f" % am if am else "-"))
    for f, ss in weak.items():
        print("%s  (E_p = t(N0,1)/t(N,p); ITEMS = %s rows per series)" % (f, sorted({len(s) for s in ss})))
        for s in ss:
            for n, p, t, e in s:
                print("  N0=%-6d N=%-6d p=%-2d %10.6f s   E_p=%6.3f" % (s[0][0], n, p, t, e))
    if a.png:
        import matplotlib
        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
        fig, ax = plt.subplots(figsize=(10, 7))
        for d in strong.values():
            for n, v in d.items():
                line, = ax.loglog([q[0] for q in v], [q[2] for q in v], marker="o", label="N = %d" % n)
                if v[0][3]:
                    ax.loglog([q[0] for q in v], [q[3] for q in v], marker="x", linestyle="dashed", color=line.get_color())
        ax.set_xlabel("$p$ (MI355X)"); ax.set_ylabel("$S_p$"); ax.legend(); fig.tight_layout(); fig.savefig(a.png + "_strong.png")
        fig, ax = plt.subplots(figsize=(10, 7))
        for ss in weak.values():
            for s in ss:
                ax.semilogx([q[1] for q in s], [q[3] for q in s], marker="o", label="$N_0 = %d$" % s[0][0])
        ax.set_ylim([0, 1.05]); ax.set_xlabel("$p$ (MI355X)"); ax.set_ylabel("$E_p$"); ax.legend(); fig.tight_layout()
        fig.savefig(a.png + "_weak.png")


if __name__ == "__main__":
    main()
