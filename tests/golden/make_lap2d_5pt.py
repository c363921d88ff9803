"""Regenerate the reference's only data fixture, lap2D_5pt_n100.mtx, from its definition.

The file is the 5-point Laplacian on a 100 x 100 grid (10000 unknowns), lower triangle only,
`coordinate real symmetric`, 1-based `i j v`, column-major order, 29800 entries
(/root/reference/code/MPI/lap2D_5pt_n100.mtx:1-3).  tests/test_oracle.py checks that the output is
byte-identical to the reference's copy whenever /root/reference is present.
"""
import os
import sys


def generate(g=100):
    n = g * g
    lines = ["%%MatrixMarket matrix coordinate real symmetric", "% Generated 20-Nov-2014"]
    entries = []
    for j in range(1, n + 1):                 # column j, rows i >= j
        entries.append("%d %d  4" % (j, j))
        if j % g != 0:                        # right neighbour in the same grid row
            entries.append("%d %d -1" % (j + 1, j))
        if j + g <= n:                        # neighbour in the next grid row
            entries.append("%d %d -1" % (j + g, j))
    lines.append("%d %d %d" % (n, n, len(entries)))
    return "\n".join(lines + entries) + "\n"


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                             "lap2D_5pt_n100.mtx")
    with open(out, "w") as f:
        f.write(generate())
    print("wrote", out)
