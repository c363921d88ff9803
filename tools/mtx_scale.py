"""How long does cgx_read_matrix take on a large Matrix-Market file?  5-point Laplacian of a g x g grid, banded
storage (a dense block cannot exist at this size).  python tools/mtx_scale.py [g]   (dev tool)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import __graft_entry__ as g_
import make_lap2d_5pt
pkg = g_.load_package()
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
path = "/tmp/lap2d_5pt_g%d.mtx" % g
t0 = time.time()
open(path, "w").write(make_lap2d_5pt.generate(g))
print("file: %.1f MB, written in %.1f s" % (os.path.getsize(path) / 1e6, time.time() - t0), flush=True)
n = g * g
for threads in (1, 4, 0):          # the parser alone (host only): one thread, four, the library's default
    t0 = time.time(); m_, n_, sym_, I, J, a = pkg.cgx.parse_matrix_market(path, threads); dt = time.time() - t0
    print("parse only, threads=%s: %.3f s (%.0f MB/s, %.1f M entries/s)" % (threads or "default", dt, os.path.getsize(path) / 1e6 / dt, len(a) / 1e6 / dt), flush=True)
    del I, J, a
with pkg.CGSolver(matrix_format=pkg.MATRIX_BANDED) as s:
    t0 = time.time(); s.read_matrix(path); t1 = time.time()
    print("cgx_read_matrix: %.2f s for n=%d; diagonals %s" % (t1 - t0, n, s.matrix_format(0)[1]), flush=True)
    s.init_source_term(1.0 / n); s.set_max_iter(200)
    x = np.zeros(n); r = s.solve(x)
    print("200 iterations: loop %.4f s, residual %.6e" % (r["seconds_loop"], r["residual_prev"]), flush=True)
os.remove(path)
