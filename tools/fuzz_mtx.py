"""Feed the Matrix-Market reader corrupted files: it must answer with a status, never crash or hang. (dev tool)"""
import os, sys, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
base = "%%MatrixMarket matrix coordinate real symmetric\n% c\n12 12 20\n" + "".join(
    "%d %d %s\n" % (random.randint(1, 12), random.randint(1, 12), random.choice(["1", "-2.5", "3e0", "0", "1e-3"])) for _ in range(20))
path = "/tmp/fuzz.mtx"
ok = err = 0
with pkg.CGSolver() as sd, pkg.CGSolver(matrix_format=pkg.MATRIX_BANDED) as sb:
    for it in range(3000):
        b = bytearray(base.encode())
        for _ in range(random.randint(1, 6)):
            op = random.randint(0, 4)
            pos = random.randrange(len(b)) if b else 0
            if op == 0 and b: b[pos] = random.randrange(256)
            elif op == 1 and b: del b[pos:pos + random.randint(1, 30)]
            elif op == 2: b[pos:pos] = bytes(random.randrange(256) for _ in range(random.randint(1, 20)))
            elif op == 3: b = b[:pos]
            else: b[pos:pos] = random.choice([b"\n", b" ", b"999999999999 ", b"-1 ", b"1e999 ", b"nan ", b"\x00"])
        open(path, "wb").write(bytes(b))
        for s in (sd, sb):
            try:
                s.read_matrix(path); ok += 1
            except pkg.CgxError as e:
                assert e.status in (2, 5, 7), e
                err += 1
print("fuzz_mtx: %d accepted, %d rejected with a status, no crash" % (ok, err))
