"""ctypes view of oracle/libcg_oracle.so -- the CPU parity oracle.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product package (see cg_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libcg_oracle.so")


class OracleResult(C.Structure):
    _fields_ = [
        ("iterations", C.c_int),
        ("converged", C.c_int),
        ("residual_prev", C.c_double),
        ("residual_last", C.c_double),
        ("x_norm", C.c_double),
        ("rel_residual", C.c_double),
        ("seconds_loop", C.c_double),
        ("seconds_solve", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


def build(force=False):
    """Compile the oracle with gcc (CPU only)."""
    src = [os.path.join(_HERE, f) for f in ("cg_oracle.c", "cg_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcg_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        L.oracle_partition.argtypes = [C.c_int, C.c_int, ip, ip]
        L.oracle_partition.restype = None
        L.oracle_generate_lap2d_rows.argtypes = [C.c_int, C.c_int, C.c_int, dp]
        L.oracle_generate_lap2d_rows.restype = None
        L.oracle_init_source_term.argtypes = [C.c_int, C.c_double, dp]
        L.oracle_init_source_term.restype = None
        L.oracle_solve.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(OracleResult)]
        L.oracle_solve.restype = C.c_int
        L.oracle_solve_lap2d.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, C.POINTER(OracleResult)]
        L.oracle_solve_lap2d.restype = C.c_int
        L.oracle_solve_lap2d_banded.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, C.POINTER(OracleResult)]
        L.oracle_solve_lap2d_banded.restype = C.c_int
        L.oracle_gemv.argtypes = [C.c_int, C.c_int, dp, C.c_long, dp, dp]
        L.oracle_gemv.restype = None
        L.oracle_dot.argtypes = [C.c_int, dp, dp]
        L.oracle_dot.restype = C.c_double
        L.oracle_axpy.argtypes = [C.c_int, C.c_double, dp, dp]
        L.oracle_axpy.restype = None
        L.oracle_read_mtx_dense.argtypes = [C.c_char_p, ip, ip, ip, ip, C.POINTER(dp)]
        L.oracle_read_mtx_dense.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_set_threads.restype = None
        L.oracle_hash_rows.argtypes = [C.c_int, C.c_long, C.c_long, C.c_ulonglong, C.c_int, C.c_double, dp]
        L.oracle_hash_rows.restype = None
        L.oracle_time_gemv_rows.argtypes = [C.c_int, C.c_int, C.c_int]
        L.oracle_time_gemv_rows.restype = C.c_double
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def partition(n, psize):
    s = (C.c_int * psize)()
    c = (C.c_int * psize)()
    lib().oracle_partition(n, psize, s, c)
    return list(s), list(c)


def generate_lap2d(n, row0=0, nrows=None):
    nrows = n if nrows is None else nrows
    A = np.empty((nrows, n), dtype=np.float64)
    lib().oracle_generate_lap2d_rows(n, row0, nrows, _dp(A))
    return A


def hash_rows(n, row0, nrows, seed, symmetric=False, diag=0.0):
    """Rows [row0, row0+nrows) of the n x n counter-based hash matrix (oracle_hash_rows): what cgx_probe_fill_matrix_hash leaves
    in HBM, rebuilt on the host.  Not a reference function: dense, incompressible test data for the GEMV of cg.cc:101-102."""
    A = np.empty((nrows, n), dtype=np.float64)
    lib().oracle_hash_rows(n, int(row0), int(nrows), int(seed), 1 if symmetric else 0, float(diag), _dp(A))
    return A


def hash_rows_numpy(n, rows, seed, symmetric=False, diag=0.0):
    """The same definition written a second time, in numpy integer arithmetic (rows: any list of row indices): the
    cross-check of the C loop above, and of the device fill at small sizes."""
    M = (1 << 64) - 1

    def mix(z):   # splitmix64 finaliser on uint64 arrays (wrapping arithmetic)
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))

    with np.errstate(over="ignore"):
        sm = mix(np.array([int(seed) & M], dtype=np.uint64))[0]
        i = np.asarray(rows, dtype=np.uint64)[:, None]
        j = np.arange(n, dtype=np.uint64)[None, :]
        a, b = (np.minimum(i, j), np.maximum(i, j)) if symmetric else (np.broadcast_to(i, (i.shape[0], n)), np.broadcast_to(j, (i.shape[0], n)))
        h = mix(sm ^ ((a << np.uint64(32)) | b))
    A = (h >> np.uint64(11)).astype(np.float64) * 2.0 ** -52 - 1.0
    if diag != 0.0:
        for k, r in enumerate(np.asarray(rows, dtype=np.int64)):
            if r < n:
                A[k, r] = diag
    return A


def init_source_term(n, h=None):
    b = np.empty(n, dtype=np.float64)
    lib().oracle_init_source_term(n, (1.0 / n) if h is None else h, _dp(b))
    return b


def gemv(A, x):
    A = np.ascontiguousarray(A, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    m, n = A.shape
    y = np.empty(m, dtype=np.float64)
    lib().oracle_gemv(m, n, _dp(A), n, _dp(x), _dp(y))
    return y


def dot(x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().oracle_dot(x.size, _dp(x), _dp(y))


def solve(A, b, x0=None, max_iter=None, tol=1e-10, psize=1):
    """CGSolver::solve (code/MPI/cg.cc:38-156) with psize logical ranks. Returns (x, dict)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = b.size
    assert A.shape == (n, n)
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    res = OracleResult()
    rc = lib().oracle_solve(_dp(A), _dp(b), _dp(x), n, n if max_iter is None else max_iter,
                            tol, psize, C.byref(res))
    if rc:
        raise RuntimeError("oracle_solve failed: %d" % rc)
    return x, res.as_dict()


def fp_state():
    """MXCSR of the calling thread, exception flags masked out (0x1f80 = round to nearest, no flush-to-zero)."""
    L = lib()
    L.oracle_fp_state.restype = C.c_uint
    return L.oracle_fp_state() & ~0x3f


def set_threads(nthreads):
    """Host threads working on the logical ranks' row blocks (results do not depend on it)."""
    lib().oracle_set_threads(int(nthreads))


def solve_lap2d(n, max_iter=None, tol=1e-10, psize=1):
    """generate_lap2d_matrix(n) + init_source_term(1/n) + solve, x0 = 0 (cg_main.cc:28-55)."""
    x = np.zeros(n, dtype=np.float64)
    res = OracleResult()
    rc = lib().oracle_solve_lap2d(n, n if max_iter is None else max_iter, tol, psize, _dp(x), C.byref(res))
    if rc:
        raise RuntimeError("oracle_solve_lap2d failed: %d" % rc)
    return x, res.as_dict()


def solve_lap2d_banded(n, max_iter=None, tol=1e-10, psize=1):
    """solve_lap2d with the generator's rule applied on the fly (no n x n block): checker for large n only."""
    x = np.zeros(n, dtype=np.float64)
    res = OracleResult()
    rc = lib().oracle_solve_lap2d_banded(n, n if max_iter is None else max_iter, tol, psize, _dp(x), C.byref(res))
    if rc:
        raise RuntimeError("oracle_solve_lap2d_banded failed: %d" % rc)
    return x, res.as_dict()


def read_mtx_dense(path):
    m = C.c_int()
    n = C.c_int()
    nz = C.c_int()
    sym = C.c_int()
    ptr = C.POINTER(C.c_double)()
    rc = lib().oracle_read_mtx_dense(os.fsencode(path), C.byref(m), C.byref(n), C.byref(nz), C.byref(sym),
                                     C.byref(ptr))
    if rc:
        raise RuntimeError("oracle_read_mtx_dense failed: %d" % rc)
    A = np.ctypeslib.as_array(ptr, shape=(m.value, n.value)).copy()
    C.CDLL(None).free(ptr)
    return A, nz.value, bool(sym.value)


_REF_LIB_PATH = os.path.join(_HERE, "_ref", "libref_matrix.so")
_ref = None


def ref_available():
    """True if oracle/_ref/libref_matrix.so (the REFERENCE's own Matrix-Market reader, built from its sources by
    `make -C oracle ref` where /root/reference exists) is present.  The built file travels to the GPU box."""
    return os.path.exists(_REF_LIB_PATH)


def ref_read_mtx_dense(path):
    """Matrix::read of the reference itself (code/MPI/matrix.cc:6-22).  Valid files only: it exit()s on errors."""
    global _ref
    if _ref is None:
        _ref = C.CDLL(_REF_LIB_PATH)
        _ref.ref_matrix_read.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double))]
        _ref.ref_matrix_read.restype = C.c_int
    m = C.c_int()
    n = C.c_int()
    ptr = C.POINTER(C.c_double)()
    rc = _ref.ref_matrix_read(os.fsencode(path), C.byref(m), C.byref(n), C.byref(ptr))
    if rc:
        raise RuntimeError("ref_matrix_read failed: %d" % rc)
    A = np.ctypeslib.as_array(ptr, shape=(m.value, n.value)).copy()
    C.CDLL(None).free(ptr)
    return A


def time_gemv_rows(n, nrows, reps):
    return lib().oracle_time_gemv_rows(n, nrows, reps)
