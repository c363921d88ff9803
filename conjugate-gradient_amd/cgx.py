"""ctypes binding of libcgx.so (include/cgx.h) plus a Python mirror of the reference's CGSolver.

Only test/bench plumbing lives here: every number is computed by the HIP library.  There is no CPU
fallback: if libcgx.so is missing or no MI355X is visible, the calls raise.
Reference interface mirrored: class CGSolver, code/MPI/cg.hh:11-57 and code/CUDA/cg.hh:13-45.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libcgx.so")
UNIQUE_ID_BYTES = 128

IPC_HANDLE_BYTES = 64
COMM_SELF, COMM_LOOPBACK, COMM_RCCL, COMM_P2P = 0, 1, 2, 3
MATRIX_DENSE, MATRIX_BANDED = 0, 1   # BANDED: opt-in fast path, not the reference's storage (include/cgx.h)
MAX_DIAGONALS = 64

EXPORTS = [
    "cgx_config_init", "cgx_comm_unique_id", "cgx_create", "cgx_destroy", "cgx_last_error", "cgx_status_string",
    "cgx_get_comm_info", "cgx_get_gemv_plan", "cgx_get_resident_record", "cgx_p2p_export", "cgx_p2p_import", "cgx_p2p_selftest",
    "cgx_partition", "cgx_generate_lap2d_matrix", "cgx_set_matrix_dense", "cgx_read_matrix",
    "cgx_init_source_term", "cgx_set_source_term", "cgx_set_max_iter", "cgx_set_tolerance", "cgx_get_size",
    "cgx_get_matrix_format",
    "cgx_solve", "cgx_solve_begin", "cgx_solve_steps", "cgx_solve_end", "cgx_get_gemv_samples",
    "cgx_get_update_samples",
    "cgx_probe_gemv", "cgx_probe_time_gemv", "cgx_probe_vector_ops", "cgx_probe_get_matrix_rows",
    "cgx_probe_get_source_term", "cgx_probe_set_fault_after", "cgx_probe_set_resident_limit", "cgx_probe_persistent_plan",
    "cgx_probe_parse_matrix_market", "cgx_probe_p2p_mailbox_to_host", "cgx_probe_fill_matrix_hash",
    "cgx_probe_set_p2p_epoch", "cgx_probe_get_p2p_epoch", "cgx_probe_p2p_host_mailboxes", "cgx_probe_resident_test",
]


class Config(C.Structure):
    _fields_ = [
        ("struct_version", C.c_int), ("comm_mode", C.c_int), ("device", C.c_int), ("rank", C.c_int),
        ("nranks", C.c_int), ("unique_id", C.c_ubyte * UNIQUE_ID_BYTES), ("gemv_variant", C.c_int),
        ("lda_pad", C.c_int), ("check_every", C.c_int), ("profile_gemv", C.c_int), ("profile_update", C.c_int),
        ("p2p_mailbox_kib", C.c_int), ("p2p_timeout_ms", C.c_int), ("p2p_separate_exchange", C.c_int),
        ("matrix_format", C.c_int), ("profile_first", C.c_int), ("profile_markers", C.c_int),
        ("p2p_no_acquire_fence", C.c_int), ("p2p_tagged", C.c_int),
    ]


class Result(C.Structure):
    _fields_ = [
        ("iterations", C.c_int), ("converged", C.c_int), ("residual_prev", C.c_double),
        ("residual_last", C.c_double), ("x_norm", C.c_double), ("rel_residual", C.c_double),
        ("seconds_solve", C.c_double), ("seconds_loop", C.c_double), ("gemv_ms_avg", C.c_double),
        ("gemv_ms_min", C.c_double), ("gemv_launches", C.c_longlong), ("gemv_bytes", C.c_double),
        ("gemv_ms_median", C.c_double), ("gemv_ms_max", C.c_double), ("gemv_discarded", C.c_longlong),
        ("steps_device_ms", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


class CgxError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("libcgx status %d: %s" % (status, msg))
        self.status = status


def build(force=False):
    """hipcc --offload-arch=gfx950 build of libcgx.so + cgsolver (cross-compiles without a GPU)."""
    args = ["make", "-C", _PKG, "-s"] + (["-B"] if force else []) + ["all"]
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib():
    """Load libcgx.so.  Raises OSError if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("libcgx.so not built (run `make -C conjugate-gradient_amd` or __graft_entry__.build()); "
                          "the product path has no CPU fallback")
        # torch first, if it is there: libcgx then binds to the one HIP / RCCL runtime torch ships instead of bringing
        # /opt/rocm's as a second one into the process -- with two, whichever comes second may see no device (seen on the GPU
        # box when the library had been loaded before torch).  CGX_NO_TORCH=1 keeps torch out (the no-torch dev tools).
        if "torch" not in sys.modules and os.environ.get("CGX_NO_TORCH") != "1":
            try:
                import torch  # noqa: F401
            except Exception:   # noqa: BLE001 -- a box without torch: /opt/rocm's runtime alone is fine
                pass
        L = C.CDLL(LIB_PATH)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        vp = C.c_void_p
        L.cgx_config_init.argtypes = [C.POINTER(Config)]
        L.cgx_config_init.restype = None
        L.cgx_comm_unique_id.argtypes = [C.POINTER(C.c_ubyte)]
        L.cgx_create.argtypes = [C.POINTER(vp), C.POINTER(Config)]
        L.cgx_destroy.argtypes = [vp]
        L.cgx_last_error.argtypes = [vp]
        L.cgx_last_error.restype = C.c_char_p
        L.cgx_status_string.argtypes = [C.c_int]
        L.cgx_status_string.restype = C.c_char_p
        L.cgx_get_comm_info.argtypes = [vp, ip, ip, ip, C.c_char_p]
        L.cgx_get_gemv_plan.argtypes = [vp, C.c_int, ip]
        L.cgx_get_resident_record.argtypes = [vp, C.POINTER(C.c_longlong)]
        L.cgx_p2p_export.argtypes = [vp, C.POINTER(C.c_ubyte)]
        L.cgx_p2p_import.argtypes = [vp, C.POINTER(C.c_ubyte)]
        L.cgx_p2p_selftest.argtypes = [vp, C.c_int, ip]
        L.cgx_partition.argtypes = [C.c_int, C.c_int, ip, ip]
        L.cgx_generate_lap2d_matrix.argtypes = [vp, C.c_int]
        L.cgx_set_matrix_dense.argtypes = [vp, dp, C.c_long, C.c_int]
        L.cgx_read_matrix.argtypes = [vp, C.c_char_p]
        L.cgx_init_source_term.argtypes = [vp, C.c_double]
        L.cgx_set_source_term.argtypes = [vp, dp]
        L.cgx_set_max_iter.argtypes = [vp, C.c_int]
        L.cgx_set_tolerance.argtypes = [vp, C.c_double]
        L.cgx_get_size.argtypes = [vp, ip, ip]
        L.cgx_get_matrix_format.argtypes = [vp, C.c_int, ip, ip, ip, dp]
        L.cgx_solve.argtypes = [vp, dp, C.POINTER(Result)]
        L.cgx_solve_begin.argtypes = [vp, dp]
        L.cgx_solve_steps.argtypes = [vp, C.c_int, ip]
        L.cgx_solve_end.argtypes = [vp, dp, C.POINTER(Result)]
        L.cgx_get_gemv_samples.argtypes = [vp, dp, C.c_int, ip]
        L.cgx_get_update_samples.argtypes = [vp, dp, C.c_int, ip]
        L.cgx_probe_gemv.argtypes = [vp, dp, dp, dp]
        L.cgx_probe_time_gemv.argtypes = [vp, C.c_int, dp]
        L.cgx_probe_vector_ops.argtypes = [vp, C.c_int, C.c_double, C.c_double, dp, dp, dp, dp, dp]
        L.cgx_probe_get_matrix_rows.argtypes = [vp, C.c_int, dp, ip, ip]
        L.cgx_probe_get_source_term.argtypes = [vp, C.c_int, dp]
        L.cgx_probe_set_fault_after.argtypes = [vp, C.c_int]
        L.cgx_probe_set_resident_limit.argtypes = [vp, C.c_int]
        L.cgx_probe_persistent_plan.argtypes = [C.c_int, C.c_int, C.c_long, C.c_int, C.POINTER(C.c_long)]
        L.cgx_probe_p2p_mailbox_to_host.argtypes = [vp]
        L.cgx_probe_fill_matrix_hash.argtypes = [vp, C.c_ulonglong, C.c_int, C.c_double]
        L.cgx_probe_p2p_host_mailboxes.argtypes = [vp, C.c_char_p, C.c_int]
        L.cgx_probe_set_p2p_epoch.argtypes = [vp, C.c_int, C.c_ulonglong]
        L.cgx_probe_resident_test.argtypes = [vp, C.c_ulonglong, C.c_int]
        L.cgx_probe_get_p2p_epoch.argtypes = [vp, C.c_int, C.POINTER(C.c_ulonglong)]
        L.cgx_probe_parse_matrix_market.argtypes = [C.c_char_p, C.c_int, ip, ip, ip, ip, ip, ip, dp, C.c_long, C.c_char_p, C.c_int]
        for name in EXPORTS:
            fn = getattr(L, name)
            if fn.restype is C.c_int and name not in ("cgx_config_init",):
                fn.restype = C.c_int
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def partition(n, psize):
    """CGSolver::partition_matrix (code/MPI/cg.cc:236-268)."""
    s = (C.c_int * psize)()
    c = (C.c_int * psize)()
    st = lib().cgx_partition(n, psize, s, c)
    if st:
        raise CgxError(st, "cgx_partition")
    return list(s), list(c)


def comm_unique_id():
    buf = (C.c_ubyte * UNIQUE_ID_BYTES)()
    st = lib().cgx_comm_unique_id(buf)
    if st:
        raise CgxError(st, lib().cgx_last_error(None).decode(errors="replace"))
    return bytes(buf)


def parse_matrix_market(path, threads=0):
    """The library's Matrix-Market parser alone (host only): (m, n, symmetric, I, J, a) with 0-based indices in file order."""
    m, n, nz, sym = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    err = C.create_string_buffer(512)
    st = lib().cgx_probe_parse_matrix_market(os.fsencode(path), int(threads), C.byref(m), C.byref(n), C.byref(nz), C.byref(sym),
                                             None, None, None, 0, err, 512)
    if st:
        raise CgxError(st, err.value.decode(errors="replace"))
    I = np.zeros(nz.value, dtype=np.int32)
    J = np.zeros(nz.value, dtype=np.int32)
    a = np.zeros(nz.value, dtype=np.float64)
    st = lib().cgx_probe_parse_matrix_market(os.fsencode(path), int(threads), C.byref(m), C.byref(n), C.byref(nz), C.byref(sym),
                                             I.ctypes.data_as(C.POINTER(C.c_int)), J.ctypes.data_as(C.POINTER(C.c_int)), _dp(a),
                                             nz.value, err, 512)
    if st:
        raise CgxError(st, err.value.decode(errors="replace"))
    return m.value, n.value, bool(sym.value), I, J, a


class CGSolver:
    """Python mirror of the reference's CGSolver (same member names) over the C ABI.

    read_matrix / generate_lap2d_matrix / init_source_term / set_max_iter / tolerance / solve / m / n
    keep the reference's meaning; the constructor takes what `srun -n P` and MPI_Init gave the reference.
    """

    def __init__(self, comm_mode=COMM_SELF, nranks=1, rank=0, device=0, unique_id=None, gemv_variant=0,
                 lda_pad=-1, check_every=0, profile_gemv=False, p2p_timeout_ms=0, p2p_mailbox_kib=0,
                 p2p_separate_exchange=False, matrix_format=MATRIX_DENSE, profile_first=False,
                 profile_markers=False, p2p_no_acquire_fence=False, profile_update=False, p2p_tagged=False):
        L = lib()
        cfg = Config()
        L.cgx_config_init(C.byref(cfg))
        cfg.comm_mode = comm_mode
        cfg.nranks = nranks
        cfg.rank = rank
        cfg.device = device
        cfg.gemv_variant = gemv_variant
        cfg.lda_pad = lda_pad
        cfg.check_every = check_every
        cfg.profile_gemv = int(profile_gemv)   # n > 0: every n-th K1 launch is event-timed
        cfg.profile_update = 1 if profile_update else 0   # ... and the update kernel of the same iterations
        cfg.p2p_timeout_ms = p2p_timeout_ms
        cfg.p2p_mailbox_kib = p2p_mailbox_kib
        cfg.p2p_separate_exchange = 1 if p2p_separate_exchange else 0
        cfg.matrix_format = matrix_format
        cfg.profile_first = 1 if profile_first else 0
        cfg.profile_markers = 1 if profile_markers else 0
        cfg.p2p_no_acquire_fence = 1 if p2p_no_acquire_fence else 0
        cfg.p2p_tagged = 1 if p2p_tagged else 0
        if unique_id is not None:
            assert len(unique_id) == UNIQUE_ID_BYTES
            C.memmove(cfg.unique_id, bytes(unique_id), UNIQUE_ID_BYTES)
        self._h = C.c_void_p()
        st = L.cgx_create(C.byref(self._h), C.byref(cfg))
        if st:
            self._h = C.c_void_p()
            raise CgxError(st, L.cgx_last_error(None).decode(errors="replace"))
        self.nranks = nranks
        self.rank = rank

    # -- plumbing -------------------------------------------------------------------------------
    def _check(self, st):
        if st:
            raise CgxError(st, lib().cgx_last_error(self._h).decode(errors="replace"))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().cgx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def comm_info(self):
        """What the transport really spans: {'comm_mode', 'ranks_wired', 'rank_seen', 'device_id'} (cgx_get_comm_info)."""
        mode, wired, seen = C.c_int(), C.c_int(), C.c_int()
        dev = C.create_string_buffer(32)
        self._check(lib().cgx_get_comm_info(self._h, C.byref(mode), C.byref(wired), C.byref(seen), dev))
        return {"comm_mode": mode.value, "ranks_wired": wired.value, "rank_seen": seen.value,
                "device_id": dev.value.decode(errors="replace")}

    def gemv_plan(self, local_shard=0):
        """The K1 launch shape of a local shard (cgx_get_gemv_plan): which kernel form runs on it."""
        v = (C.c_int * 8)()
        self._check(lib().cgx_get_gemv_plan(self._h, int(local_shard), v))
        return dict(zip(("variant", "R", "U", "waves", "light", "split", "grid", "ncols"), list(v)))

    def resident_record(self):
        """What the waits inside the persistent launches of the current / most recent solve cost (cgx_get_resident_record)."""
        v = (C.c_longlong * 10)()
        self._check(lib().cgx_get_resident_record(self._h, v))
        return dict(zip(("iterations", "watch_repeats", "gather_repeats", "launches", "wg0_first_wait_ticks", "wg0_longest_wait_ticks",
                         "max_first_wait_ticks", "max_longest_wait_ticks", "fallbacks", "persistent"), list(v)))

    # -- direct peer exchange wire-up (COMM_P2P) ------------------------------------------------------
    def p2p_export(self):
        buf = (C.c_ubyte * IPC_HANDLE_BYTES)()
        self._check(lib().cgx_p2p_export(self._h, buf))
        return bytes(buf)

    def p2p_import(self, handles):
        """handles: bytes of length nranks*64, rank order (every rank's p2p_export())."""
        assert len(handles) == self.nranks * IPC_HANDLE_BYTES
        buf = (C.c_ubyte * len(handles)).from_buffer_copy(handles)
        self._check(lib().cgx_p2p_import(self._h, buf))

    def p2p_selftest(self, rounds=32):
        ok = C.c_int()
        self._check(lib().cgx_p2p_selftest(self._h, int(rounds), C.byref(ok)))
        return bool(ok.value)

    def matrix_format(self, local_shard=0):
        """(format, offsets, device bytes) of a local shard's row block; offsets is [] for dense storage."""
        fmt, nd, nbytes = C.c_int(), C.c_int(), C.c_double()
        offs = (C.c_int * MAX_DIAGONALS)()
        self._check(lib().cgx_get_matrix_format(self._h, int(local_shard), C.byref(fmt), C.byref(nd), offs, C.byref(nbytes)))
        return fmt.value, list(offs[:nd.value]), nbytes.value

    # -- reference interface --------------------------------------------------------------------
    def generate_lap2d_matrix(self, size):
        self._check(lib().cgx_generate_lap2d_matrix(self._h, int(size)))

    def read_matrix(self, filename):
        self._check(lib().cgx_read_matrix(self._h, os.fsencode(filename)))

    def set_matrix_dense(self, A):
        A = np.ascontiguousarray(A, dtype=np.float64)
        assert A.ndim == 2 and A.shape[0] == A.shape[1]
        self._check(lib().cgx_set_matrix_dense(self._h, _dp(A), A.shape[1], A.shape[0]))

    def init_source_term(self, h):
        self._check(lib().cgx_init_source_term(self._h, float(h)))

    def set_source_term(self, b):
        b = np.ascontiguousarray(b, dtype=np.float64)
        self._check(lib().cgx_set_source_term(self._h, _dp(b)))

    def set_max_iter(self, max_iter):
        self._check(lib().cgx_set_max_iter(self._h, int(max_iter)))

    def tolerance(self, tol):
        self._check(lib().cgx_set_tolerance(self._h, float(tol)))

    def _size(self):
        m = C.c_int()
        n = C.c_int()
        self._check(lib().cgx_get_size(self._h, C.byref(m), C.byref(n)))
        return m.value, n.value

    def m(self):
        return self._size()[0]

    def n(self):
        return self._size()[1]

    def solve(self, x):
        """x: float64 array of length n, initial guess in / solution out.  Returns the result dict."""
        assert x.dtype == np.float64 and x.flags["C_CONTIGUOUS"] and x.size == self.n()
        res = Result()
        self._check(lib().cgx_solve(self._h, _dp(x), C.byref(res)))
        return res.as_dict()

    # -- stepping interface used by bench.py ----------------------------------------------------------
    def solve_begin(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        self._check(lib().cgx_solve_begin(self._h, _dp(x0)))

    def solve_steps(self, nsteps):
        done = C.c_int()
        self._check(lib().cgx_solve_steps(self._h, int(nsteps), C.byref(done)))
        return bool(done.value)

    def solve_end(self, x=None):
        res = Result()
        xp = _dp(x) if x is not None else None
        self._check(lib().cgx_solve_end(self._h, xp, C.byref(res)))
        return res.as_dict()

    def gemv_samples(self):
        """K1 durations (ms) of the most recent solve_steps call, launch order (needs profile_gemv)."""
        cnt = C.c_int()
        self._check(lib().cgx_get_gemv_samples(self._h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, dtype=np.float64)
        if cnt.value:
            self._check(lib().cgx_get_gemv_samples(self._h, _dp(out), cnt.value, C.byref(cnt)))
        return out

    def update_samples(self):
        """Durations (ms) of the event-timed update kernels (K3 / K3 with the exchange inside) of the most recent solve_steps
        call, launch order (needs profile_gemv and profile_update)."""
        cnt = C.c_int()
        self._check(lib().cgx_get_update_samples(self._h, None, 0, C.byref(cnt)))
        out = np.zeros(cnt.value, dtype=np.float64)
        if cnt.value:
            self._check(lib().cgx_get_update_samples(self._h, _dp(out), cnt.value, C.byref(cnt)))
        return out

    # -- kernel probes -------------------------------------------------------------------------------
    def probe_gemv(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        y = np.zeros(self.n(), dtype=np.float64)
        pap = C.c_double()
        self._check(lib().cgx_probe_gemv(self._h, _dp(p), _dp(y), C.byref(pap)))
        return y, pap.value

    def probe_time_gemv(self, reps=20):
        ms = C.c_double()
        self._check(lib().cgx_probe_time_gemv(self._h, int(reps), C.byref(ms)))
        return ms.value

    def probe_vector_ops(self, alpha, beta, x, r, p, Ap):
        x, r, p = (np.array(v, dtype=np.float64, copy=True) for v in (x, r, p))
        Ap = np.ascontiguousarray(Ap, dtype=np.float64)
        rr = C.c_double()
        self._check(lib().cgx_probe_vector_ops(self._h, x.size, alpha, beta, _dp(x), _dp(r), _dp(p), _dp(Ap),
                                               C.byref(rr)))
        return x, r, p, rr.value

    def _set_fault_after(self, calls):
        """Error-path tests: the HIP call after `calls` more of this context fails (-1 = off)."""
        lib().cgx_probe_set_fault_after(self._h, int(calls))

    def _mailbox_to_host(self):
        """Test hook: the mailbox of a one-rank P2P context moves to pinned coherent host memory (exchange over PCIe)."""
        self._check(lib().cgx_probe_p2p_mailbox_to_host(self._h))

    def _host_mailboxes(self, prefix, stage):
        """Test hook: every rank's mailbox in POSIX shared host memory (stage 0: create own; barrier; stage 1: map the peers')."""
        self._check(lib().cgx_probe_p2p_host_mailboxes(self._h, prefix.encode(), int(stage)))

    def _set_p2p_epoch(self, chan, value):
        """Test hook: move the epoch counter of a mailbox channel forward (same call on every rank, between two solves)."""
        self._check(lib().cgx_probe_set_p2p_epoch(self._h, int(chan), int(value)))

    def _p2p_epoch(self, chan):
        v = C.c_ulonglong()
        self._check(lib().cgx_probe_get_p2p_epoch(self._h, int(chan), C.byref(v)))
        return v.value

    def _resident_test(self, epoch=0, mute_workgroup=-1):
        """Test hook of the LDS-resident solver: move its epoch counter forward / make one workgroup of the next launch skip a publish."""
        self._check(lib().cgx_probe_resident_test(self._h, int(epoch), int(mute_workgroup)))

    def _set_resident_limit(self, workgroups):
        """Test hook: bound of co-resident workgroups the fused P2P update may assume (0 = ask the runtime)."""
        lib().cgx_probe_set_resident_limit(self._h, int(workgroups))

    def probe_fill_matrix_hash(self, seed, symmetric=False, diag=0.0):
        """Test probe: the dense row blocks of the current problem overwritten on the device with the counter-based hash matrix
        (every element a different number in [-1, 1); the tests rebuild any row on the host from the same definition)."""
        self._check(lib().cgx_probe_fill_matrix_hash(self._h, int(seed), 1 if symmetric else 0, float(diag)))

    def probe_source_term(self, local_shard=0):
        """The device copy of b (n doubles) of a local shard."""
        b = np.zeros(self.n(), dtype=np.float64)
        self._check(lib().cgx_probe_get_source_term(self._h, int(local_shard), _dp(b)))
        return b

    def probe_matrix_rows(self, local_shard=0):
        row0 = C.c_int()
        rows = C.c_int()
        self._check(lib().cgx_probe_get_matrix_rows(self._h, local_shard, None, C.byref(row0), C.byref(rows)))
        A = np.zeros((rows.value, self.n()), dtype=np.float64)
        if rows.value:
            self._check(lib().cgx_probe_get_matrix_rows(self._h, local_shard, _dp(A), C.byref(row0), C.byref(rows)))
        return A, row0.value
