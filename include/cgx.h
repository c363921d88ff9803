/*
 * cgx.h -- C ABI of libcgx, the MI355X (gfx950) drop-in for the reference's CGSolver::solve() path.
 *
 * The reference (federicobetti99/Conjugate-Gradient) has no plugin/FFI layer: its seam is the C++
 * class CGSolver (code/MPI/cg.hh:11-57, code/CUDA/cg.hh:13-45) called once from main
 * (code/MPI/cg_main.cc:28-55).  Each entry point below names the reference member it replaces.
 * A host-side `class CGSolver` with the reference's method names, built on this ABI, lives in
 * conjugate-gradient_amd/host/cg.hh; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in any signature; all pointers are HOST pointers;
 *   - every function returns a cgx_status (0 = ok) and never calls exit();
 *   - a context is not thread-safe: one caller thread (the reference is MPI_THREAD_SINGLE,
 *     code/MPI/cg_main.cc:15);
 *   - all floating point is IEEE fp64, indices are int like the reference (code/MPI/matrix.hh:17)
 *     but device offsets are 64-bit.
 *
 * Sharding (code/MPI/cg.cc:59-75, 236-268): the matrix is row-block partitioned over `nranks`
 * shards exactly as partition_matrix does.  CGX_COMM_RCCL = one OS process per GPU (the MPI model),
 * one RCCL AllGather per iteration over xGMI in place of 2 x MPI_Allreduce + MPI_Allgatherv.
 * CGX_COMM_LOOPBACK = `nranks` logical shards on ONE device in one process (same kernels, same
 * collective sequencing, in-process exchange) -- the CI stand-in for a multi-GPU node.
 */
#ifndef CGX_H
#define CGX_H

#ifdef __cplusplus
extern "C" {
#endif

#define CGX_VERSION 1
#define CGX_UNIQUE_ID_BYTES 128
#define CGX_IPC_HANDLE_BYTES 64

typedef enum cgx_status {
    CGX_OK = 0,
    CGX_ERR_BAD_ARG = 1,      /* null pointer, n <= 0, call out of sequence ...            */
    CGX_ERR_IO = 2,           /* file could not be opened / parsed (matrix_coo.cc:14-33)    */
    CGX_ERR_HIP = 3,          /* a HIP runtime call failed                                  */
    CGX_ERR_RCCL = 4,         /* an RCCL call failed or librccl could not be loaded         */
    CGX_ERR_OOM = 5,          /* host or device allocation failed                           */
    CGX_ERR_NO_DEVICE = 6,    /* no gfx950 device visible: the product path has NO CPU fallback */
    CGX_ERR_UNSUPPORTED = 7,  /* e.g. Matrix-Market field/format this reader does not take  */
    CGX_ERR_P2P = 8           /* direct peer exchange: a bounded wait for a peer expired, or IPC mapping failed */
} cgx_status;

typedef enum cgx_comm_mode {
    CGX_COMM_SELF = 0,        /* 1 shard, 1 device, no collectives (psize == 1)             */
    CGX_COMM_LOOPBACK = 1,    /* nranks logical shards on one device, in-process exchange   */
    CGX_COMM_RCCL = 2,        /* this process is shard `rank` of `nranks`, RCCL over xGMI   */
    CGX_COMM_P2P = 3          /* same process model, but the one exchange per iteration stores straight into the
                                 peers' IPC-mapped mailboxes over xGMI, folded into the update kernel (no RCCL at
                                 all); needs cgx_p2p_export / cgx_p2p_import after cgx_create; two forms of handing the
                                 bytes over: payload + flag words (default), or tagged words (cgx_config.p2p_tagged)   */
} cgx_comm_mode;

/* How the row block is held on the device.  DENSE is the reference's contract (Matrix, code/MPI/matrix.hh:7-29:
 * every one of the n*n entries is stored and streamed by the GEMV) and the only format bench.py measures.
 * BANDED is an OPT-IN fast path that is NOT in the reference: the block is held as its non-zero diagonals (at
 * most CGX_MAX_DIAGONALS of them) and K1 becomes a banded mat-vec -- same CG recurrence, same results to rounding,
 * n/ndiag times less matrix traffic (the reference's unused MatrixCOO::mat_vec,
 * code/MPI/matrix_coo.hh:22-34, is the hint).  A matrix with more diagonals is refused with
 * CGX_ERR_UNSUPPORTED, never silently densified. */
typedef enum cgx_matrix_format {
    CGX_MATRIX_DENSE = 0,
    CGX_MATRIX_BANDED = 1
} cgx_matrix_format;
#define CGX_MAX_DIAGONALS 64

typedef struct cgx_config {
    int  struct_version;      /* = CGX_VERSION                                              */
    int  comm_mode;           /* cgx_comm_mode                                              */
    int  device;              /* HIP device ordinal this process drives                     */
    int  rank;                /* CGX_COMM_RCCL: this process's shard; else 0                */
    int  nranks;              /* number of row blocks (the reference's psize)               */
    unsigned char unique_id[CGX_UNIQUE_ID_BYTES]; /* CGX_COMM_RCCL: from cgx_comm_unique_id */
    int  gemv_variant;        /* 0 = library default: the per-launch path (K1 + K3 per iteration) with the K1 shape chosen from the
                                 block size -- or, for a dense matrix of n <= 16384 on one GPU (CGX_COMM_SELF), a PERSISTENT kernel:
                                 the whole loop cg.cc:95-137 as ONE launch whose workgroups exchange Ap among themselves.
                                 n <= 4096: every row group of A stays on the chip, in a CU's LDS (n <= 2048) or in its LDS and
                                 registers with a streamed rest (csrc/cgx_resident.hip; 2-6 us per iteration instead of 7-26);
                                 4096 < n <= 16384: the rows are streamed (all but the few that fit beside them), the vectors
                                 stay in registers (csrc/cgx_stream.hip; the default up to n = 10000, where it measures faster:
                                 18 / 71 / 92-95 us per iteration at n = 5120 / 8192 / 9216 instead of 36 / 80 / 104, the solve
                                 at n = 10000 in 71-74 ms instead of 78; CGX_STREAM_MAX
                                 moves that end).  What the default choice rests on, and what happens when it fails:
                                 all workgroups of such a kernel must be resident at once (checked against the runtime's
                                 occupancy when the problem is set; another tenant of the GPU can still break it), and the
                                 exchange rests on an aligned 8-byte half of a 16-byte write-through store being seen untorn
                                 by another CU -- observed on gfx950 / ROCm 7.2, not an architectural guarantee (the same
                                 caveat as p2p_tagged below).  Every wait inside the kernel is bounded (p2p_timeout_ms); a
                                 launch whose wait expires has written nothing of the solver's state, and under the DEFAULT
                                 choice the library redoes it on the per-launch path and stays there for the rest of the
                                 problem: the call returns CGX_OK, cgx_get_gemv_plan then reports the per-launch shape,
                                 cgx_get_resident_record counts the event and cgx_last_error holds a note.
                                 40000 = ask for the persistent kernel: CGX_ERR_UNSUPPORTED where it cannot be had, and an
                                 expired wait is CGX_ERR_HIP (no fallback; the context stays usable for the next problem);
                                 50000 = the same, but the STREAMING persistent kernel whatever the size (1024 <= n <= 16384);
                                 -1 = the per-launch path with its default shape, also where a persistent kernel would fit
                                 (as does CGX_RESIDENT=0 in the environment; CGX_STREAM_MAX=n moves the upper end of the
                                 default, 4096 = never stream); v*10000 + R*100 + U*10 + d = an explicit per-launch K1 shape,
                                 see DESIGN.md "K1 variants".  profile_gemv, profile_update and check_every do not apply to
                                 a persistent kernel (there are no K1 launches to time and no polls: gemv_ms_* stay 0). */
    int  lda_pad;             /* extra doubles added to the row pitch (-1 = library default)*/
    int  check_every;         /* iterations between host polls of the device `done` flag (0 = default) */
    int  profile_gemv;        /* n > 0 = bracket every n-th K1 launch with HIP events (at most 2048 per cgx_solve_steps call);
                                 the FIRST launch of a cgx_solve_steps call starts on a drained stream and is never
                                 a sample unless profile_first is set (its event pair also spans the host's launch latency) */
    int  profile_update;      /* 1 = the update kernel of an iteration (K3, or K3 with the exchange inside) is event-timed as well,
                                 on the same launches as K1 (needs profile_gemv > 0): on a multi-GPU run its duration holds the
                                 wait for the peers, i.e. the cost of the exchange (cgx_get_update_samples).  Default 0: a
                                 timed dispatch costs ~5 us of stream time.  (The field was reserved0 before round 3.) */
    int  p2p_mailbox_kib;     /* CGX_COMM_P2P: mailbox size in KiB (0 = 16384)               */
    int  p2p_timeout_ms;      /* CGX_COMM_P2P: bound of every in-kernel wait (0 = 5000)     */
    int  p2p_separate_exchange; /* CGX_COMM_P2P: 1 = exchange in its own kernel between K1 and K3 (default 0: folded into K3) */
    int  matrix_format;       /* cgx_matrix_format; 0 = dense = the reference's storage     */
    int  profile_first;       /* 1 = also sample the first K1 launch of every cgx_solve_steps call (diagnostics only) */
    int  profile_markers;     /* 0 = the event pair is bound to the K1 dispatch itself (kernel begin/end, what rocprofv3 reports);
                                 1 = hipEventRecord markers before and after it (diagnostics only: adds two packets per launch) */
    int  p2p_no_acquire_fence; /* diagnostics only: 1 = leave out the system-scope acquire fence behind the flag wait of the
                                 fused update kernel (for measuring its cost); default 0 = fence on */
    int  p2p_tagged;          /* CGX_COMM_P2P with the exchange folded into the update kernel: 1 = hand the bytes over as tagged
                                 8-byte words (each carries half a double and the epoch and validates itself: no flags, no
                                 fences; the two words of a double leave as one 16-byte write-through store, readers poll with
                                 relaxed 8-byte system-scope atomic loads); 0 = payload stores + release,
                                 flag words, polls + acquire.  Both forms are checked by cgx_p2p_selftest with their own device
                                 code.  The tagged form rests on an aligned 8-byte half of a 16-byte write-through store arriving
                                 untorn at the peer, which no run on more than one GPU has shown yet: the launchers' `auto` uses
                                 the flag form and this one on request only.  (Was reserved[0].) */
} cgx_config;

typedef struct cgx_result {
    int    iterations;        /* k at loop exit, code/MPI/cg.cc:95-137 (index of converging iteration, or max_iter) */
    int    converged;         /* the break at cg.cc:120-121 was taken                       */
    double residual_prev;     /* sqrt(rsold): the "residual" the DEBUG line prints (cg.cc:152-153) */
    double residual_last;     /* sqrt(rsnew) of the last executed iteration                 */
    double x_norm;            /* ||x||             (cg.cc:151)                              */
    double rel_residual;      /* ||Ax-b|| / ||b||  (cg.cc:145-150)                          */
    double seconds_solve;     /* reference timing window: all of solve() (cg_main.cc:53-55) */
    double seconds_loop;      /* the k-loop only                                            */
    double gemv_ms_avg;       /* mean K1 launch duration (HIP events) over the most recent cgx_solve_steps call, 0 if not profiled */
    double gemv_ms_min;
    long long gemv_launches;  /* K1 launches that were event-timed (= samples behind avg / min / median / max) */
    double gemv_bytes;        /* algorithmic bytes of ONE K1 launch on this shard: 8*(rows*n + n + rows); banded: 8*(rows*ndiag + 2*rows) */
    double gemv_ms_median;    /* median of the samples (SURVEY.md section 8d asks for the median) */
    double gemv_ms_max;
    long long gemv_discarded; /* launches deliberately not sampled although profiling was on: the first one after a drained stream */
    double steps_device_ms;   /* the most recent cgx_solve_steps call on the DEVICE's clock: from a marker in front of its first
                                 kernel to one behind its last (0 if profiling is off); the host's wall clock around the same
                                 call additionally holds launch latency and the final synchronisation */
} cgx_result;

typedef struct cgx_ctx cgx_ctx;

/* ---- life cycle -------------------------------------------------------------------------- */
void        cgx_config_init(cgx_config *cfg);                 /* fills defaults (SELF, device 0, 1 rank) */
cgx_status  cgx_comm_unique_id(unsigned char out[CGX_UNIQUE_ID_BYTES]);  /* ncclGetUniqueId; rank 0 calls it, the launcher broadcasts it (replaces MPI_Init, cg_main.cc:15-20) */
cgx_status  cgx_create(cgx_ctx **out, const cgx_config *cfg);
cgx_status  cgx_destroy(cgx_ctx *ctx);
const char *cgx_last_error(const cgx_ctx *ctx);               /* ctx may be NULL: error of the last failed cgx_create on this thread */
const char *cgx_status_string(cgx_status s);

/* What the transport of this context actually spans, for the benchmark record (replaces MPI_Comm_size / MPI_Comm_rank,
 * cg.cc:50-51, as a CHECK: the values come from the transport, not from the configuration):
 *   *ranks_wired  RCCL: ncclCommCount of the communicator; P2P: mailboxes mapped (own + peers) once imported;
 *                 LOOPBACK: logical shards; SELF: 1
 *   *rank_seen    RCCL: ncclCommUserRank; otherwise the configured rank
 *   device_id     PCI bus id of the device this context drives ("0000:05:00.0"), so that a launcher can count the
 *                 distinct GPUs behind its ranks; at least 32 bytes, may be NULL. */
cgx_status  cgx_get_comm_info(cgx_ctx *ctx, int *comm_mode, int *ranks_wired, int *rank_seen, char *device_id);

/* The K1 (GEMV, cg.cc:100-102) launch shape the library planned for local shard `local_shard` of the current problem,
 * for the benchmark record (which kernel ran): out = {variant (1 column-split, 2 LDS-staged p tiles, 3 banded, 4 = the loop
 * runs in the resident persistent kernel, 5 = in the streaming persistent kernel), R rows per workgroup (variant 2: per
 * wave), U steps in flight (variant 4 / 5: column steps of 512 / 1024), waves per workgroup, light (1 = the one-round form;
 * variant 4: rows of a workgroup held in registers, 0 up to n = 2048; variant 5: rows per batch of the ring), split (column
 * pieces per row group, tied to the XCDs; variant 5: rows of a workgroup that stay in LDS and registers instead of being streamed), grid (workgroups of one fused launch; variant 4 / 5: of the persistent kernel, all
 * resident at once), ncols (columns swept)}.  After a persistent launch has been redone on the per-launch path (gemv_variant
 * above) this reports the per-launch shape. */
#define CGX_GEMV_PLAN_INTS 8
cgx_status  cgx_get_gemv_plan(const cgx_ctx *ctx, int local_shard, int out[CGX_GEMV_PLAN_INTS]);

/* What the waits inside the persistent launches of the current (or most recent) solve cost, so that a slow solve can say why:
 * out = {[0] iterations run, [1] polls of the watched word that had to be repeated, [2] gather rounds that had to be repeated,
 * [3] launches -- all four as workgroup 0 saw them --, [4] workgroup 0: 100-MHz wall-clock ticks from its publish of a
 * launch's FIRST iteration until it had all of Ap (a workgroup that was placed late shows here; largest over the launches),
 * [5] workgroup 0: the longest such span of a LATER iteration in which a poll had to be repeated (the exchange itself; 0 =
 * never), [6] / [7] the same two, largest over ALL workgroups, [8] persistent launches of this context, over its whole life,
 * whose waits expired and that were redone on the per-launch path, [9] 1 = the current problem is on a persistent kernel}.
 * Costs no synchronisation: the record comes back with {done, k_final} behind every launch. */
#define CGX_RESIDENT_RECORD_INTS 10
cgx_status  cgx_get_resident_record(const cgx_ctx *ctx, long long out[CGX_RESIDENT_RECORD_INTS]);

/* ---- CGX_COMM_P2P wire-up (replaces MPI_Init's job for the direct-xGMI transport) --------- */
/* export: this rank's mailbox as an IPC handle.  import: all ranks' handles, rank order (nranks * 64 bytes),
 * gathered by the launcher (torch.distributed, MPI_Allgather, pipes ...).  selftest: `rounds` all-gathers of
 * a known pattern, verified on every rank; *ok = 0 on any mismatch or expired wait (then use CGX_COMM_RCCL).
 * The launcher must agree on `ok` across ranks (all-reduce MIN) before anything else is exchanged: every rank must take the
 * same decision.  (That agreement is not what makes the re-layout of the mailbox for a problem safe: a passing self-test
 * ends with an exchange on the scalar channel, after which every peer has finished reading its self-test slots.) */
cgx_status  cgx_p2p_export(cgx_ctx *ctx, unsigned char out[CGX_IPC_HANDLE_BYTES]);
cgx_status  cgx_p2p_import(cgx_ctx *ctx, const unsigned char *handles);
cgx_status  cgx_p2p_selftest(cgx_ctx *ctx, int rounds, int *ok);

/* ---- CGSolver::partition_matrix, code/MPI/cg.cc:236-268 (pure host function) ------------- */
cgx_status  cgx_partition(int n, int psize, int *start_rows, int *num_rows);

/* ---- problem definition ------------------------------------------------------------------ */
/* CGSolver::generate_lap2d_matrix(size), cg.cc:159-188: builds this shard's row block ON DEVICE,
 * sets m = n = max_iter = size (cg.cc:170-172). */
cgx_status  cgx_generate_lap2d_matrix(cgx_ctx *ctx, int size);
/* CGSolver::read_matrix + Matrix::read (cg.cu:307-321, matrix.cc:6-22): caller hands the dense
 * row-major n x n matrix (host, leading dimension lda doubles); the library copies this shard's
 * rows to the device.  Sets m = n, max_iter = n (code/CUDA/cg.cu:236 loops to m_n). */
cgx_status  cgx_set_matrix_dense(cgx_ctx *ctx, const double *A, long lda, int n);
/* MatrixCOO::read + Matrix::read (matrix_coo.cc:7-60, matrix.cc:6-22) done by the library:
 * Matrix-Market `matrix coordinate {real,integer,double} {general,symmetric}` -> device row block,
 * without a dense n*n host staging copy.  Entries are assigned on the device in the file's order (a later
 * entry for the same element wins, a symmetric entry's mirror follows it), exactly as the sequential loop. */
cgx_status  cgx_read_matrix(cgx_ctx *ctx, const char *mtx_path);
/* CGSolver::init_source_term(h), cg.cc:218-234: b evaluated on the HOST with libm sin so it is
 * bit-identical to the reference's, then this shard's slice is uploaded. */
cgx_status  cgx_init_source_term(cgx_ctx *ctx, double h);
cgx_status  cgx_set_source_term(cgx_ctx *ctx, const double *b /* n doubles */);
/* CGSolver::set_max_iter (cg.cc:204-216) and CGSolver::tolerance (cg.hh:39) */
cgx_status  cgx_set_max_iter(cgx_ctx *ctx, int max_iter);
cgx_status  cgx_set_tolerance(cgx_ctx *ctx, double tol);
cgx_status  cgx_get_size(const cgx_ctx *ctx, int *m, int *n);   /* CGSolver::m(), n() */
/* Storage of local shard `local_shard`: *format = cgx_matrix_format; banded: *ndiag and offsets[0..*ndiag)
 * (column minus row, ascending; room for CGX_MAX_DIAGONALS ints or NULL); *matrix_bytes = device bytes of the block. */
cgx_status  cgx_get_matrix_format(const cgx_ctx *ctx, int local_shard, int *format, int *ndiag, int *offsets,
                                  double *matrix_bytes);

/* ---- CGSolver::solve, cg.cc:38-156 ------------------------------------------------------- */
/* x: n doubles, in = initial guess (cg_main.cc:49-50 passes zeros), out = solution (every rank
 * gets the full x; the reference fills rank 0 only, cg.cc:140-142).  res may be NULL. */
cgx_status  cgx_solve(cgx_ctx *ctx, double *x, cgx_result *res);

/* The same path cut at its seams, for bench.py's warmup / timed-steps contract:
 *   begin  = cg.cc:49-92 (setup, initial residual GEMV, p0 gather, rsold)
 *   steps  = `nsteps` bodies of the loop cg.cc:96-137 (stops early on convergence); synchronises
 *   end    = cg.cc:140-154 (gather x, DEBUG verification) and fills res. */
cgx_status  cgx_solve_begin(cgx_ctx *ctx, const double *x0);
cgx_status  cgx_solve_steps(cgx_ctx *ctx, int nsteps, int *done_out);
cgx_status  cgx_solve_end(cgx_ctx *ctx, double *x, cgx_result *res);
/* The individual K1 durations (ms, HIP events on the library's stream) of the most recent cgx_solve_steps call, in
 * launch order: at most `cap` are written, *count receives how many exist.  bench.py reports their median. */
cgx_status  cgx_get_gemv_samples(cgx_ctx *ctx, double *ms_out, int cap, int *count);

/* The same for the update kernel (cfg.profile_update): K3 `k_update_xr`, or, under CGX_COMM_P2P with the exchange folded in,
 * `k_update_xr_p2p`, whose duration includes the bounded wait for every peer's chunks (cg.cc:106,135-136). */
cgx_status  cgx_get_update_samples(cgx_ctx *ctx, double *ms_out, int cap, int *count);

/* ---- kernel probes (parity tests of the individual hot ops through the C ABI) ------------- */
/* Ap = A_shard * p  (K1; cblas_dgemv at cg.cc:101-102) for every local shard; y receives the n
 * results in global row order (LOOPBACK/SELF) or this rank's rows at their global offset (RCCL);
 * *pAp receives sum_i p_i * Ap_i over the local rows (the fused cblas_ddot of cg.cc:105). */
cgx_status  cgx_probe_gemv(cgx_ctx *ctx, const double *p, double *y, double *pAp);
/* Mean duration (ms, HIP events) of `reps` back-to-back launches of the plain K1 on the current matrix. */
cgx_status  cgx_probe_time_gemv(cgx_ctx *ctx, int reps, double *ms_per_launch);
/* One pass of K3 (cg.cc:107-116) and of the p update fused into the next K1 (cg.cc:124-129) on caller
 * data of length n, single shard: x += alpha p; r -= alpha Ap; *rr = r.r; p = r + beta p. */
cgx_status  cgx_probe_vector_ops(cgx_ctx *ctx, int n, double alpha, double beta, double *x, double *r,
                                 double *p, const double *Ap, double *rr);
/* Error-path testing: after `calls` further HIP runtime calls of this context the next one is not made and reports a
 * failure instead (-1 = off).  TEST-ONLY: tools/leak_check.py and tests/ use it to walk every early return of a probe,
 * the self-test and the solve; nothing else (no environment variable) arms it. */
cgx_status  cgx_probe_set_fault_after(cgx_ctx *ctx, int calls);
/* The Matrix-Market parser of cgx_read_matrix by itself, HOST ONLY (no context, no device): header checks as
 * matrix_coo.cc:19-40, then the nz entries "%d %d %lg" (matrix_coo.cc:44-55) parsed from the mapped file on `threads` host
 * threads (0 = the library's default; negative = exactly that many, however small the file) into 0-based I, J and a in
 * file order; at most `cap` entries are written (call with cap = 0 for the sizes).  err (may be NULL) receives the message. */
cgx_status  cgx_probe_parse_matrix_market(const char *path, int threads, int *m, int *n, int *nz, int *symmetric, int *I, int *J,
                                          double *a, long cap, char *err, int err_cap);
/* TEST ONLY: moves the mailbox of a ONE-rank CGX_COMM_P2P context (no problem set yet) into pinned, coherent host memory, so
 * that every store, poll and load of the exchange crosses PCIe: the system-scope path outside this GPU's HBM and L2. */
cgx_status  cgx_probe_p2p_mailbox_to_host(cgx_ctx *ctx);
/* TEST ONLY: the mailboxes of a multi-rank CGX_COMM_P2P job in POSIX shared HOST memory (segments <prefix>_<rank>), so that
 * the stores, polls and loads of EVERY rank cross PCIe between separate processes -- the closest a one-GPU box offers to
 * peers whose memory is remote.  Replaces cgx_p2p_export / cgx_p2p_import: stage 0 creates this rank's segment and makes it
 * the mailbox; after a launcher barrier stage 1 maps every peer's.  Before any problem is set. */
cgx_status  cgx_probe_p2p_host_mailboxes(cgx_ctx *ctx, const char *prefix, int stage);
/* Test hook for the co-residency guard of CGX_COMM_P2P's fused update kernel (its workgroups wait for each other inside the
 * kernel, so its grid must not exceed what the device keeps resident: occupancy x CUs, queried from the runtime when a
 * problem is set): workgroups > 0 replaces the queried bound for the problems set afterwards, 0 restores it.  The same bound
 * guards the LDS-resident solver (its workgroups wait for each other as well): with a bound below its grid the default falls
 * back to the per-launch path and gemv_variant 40000 is refused. */
cgx_status  cgx_probe_set_resident_limit(cgx_ctx *ctx, int workgroups);
/* TEST ONLY, host arithmetic (no device, no context): the shape the library would give a persistent kernel for a dense n x n
 * problem on a GPU of `cus` compute units with `lds_per_cu` bytes of LDS a workgroup may use -- streaming = 0: the resident
 * kernel (n <= 4096), 1: the streaming kernel (1024 <= n <= 16384).  out = {fits (0 / 1), R rows per workgroup, S column steps,
 * grid, exchange slots per parity, LDS bytes per workgroup, rows in LDS, rows in registers, rows per batch of the ring
 * (streaming), streamed rows read with the default cache policy (streaming), hybrid (resident: 1 = rows in LDS + registers + a
 * streamed rest), threads per workgroup}.  tests/test_persistent_plans.py holds the invariants for every n. */
#define CGX_PERSISTENT_PLAN_INTS 12
cgx_status  cgx_probe_persistent_plan(int n, int cus, long lds_per_cu, int streaming, long out[CGX_PERSISTENT_PLAN_INTS]);
/* TEST ONLY: the epoch counter of mailbox channel `chan` (0 = plain segment all-gathers of a tagged-word context, 1 = the
 * iteration's exchange, 2 = DEBUG scalars).  set: move it FORWARD to `value` (the next exchange is value + 1) so that tests
 * reach the wrap of the tagged form's 32-bit tag and of the epoch's low 32 bits without 4e9 exchanges; every rank makes the
 * same call between two solves.  A smaller value is refused (flag words only grow). */
cgx_status  cgx_probe_set_p2p_epoch(cgx_ctx *ctx, int chan, unsigned long long value);
cgx_status  cgx_probe_get_p2p_epoch(cgx_ctx *ctx, int chan, unsigned long long *value);
/* TEST ONLY, the resident solver (gemv_variant 0 / 40000, n <= 4096 on one GPU).  epoch > 0: move the epoch counter of
 * its tagged-word exchange FORWARD to `epoch` (the next iteration uses epoch + 1), so that tests reach the wrap of the 32-bit
 * tag without 4e9 iterations; a smaller value is refused.  mute_workgroup >= 0: that workgroup of the NEXT launch leaves out the
 * publish of its first iteration, so that every wait for it expires (the bounded-wait path: cgx_solve_steps then returns
 * CGX_ERR_HIP after p2p_timeout_ms); -1 = none.  The mute disarms itself after one launch. */
cgx_status  cgx_probe_resident_test(cgx_ctx *ctx, unsigned long long epoch, int mute_workgroup);
/* Copy the device-resident source term of local shard `local_shard` (n doubles, what cgx_init_source_term /
 * cgx_set_source_term left in HBM) back to the host: the bit-exact check of cg.cc:230-231. */
cgx_status  cgx_probe_get_source_term(cgx_ctx *ctx, int local_shard, double *b_out);
/* TEST ONLY: dense, incompressible data for K1 at the BASELINE sizes.  The reference's generator leaves 5 non-zeros per row
 * (cg.cc:178-186) while its GEMV is a general dense dgemv (cg.cc:101-102): this call overwrites the dense row block of every
 * local shard of the CURRENT problem (set one first: it defines n, the partition and the pitch) on the device with
 * A(i,j) = (double)(mix64(mix64(seed) ^ (i << 32 | j)) >> 11) * 2^-52 - 1 in [-1, 1), mix64 = the splitmix64 finaliser
 * (csrc/cgx_kernels.h hash_entry; the parity tests restate it so that a checker rebuilds any row on the host).
 * symmetric != 0: (i,j) and (j,i) share the value of (min, max).  diag != 0: A(i,i) = diag. */
cgx_status  cgx_probe_fill_matrix_hash(cgx_ctx *ctx, unsigned long long seed, int symmetric, double diag);
/* Copy this shard's device row block (rows x n, dense, row-major) back to the host (banded storage is expanded). */
cgx_status  cgx_probe_get_matrix_rows(cgx_ctx *ctx, int local_shard, double *A_out, int *row0, int *rows);

#ifdef __cplusplus
}
#endif
#endif /* CGX_H */
