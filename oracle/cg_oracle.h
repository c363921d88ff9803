/*
 * cg_oracle.h -- CPU restatement of the reference's dense fp64 conjugate-gradient path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it.  Nothing under conjugate-gradient_amd/ links,
 * imports or executes it; the product path fails loudly when the HIP library is missing.
 *
 * Each function cites the reference file:line (relative to /root/reference) it restates.
 *
 * Pinning status: the reference has no tests and no golden vectors, and its solver (cg.cc) needs
 * <cblas.h> + a CBLAS library that this image does not ship, so it cannot be built into oracle/_ref
 * under the "no stand-in headers" rule (only its Matrix-Market reader can: `make -C oracle ref`).  The oracle is pinned against the reference outputs the
 * survey stage recorded from the compiled reference (committed as
 * tests/golden/reference_probe.json).  Those values are not regenerable from this repo.
 *
 * Third-party arithmetic: GEMV/dot/axpy live in OpenBLAS (unpinned, `module load openblas`,
 * code/MPI/cg.run:6; 0.3.10 per figures/gprof.png).  Its internal summation order is not part
 * of the reference; this restatement uses the published BLAS definitions (y = A x row by row,
 * dot = sum_i x_i y_i, axpy y += a x) and parity is stated as a tolerance, not bitwise.  tests/test_oracle.py drives the same
 * recurrence through the OpenBLAS 0.3.29 that numpy / scipy bundle and holds this restatement to it (x 1e-12, residual 1e-10).
 */
#ifndef CG_ORACLE_H
#define CG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_result {
    int    iterations;      /* k at loop exit (code/MPI/cg.cc:95-137): index of the converging iteration, or max_iter */
    int    converged;       /* 1 if the break at cg.cc:120-121 was taken */
    double residual_prev;   /* sqrt(rsold) -- what the DEBUG line prints as "residual" (cg.cc:152-153) */
    double residual_last;   /* sqrt(rsnew) of the last executed iteration */
    double x_norm;          /* ||x||  (cg.cc:151) */
    double rel_residual;    /* ||Ax-b|| / ||b||  (cg.cc:145-150) */
    double seconds_loop;    /* wall time of the k-loop only */
    double seconds_solve;   /* wall time of the whole solve(), reference timing window (cg_main.cc:53-55) */
} oracle_result;

/* code/MPI/cg.cc:236-268 */
void oracle_partition(int N, int psize, int *start_rows, int *num_rows);

/* code/MPI/cg.cc:159-188, rows [row0,row0+nrows) of the size x size matrix, row-major, ld = size */
void oracle_generate_lap2d_rows(int size, int row0, int nrows, double *A);

/* code/MPI/cg.cc:218-234 with h = 1./n as in cg_main.cc:45-46 */
void oracle_init_source_term(int n, double h, double *b);

/* code/MPI/cg.cc:38-156 with psize logical ranks run in-process (rank order reductions).
 * A: n x n row-major; b: n; x: n (in: initial guess, out: solution). */
int oracle_solve(const double *A, const double *b, double *x, int n, int max_iter,
                 double tol, int psize, oracle_result *res);

/* Host threads that work on the row blocks of the psize logical ranks (default 1 = the serial path).  Results do
 * not depend on it: reductions over ranks are always done sequentially in rank order. */
void oracle_set_threads(int nthreads);
unsigned oracle_fp_state(void);   /* MXCSR of the calling thread (0x1f80 = default: round to nearest, no FTZ/DAZ) */

/* Same recurrence, but A is never materialised as one block by the caller: the oracle
 * allocates psize row blocks itself with oracle_generate_lap2d_rows (used for large N). */
int oracle_solve_lap2d(int n, int max_iter, double tol, int psize, double *x, oracle_result *res);

/* The same solve with the generator's rule (cg.cc:181-185) applied on the fly instead of a stored n x n block:
 * the checker for libcgx's opt-in banded storage at sizes where the dense block cannot exist.  It is validated
 * against oracle_solve_lap2d at small n (tests/test_oracle.py); it is not a path of the reference. */
int oracle_solve_lap2d_banded(int n, int max_iter, double tol, int psize, double *x, oracle_result *res);

/* y = A[0:m, 0:n] * x, row-major, lda -- the cblas_dgemv call at cg.cc:101-102 */
void oracle_gemv(int m, int n, const double *A, long lda, const double *x, double *y);
double oracle_dot(int n, const double *x, const double *y);            /* cg.cc:105,116 */
void oracle_axpy(int n, double a, const double *x, double *y);         /* cg.cc:110,113,128 */

/* code/MPI/matrix_coo.cc:7-60 + matrix.cc:6-22: Matrix-Market coordinate -> dense row-major.
 * On success *A_out is malloc'ed (m*n doubles, caller frees).  Returns 0 or a negative code. */
int oracle_read_mtx_dense(const char *path, int *m, int *n, int *nz_stored, int *is_sym, double **A_out);

/* NOT a reference function: rows [row0, row0+nrows) of the n x n counter-based hash matrix that the device fills through
 * cgx_probe_fill_matrix_hash (csrc/cgx_kernels.h hash_entry), row-major, ld = n -- dense, incompressible test data for K1.
 * Bit-identical to the device fill.  Uses the threads of oracle_set_threads. */
void oracle_hash_rows(int n, long row0, long nrows, unsigned long long seed, int symmetric, double diag, double *A);

/* One timed pass of the GEMV over a row block: used by bench.py's cpu_baseline leg. */
double oracle_time_gemv_rows(int n, int nrows, int reps);

#ifdef __cplusplus
}
#endif
#endif
