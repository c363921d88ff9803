/*
 * cg_oracle.c -- CPU restatement of the reference CG path.  TEST INFRASTRUCTURE ONLY
 * (see cg_oracle.h for the rules and the pinning status).
 *
 * Citations are file:line under /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "cg_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* code/MPI/cg.cc:8 */
static const double NEARZERO = 1.0e-14;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- code/MPI/cg.cc:236-268 : partition_matrix --------------------------------------- */
/* The oracle computes under the default floating-point environment whatever the host process has done to the calling
 * thread's MXCSR (a library loaded with -ffast-math start-up code sets flush-to-zero; anything may change the rounding
 * mode): round to nearest, no FTZ / DAZ, exceptions masked.  fp_enter returns the caller's MXCSR, fp_leave puts it back.
 * (Threads OpenMP creates start with the default anyway.) */
#if defined(__x86_64__) || defined(__i386__)
static unsigned fp_enter(void)
{
    unsigned saved = 0, dflt = 0x1f80;
    __asm__ volatile("stmxcsr %0" : "=m"(saved));
    __asm__ volatile("ldmxcsr %0" : : "m"(dflt));
    return saved;
}
static void fp_leave(unsigned saved) { __asm__ volatile("ldmxcsr %0" : : "m"(saved)); }
#else   /* any other host: the C99 environment calls; the "state" reported is then 0x1f80 iff it is the default one */
#include <fenv.h>
static fenv_t g_saved_env;
static unsigned fp_enter(void) { fegetenv(&g_saved_env); fesetenv(FE_DFL_ENV); return 0; }
static void fp_leave(unsigned saved) { (void)saved; fesetenv(&g_saved_env); }
#endif

void oracle_partition(int N, int psize, int *start_rows, int *num_rows)
{
    if (psize == 1) {               /* cg.cc:248-252 */
        start_rows[0] = 0;
        num_rows[0] = N;
        return;
    }
    int n_loc = N / psize;          /* cg.cc:255 : floor, remainder goes to the last rank */
    int i0 = 0;
    for (int r = 0; r < psize - 1; ++r) {   /* cg.cc:256-264 */
        start_rows[r] = i0;
        num_rows[r] = n_loc;
        i0 += n_loc;
    }
    start_rows[psize - 1] = i0;     /* cg.cc:265-266 */
    num_rows[psize - 1] = N - i0;
}

/* ---- code/MPI/cg.cc:159-188 : generate_lap2d_matrix ---------------------------------- */
void oracle_generate_lap2d_rows(int size, int row0, int nrows, double *A)
{
    int inc = (int)floor(sqrt((double)size));     /* cg.cc:175 */
    for (int li = 0; li < nrows; ++li) {
        int i = row0 + li;
        double *row = A + (size_t)li * (size_t)size;
        memset(row, 0, (size_t)size * sizeof(double));              /* cg.cc:178-180 */
        if (i > inc) row[i - 1 - inc] = -1.0;                       /* cg.cc:181 */
        if (i > 0) row[i - 1] = -1.0;                               /* cg.cc:182 */
        row[i] = 4.0;                                               /* cg.cc:183 */
        if (i < size - 1) row[i + 1] = -1.0;                        /* cg.cc:184 */
        if (i < size - 1 - inc) row[i + 1 + inc] = -1.0;            /* cg.cc:185 */
    }
}

/* ---- code/MPI/cg.cc:218-234 : init_source_term --------------------------------------- */
void oracle_init_source_term(int n, double h, double *b)
{
    for (int i = 0; i < n; i++) {
        /* same expression, same left-to-right evaluation order as cg.cc:230-231 */
        b[i] = -2. * i * M_PI * M_PI * sin(10. * M_PI * i * h) * sin(10. * M_PI * i * h);
    }
}

/* ---- BLAS restatements (call sites cg.cc:80,82,91,101,105,110,113,116,128) ----------- */

/* One row, four interleaved partial sums (lane l takes columns j = l mod 4), combined as
 * (s0+s1)+(s2+s3).  Fixed order => identical results on every x86-64 host (-ffp-contract=off). */
__attribute__((target_clones("avx2", "default")))
static void gemv_rows4(int n, const double *a0, const double *a1, const double *a2, const double *a3,
                       const double *x, double *y4)
{
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0}, s3[4] = {0, 0, 0, 0};
    int j = 0;
    for (; j + 4 <= n; j += 4) {
        for (int l = 0; l < 4; ++l) {
            double xv = x[j + l];
            s0[l] += a0[j + l] * xv;
            s1[l] += a1[j + l] * xv;
            s2[l] += a2[j + l] * xv;
            s3[l] += a3[j + l] * xv;
        }
    }
    for (int l = 0; j < n; ++j, ++l) {
        double xv = x[j];
        s0[l] += a0[j] * xv;
        s1[l] += a1[j] * xv;
        s2[l] += a2[j] * xv;
        s3[l] += a3[j] * xv;
    }
    y4[0] = (s0[0] + s0[1]) + (s0[2] + s0[3]);
    y4[1] = (s1[0] + s1[1]) + (s1[2] + s1[3]);
    y4[2] = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    y4[3] = (s3[0] + s3[1]) + (s3[2] + s3[3]);
}

void oracle_gemv(int m, int n, const double *A, long lda, const double *x, double *y)
{
    int i = 0;
    for (; i + 4 <= m; i += 4) {
        const double *a = A + (size_t)i * (size_t)lda;
        gemv_rows4(n, a, a + lda, a + 2 * lda, a + 3 * lda, x, y + i);
    }
    for (; i < m; ++i) {
        const double *a = A + (size_t)i * (size_t)lda;
        double t[4];
        gemv_rows4(n, a, a, a, a, x, t);
        y[i] = t[0];
    }
}

double oracle_dot(int n, const double *x, const double *y)
{
    double s[4] = {0, 0, 0, 0};
    int j = 0;
    for (; j + 4 <= n; j += 4)
        for (int l = 0; l < 4; ++l) s[l] += x[j + l] * y[j + l];
    for (int l = 0; j < n; ++j, ++l) s[l] += x[j] * y[j];
    return (s[0] + s[1]) + (s[2] + s[3]);
}

void oracle_axpy(int n, double a, const double *x, double *y)
{
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
}

/* ---- code/MPI/cg.cc:38-156 : CGSolver::solve ------------------------------------------ */

typedef struct {
    const double *A;   /* row block, ld = n; NULL = the generate_lap2d rule applied on the fly (banded twin) */
    int start, count;
    double *r, *x, *Ap, *p, *tmp;   /* *_sub vectors, cg.cc:71-75 */
    double part;                    /* this rank's contribution to the pending reduction */
} rank_state;

/* Ap_sub = A_sub * v for one rank (cblas_dgemv, cg.cc:80-81,101-102,146-147).  With no stored block the five
 * entries of row i are taken from the generator's rule (cg.cc:181-185) in ascending column order: the same
 * matrix, never materialised -- used only where a dense n x n block cannot exist (oracle_solve_lap2d_banded). */
static void block_matvec(const rank_state *s, int n, const double *v)
{
    if (s->A) {
        oracle_gemv(s->count, n, s->A, n, v, s->Ap);
        return;
    }
    const int inc = (int)floor(sqrt((double)n));                   /* cg.cc:175 */
    for (int li = 0; li < s->count; ++li) {
        const int i = s->start + li;
        double acc = 0.0;
        if (i > inc) acc += -1.0 * v[i - 1 - inc];                  /* cg.cc:181 */
        if (i > 0) acc += -1.0 * v[i - 1];                          /* cg.cc:182 */
        acc += 4.0 * v[i];                                          /* cg.cc:183 */
        if (i < n - 1) acc += -1.0 * v[i + 1];                      /* cg.cc:184 */
        if (i < n - 1 - inc) acc += -1.0 * v[i + 1 + inc];          /* cg.cc:185 */
        s->Ap[li] = acc;
    }
}

/* Row blocks may be worked on by several host threads (the reference's MPI ranks, cg.run: srun -n P): every
 * loop over ranks below is embarrassingly parallel; the reductions over ranks stay sequential, in rank order,
 * so the result does not depend on the thread count. */
static int g_threads = 1;

static int solve_blocks(rank_state *rk, int psize, const double *b, double *x, int n, int max_iter,
                        double tol, oracle_result *res)
{
    double t_solve0 = now_s();
    double *p = (double *)malloc((size_t)n * sizeof(double));   /* replicated p, cg.cc:57 */
    if (!p) return -1;

    /* r_sub = b[rows], x_sub = x[rows]; r_sub -= A_sub * x  (cg.cc:71-82) */
    for (int q = 0; q < psize; ++q) {
        rank_state *s = &rk[q];
        memcpy(s->r, b + s->start, (size_t)s->count * sizeof(double));
        memcpy(s->x, x + s->start, (size_t)s->count * sizeof(double));
        block_matvec(s, n, x);
        oracle_axpy(s->count, -1.0, s->Ap, s->r);
        memcpy(s->p, s->r, (size_t)s->count * sizeof(double));            /* cg.cc:85 */
        memcpy(p + s->start, s->p, (size_t)s->count * sizeof(double));    /* Allgatherv, cg.cc:87-88 */
    }
    /* rsold = sum over ranks of r_sub . p_sub  (cg.cc:91-92) */
    double rsold = 0.0;
    for (int q = 0; q < psize; ++q) rsold += oracle_dot(rk[q].count, rk[q].r, rk[q].p);

    double rsnew = rsold;
    int converged = 0;
    double t_loop0 = now_s();
    int k = 0;
    for (; k < max_iter; ++k) {                                           /* cg.cc:96 */
        double conj = 0.0;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
        for (int q = 0; q < psize; ++q) {
            rank_state *s = &rk[q];
            block_matvec(s, n, p);                                        /* cg.cc:100-102 */
            s->part = oracle_dot(s->count, s->p, s->Ap);                  /* cg.cc:105 */
        }
        for (int q = 0; q < psize; ++q) conj += rk[q].part;              /* MPI_Allreduce, cg.cc:106 */
        double safe = rsold * NEARZERO;
        double alpha = rsold / ((conj < safe) ? safe : conj);             /* cg.cc:107: std::max(a, b) is (a < b) ? b : a, so a NaN conj stays NaN */
        rsnew = 0.0;
        for (int q = 0; q < psize; ++q) {
            rank_state *s = &rk[q];
            oracle_axpy(s->count, alpha, s->p, s->x);                     /* cg.cc:110 */
            oracle_axpy(s->count, -alpha, s->Ap, s->r);                   /* cg.cc:113 */
            rsnew += oracle_dot(s->count, s->r, s->r);                    /* cg.cc:116-117 */
        }
        if (sqrt(rsnew) < tol) {                                          /* cg.cc:120-121 */
            converged = 1;
            break;
        }
        double beta = rsnew / rsold;                                      /* cg.cc:124 */
        for (int q = 0; q < psize; ++q) {
            rank_state *s = &rk[q];
            memcpy(s->tmp, s->r, (size_t)s->count * sizeof(double));      /* cg.cc:127 */
            oracle_axpy(s->count, beta, s->p, s->tmp);                    /* cg.cc:128 */
            memcpy(s->p, s->tmp, (size_t)s->count * sizeof(double));      /* cg.cc:129 */
            memcpy(p + s->start, s->p, (size_t)s->count * sizeof(double));/* cg.cc:135-136 */
        }
        rsold = rsnew;                                                    /* cg.cc:132 */
    }
    double t_loop1 = now_s();

    for (int q = 0; q < psize; ++q)                                       /* Gatherv, cg.cc:140-142 */
        memcpy(x + rk[q].start, rk[q].x, (size_t)rk[q].count * sizeof(double));

    /* DEBUG block, cg.cc:144-154 (inside the reference's timing window) */
    double rr = 0.0;
    for (int q = 0; q < psize; ++q) {
        rank_state *s = &rk[q];
        block_matvec(s, n, x);
        for (int i = 0; i < s->count; ++i) {
            double d = s->Ap[i] - b[s->start + i];
            s->Ap[i] = d;
        }
        rr += oracle_dot(s->count, s->Ap, s->Ap);
    }
    double bb = oracle_dot(n, b, b);
    double xx = oracle_dot(n, x, x);
    double t_solve1 = now_s();

    if (res) {
        res->iterations = k;
        res->converged = converged;
        res->residual_prev = sqrt(rsold);
        res->residual_last = sqrt(rsnew);
        res->x_norm = sqrt(xx);
        res->rel_residual = sqrt(rr) / sqrt(bb);
        res->seconds_loop = t_loop1 - t_loop0;
        res->seconds_solve = t_solve1 - t_solve0;
    }
    free(p);
    return 0;
}

static int alloc_rank_vectors(rank_state *s)
{
    size_t c = (size_t)(s->count > 0 ? s->count : 1);
    s->r = (double *)malloc(c * sizeof(double));
    s->x = (double *)malloc(c * sizeof(double));
    s->Ap = (double *)malloc(c * sizeof(double));
    s->p = (double *)malloc(c * sizeof(double));
    s->tmp = (double *)malloc(c * sizeof(double));
    return (s->r && s->x && s->Ap && s->p && s->tmp) ? 0 : -1;
}

static void free_rank_vectors(rank_state *s)
{
    free(s->r); free(s->x); free(s->Ap); free(s->p); free(s->tmp);
}

static int oracle_solve_impl(const double *A, const double *b, double *x, int n, int max_iter, double tol,
                 int psize, oracle_result *res)
{
    if (!A || !b || !x || n <= 0 || psize <= 0) return -2;
    int *start = (int *)malloc(sizeof(int) * (size_t)psize);
    int *num = (int *)malloc(sizeof(int) * (size_t)psize);
    rank_state *rk = (rank_state *)calloc((size_t)psize, sizeof(rank_state));
    if (!start || !num || !rk) return -1;
    oracle_partition(n, psize, start, num);                               /* cg.cc:59-64 */
    int rc = 0;
    for (int q = 0; q < psize; ++q) {
        rk[q].A = A + (size_t)start[q] * (size_t)n;                       /* cg.cc:80 pointer offset */
        rk[q].start = start[q];
        rk[q].count = num[q];
        if (alloc_rank_vectors(&rk[q])) rc = -1;
    }
    if (!rc) rc = solve_blocks(rk, psize, b, x, n, max_iter, tol, res);
    for (int q = 0; q < psize; ++q) free_rank_vectors(&rk[q]);
    free(rk); free(start); free(num);
    return rc;
}

static int oracle_solve_lap2d_impl(int n, int max_iter, double tol, int psize, double *x, oracle_result *res)
{
    if (!x || n <= 0 || psize <= 0) return -2;
    int *start = (int *)malloc(sizeof(int) * (size_t)psize);
    int *num = (int *)malloc(sizeof(int) * (size_t)psize);
    rank_state *rk = (rank_state *)calloc((size_t)psize, sizeof(rank_state));
    double *b = (double *)malloc((size_t)n * sizeof(double));
    double **blocks = (double **)calloc((size_t)psize, sizeof(double *));
    if (!start || !num || !rk || !b || !blocks) return -1;
    oracle_partition(n, psize, start, num);
    oracle_init_source_term(n, 1. / n, b);                                /* cg_main.cc:45-46 */
    int rc = 0;
#pragma omp parallel for num_threads(g_threads) schedule(static) if (g_threads > 1)
    for (int q = 0; q < psize; ++q) {
        blocks[q] = (double *)malloc((size_t)(num[q] > 0 ? num[q] : 1) * (size_t)n * sizeof(double));
        if (!blocks[q]) { rc = -1; continue; }
        oracle_generate_lap2d_rows(n, start[q], num[q], blocks[q]);
        rk[q].A = blocks[q];
        rk[q].start = start[q];
        rk[q].count = num[q];
        if (alloc_rank_vectors(&rk[q])) rc = -1;
    }
    if (!rc) rc = solve_blocks(rk, psize, b, x, n, max_iter, tol, res);
    for (int q = 0; q < psize; ++q) { free_rank_vectors(&rk[q]); free(blocks[q]); }
    free(blocks); free(b); free(rk); free(start); free(num);
    return rc;
}

static int oracle_solve_lap2d_banded_impl(int n, int max_iter, double tol, int psize, double *x, oracle_result *res)
{
    if (!x || n <= 0 || psize <= 0) return -2;
    int *start = (int *)malloc(sizeof(int) * (size_t)psize);
    int *num = (int *)malloc(sizeof(int) * (size_t)psize);
    rank_state *rk = (rank_state *)calloc((size_t)psize, sizeof(rank_state));
    double *b = (double *)malloc((size_t)n * sizeof(double));
    if (!start || !num || !rk || !b) return -1;
    oracle_partition(n, psize, start, num);
    oracle_init_source_term(n, 1. / n, b);                                /* cg_main.cc:45-46 */
    int rc = 0;
    for (int q = 0; q < psize; ++q) {
        rk[q].A = NULL;                                                   /* rule applied on the fly */
        rk[q].start = start[q];
        rk[q].count = num[q];
        if (alloc_rank_vectors(&rk[q])) rc = -1;
    }
    if (!rc) rc = solve_blocks(rk, psize, b, x, n, max_iter, tol, res);
    for (int q = 0; q < psize; ++q) free_rank_vectors(&rk[q]);
    free(b); free(rk); free(start); free(num);
    return rc;
}

/* ---- code/MPI/matrix_coo.cc:7-60 + matrix.cc:6-22 : Matrix-Market coordinate reader --- */
int oracle_read_mtx_dense(const char *path, int *m_out, int *n_out, int *nz_out, int *sym_out, double **A_out)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;                                 /* matrix_coo.cc:14-17 */
    char line[1100];
    if (!fgets(line, sizeof line, f)) { fclose(f); return -2; }
    /* banner: %%MatrixMarket object format field symmetry, tokens lower-cased (mmio.c:96-179) */
    char banner[64], obj[64], fmt[64], field[64], symm[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, obj, fmt, field, symm) != 5) { fclose(f); return -2; }
    for (char *c = obj; *c; ++c) *c = (char)tolower((unsigned char)*c);
    for (char *c = fmt; *c; ++c) *c = (char)tolower((unsigned char)*c);
    for (char *c = symm; *c; ++c) *c = (char)tolower((unsigned char)*c);
    if (strcmp(banner, "%%MatrixMarket") != 0) { fclose(f); return -2; }
    if (strcmp(obj, "matrix") != 0 || strcmp(fmt, "coordinate") != 0) { fclose(f); return -3; } /* matrix_coo.cc:25-29 */
    int is_sym = strcmp(symm, "symmetric") == 0;       /* matrix_coo.cc:43 */
    /* size line, skipping % comments (mmio.c:198-206) */
    int m = 0, n = 0, nz = 0;
    for (;;) {
        if (!fgets(line, sizeof line, f)) { fclose(f); return -4; }
        if (line[0] == '%') continue;
        if (sscanf(line, "%d %d %d", &m, &n, &nz) == 3) break;
    }
    double *A = (double *)calloc((size_t)m * (size_t)n, sizeof(double));   /* Matrix::resize zero-fills, matrix.hh:11-15 */
    if (!A) { fclose(f); return -5; }
    for (int z = 0; z < nz; ++z) {
        int I, J; double a;
        if (fscanf(f, "%d %d %lg\n", &I, &J, &a) != 3) { free(A); fclose(f); return -6; }  /* matrix_coo.cc:48 */
        I--; J--;                                                     /* matrix_coo.cc:49-50 */
        A[(size_t)I * (size_t)n + (size_t)J] = a;                     /* matrix.cc:17 */
        if (is_sym) A[(size_t)J * (size_t)n + (size_t)I] = a;         /* matrix.cc:18-20 */
    }
    fclose(f);
    *m_out = m; *n_out = n; *nz_out = nz; *sym_out = is_sym; *A_out = A;
    return 0;
}

void oracle_set_threads(int nthreads) { g_threads = nthreads > 0 ? nthreads : 1; }

/* The floating-point environment of the calling thread: MXCSR (rounding mode bits 13-14, flush-to-zero bit 15,
 * denormals-are-zero bit 6).  The oracle's results are only reproducible under the default 0x1f80 (round to nearest, no
 * FTZ / DAZ); a test harness can check that nothing in the process has changed it. */
unsigned oracle_fp_state(void)
{
#if defined(__x86_64__) || defined(__i386__)
    unsigned csr = 0;
    __asm__ volatile("stmxcsr %0" : "=m"(csr));
    return csr;
#else
    return fegetround() == FE_TONEAREST ? 0x1f80u : 0u;
#endif
}

/* ---- dense, incompressible test data: NOT a reference function ------------------------------------------------
 * Restates csrc/cgx_kernels.h hash_entry (the device fill behind cgx_probe_fill_matrix_hash) so that the checker can
 * rebuild any row of the hash matrix on the host: A(i,j) = (double)(mix64(mix64(seed) ^ (a << 32 | b)) >> 11) * 2^-52 - 1
 * with (a,b) = (i,j), or (min,max) when symmetric; A(i,i) = diag when diag != 0.  mix64 is the splitmix64 finaliser.
 * Every step is exact in fp64, so this and the device agree bit for bit.  (The reference's generator, cg.cc:178-186,
 * leaves 5 non-zeros per row; its GEMV, cg.cc:101-102, is a general dense dgemv -- this is the data that exercises it.) */
static unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_hash_rows(int n, long row0, long nrows, unsigned long long seed, int symmetric, double diag, double *A)
{
    const unsigned long long sm = mix64(seed);
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long li = 0; li < nrows; ++li) {
        const long i = row0 + li;
        double *row = A + (size_t)li * (size_t)n;
        for (long j = 0; j < n; ++j) {
            if (i == j && diag != 0.0) { row[j] = diag; continue; }
            const unsigned long long a = (symmetric && j < i) ? (unsigned long long)j : (unsigned long long)i;
            const unsigned long long b = (symmetric && j < i) ? (unsigned long long)i : (unsigned long long)j;
            const unsigned long long h = mix64(sm ^ ((a << 32) | b));
            row[j] = (double)(h >> 11) * 0x1.0p-52 - 1.0;
        }
    }
}

/* ---- bench.py cpu_baseline leg: time `reps` GEMV passes over `nrows` generated rows -------- */
double oracle_time_gemv_rows(int n, int nrows, int reps)
{
    double *A = (double *)malloc((size_t)nrows * (size_t)n * sizeof(double));
    double *p = (double *)malloc((size_t)n * sizeof(double));
    double *y = (double *)malloc((size_t)nrows * sizeof(double));
    if (!A || !p || !y) { free(A); free(p); free(y); return -1.0; }
    oracle_generate_lap2d_rows(n, 0, nrows, A);
    oracle_init_source_term(n, 1. / n, p);
    oracle_gemv(nrows, n, A, n, p, y);   /* warm */
    double t0 = now_s();
    for (int r = 0; r < reps; ++r) oracle_gemv(nrows, n, A, n, p, y);
    double t1 = now_s();
    free(A); free(p); free(y);
    return t1 - t0;
}

int oracle_solve(const double *A, const double *b, double *x, int n, int max_iter, double tol,
                 int psize, oracle_result *res)
{
    const unsigned fp = fp_enter();
    const int rc = oracle_solve_impl(A, b, x, n, max_iter, tol, psize, res);
    fp_leave(fp);
    return rc;
}

int oracle_solve_lap2d(int n, int max_iter, double tol, int psize, double *x, oracle_result *res)
{
    const unsigned fp = fp_enter();
    const int rc = oracle_solve_lap2d_impl(n, max_iter, tol, psize, x, res);
    fp_leave(fp);
    return rc;
}

int oracle_solve_lap2d_banded(int n, int max_iter, double tol, int psize, double *x, oracle_result *res)
{
    const unsigned fp = fp_enter();
    const int rc = oracle_solve_lap2d_banded_impl(n, max_iter, tol, psize, x, res);
    fp_leave(fp);
    return rc;
}
