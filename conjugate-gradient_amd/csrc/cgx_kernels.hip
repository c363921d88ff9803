// cgx_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the dense fp64 CG hot path.
//
// The path restated here is CGSolver::solve's loop body, code/MPI/cg.cc:96-137 (reference file:line):
//   K1 gemv       cblas_dgemv  cg.cc:100-102   + fused cblas_ddot(p_sub, Ap_sub) cg.cc:105
//   K2 reduce     the local half of MPI_Allreduce, cg.cc:106,117
//   K3 update_xr  alpha cg.cc:107, two cblas_daxpy cg.cc:110,113, cblas_ddot(r,r) cg.cc:116
//   K4 update_p   convergence test cg.cc:120-121, beta cg.cc:124, p = r + beta p cg.cc:127-129, rsold = rsnew cg.cc:132
// None of it is derived from code/CUDA/cg.cu: that file uses thread-per-row-chunk kernels with
// atomicAdd; these are streaming kernels without atomics, deterministic for a fixed launch shape.
//
// Everything is HBM-bound (0.25 flop/byte): no MFMA.  What matters is 16 B/lane coalesced loads of
// A's rows, enough independent loads in flight per CU, p served from L2/LDS instead of HBM, and
// keeping every scalar on the device.
#include "cgx_kernels.h"

namespace cgx {

typedef double d2 __attribute__((ext_vector_type(2)));

static constexpr double kNearZero = 1.0e-14;   // NEARZERO, code/MPI/cg.cc:8

// ------------------------------------------------------------------------------------------------
// reductions: fixed order => bitwise reproducible for a given launch shape
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;   // every lane holds the total
}

template <int WAVES>
__device__ __forceinline__ double block_sum(double v, double *lds /* >= WAVES doubles */)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();   // protect lds against a previous use
    if (lane == 0) lds[w] = v;
    __syncthreads();
    double s = lds[0];
#pragma unroll
    for (int i = 1; i < WAVES; ++i) s += lds[i];
    return s;
}

template <bool NT>
__device__ __forceinline__ d2 load_a(const double *ptr)
{
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(ptr));
    else return *reinterpret_cast<const d2 *>(ptr);
}

// ------------------------------------------------------------------------------------------------
// K1, variant 1: column-split.  One workgroup owns R consecutive rows; its WAVES waves split the
// columns in 1 KiB pieces (lane = 16 B), so every global_load_dwordx4 of A is a fully coalesced
// 1 KiB wave access and the workgroup sweeps WAVES KiB of each row per step.  p is loaded once per
// step (16 B/lane, L2 hit) and reused from registers by all R rows: p traffic = 1/R of A traffic.
// U steps are issued back to back: R*U independent 1 KiB loads in flight per wave.
// Epilogue: DPP/shuffle wave reduction, LDS cross-wave combine in fixed wave order, Ap store, and the
// fused p.Ap partial of the workgroup's rows (cg.cc:105).
// ------------------------------------------------------------------------------------------------
template <int R, int U, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_gemv_colsplit(const double *__restrict__ A, long lda, int rows,
                                                               const double *__restrict__ p,
                                                               const double *__restrict__ p_local,
                                                               double *__restrict__ Ap, double *__restrict__ partials,
                                                               const int *__restrict__ done)
{
    if (done && *done) return;   // converged earlier: the whole grid drains immediately
    __shared__ double red[WAVES][R];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long row0 = (long)blockIdx.x * R;
    const int ncols = (int)lda;   // pad columns hold zeros in A and in p

    const double *a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long row = row0 + r;
        if (row > rows - 1) row = rows - 1;   // tail workgroup: re-read the last row, result discarded
        a[r] = A + row * lda;
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }

    constexpr int kStep = WAVES * 128;   // doubles swept by the workgroup per step
    int c = w * 128 + lane * 2;
    for (; c + (U - 1) * kStep < ncols; c += U * kStep) {
        d2 pv[U];
        d2 av[U][R];
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = *reinterpret_cast<const d2 *>(p + c + u * kStep);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) av[u][r] = load_a<NT>(a[r] + c + u * kStep);
        // Keep all R*U+U loads in flight: without this fence hipcc's occupancy-driven scheduler
        // re-serialises them as load / s_waitcnt vmcnt(0) / fma pairs (measured in the .s).
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc0[r] = fma(av[u][r].x, pv[u].x, acc0[r]);
                acc1[r] = fma(av[u][r].y, pv[u].y, acc1[r]);
            }
    }
    for (; c < ncols; c += kStep) {   // remaining single steps (lda is even, so c+1 < lda)
        d2 pv = *reinterpret_cast<const d2 *>(p + c);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            d2 av = load_a<NT>(a[r] + c);
            acc0[r] = fma(av.x, pv.x, acc0[r]);
            acc1[r] = fma(av.y, pv.y, acc1[r]);
        }
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
        double s = wave_sum(acc0[r] + acc1[r]);
        if (lane == 0) red[w][r] = s;
    }
    __syncthreads();
    if (w == 0) {
        double d = 0.0;
        if (lane < R) {
            double s = red[0][lane];
#pragma unroll
            for (int i = 1; i < WAVES; ++i) s += red[i][lane];
            const long row = row0 + lane;
            if (row < rows) {
                Ap[row] = s;
                d = p_local[row] * s;
            }
        }
        d = wave_sum(d);
        if (lane == 0) partials[blockIdx.x] = d;
    }
}

// ------------------------------------------------------------------------------------------------
// K1, variant 2: row-split with LDS-staged p.  Each of the WAVES waves owns R rows (the workgroup
// WAVES*R rows) and all waves sweep the same columns, so the p tile (TILE doubles) is fetched from
// L2 once per workgroup into LDS (double buffered, one barrier per tile) and read back with
// conflict-free ds_read_b128 (lane = 16 B).  p traffic from L2 = 1/(WAVES*R) of A traffic.  No cross-wave
// combine: each wave finishes its own rows with a shuffle reduction.
// ------------------------------------------------------------------------------------------------
template <int R, int U, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void k_gemv_ldsp(const double *__restrict__ A, long lda, int rows,
                                                           const double *__restrict__ p,
                                                           const double *__restrict__ p_local,
                                                           double *__restrict__ Ap, double *__restrict__ partials,
                                                           const int *__restrict__ done)
{
    if (done && *done) return;
    constexpr int TILE = 2048;                       // doubles of p per LDS buffer (16 KiB)
    constexpr int kThreads = WAVES * 64;
    constexpr int kPerThread = TILE / 2 / kThreads;  // 16-B pieces each thread stages per tile
    static_assert(TILE % (2 * kThreads) == 0, "tile must split evenly");
    __shared__ __attribute__((aligned(16))) double ptile[2][TILE];
    __shared__ double red[WAVES];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const long row0 = ((long)blockIdx.x * WAVES + w) * R;
    const int ncols = (int)lda;

    const double *a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long row = row0 + r;
        if (row > rows - 1) row = rows - 1;
        a[r] = A + row * lda;
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }

    const int ntiles = (ncols + TILE - 1) / TILE;
    d2 stage[kPerThread];
    // prologue: tile 0 -> LDS buffer 0
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        int c = (i * kThreads + threadIdx.x) * 2;
        stage[i] = (c < ncols) ? *reinterpret_cast<const d2 *>(p + c) : d2{0.0, 0.0};
    }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i)
        *reinterpret_cast<d2 *>(&ptile[0][(i * kThreads + threadIdx.x) * 2]) = stage[i];
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const int base = t * TILE;
        // issue the next tile's p loads early; they land in registers while this tile streams A
        if (t + 1 < ntiles) {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i) {
                int c = base + TILE + (i * kThreads + threadIdx.x) * 2;
                stage[i] = (c < ncols) ? *reinterpret_cast<const d2 *>(p + c) : d2{0.0, 0.0};
            }
        }
        const int cend = (base + TILE < ncols) ? TILE : (ncols - base);   // valid doubles in this tile (even)
        int c = lane * 2;
        for (; c + (U - 1) * 128 < cend; c += U * 128) {
            d2 pv[U];
            d2 av[U][R];
#pragma unroll
            for (int u = 0; u < U; ++u) pv[u] = *reinterpret_cast<const d2 *>(&ptile[buf][c + u * 128]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) av[u][r] = load_a<NT>(a[r] + base + c + u * 128);
            __builtin_amdgcn_sched_barrier(0);   // all loads issued before the first fma (see variant 1)
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc0[r] = fma(av[u][r].x, pv[u].x, acc0[r]);
                    acc1[r] = fma(av[u][r].y, pv[u].y, acc1[r]);
                }
        }
        for (; c < cend; c += 128) {
            d2 pv = *reinterpret_cast<const d2 *>(&ptile[buf][c]);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                d2 av = load_a<NT>(a[r] + base + c);
                acc0[r] = fma(av.x, pv.x, acc0[r]);
                acc1[r] = fma(av.y, pv.y, acc1[r]);
            }
        }
        if (t + 1 < ntiles) {
#pragma unroll
            for (int i = 0; i < kPerThread; ++i)
                *reinterpret_cast<d2 *>(&ptile[buf ^ 1][(i * kThreads + threadIdx.x) * 2]) = stage[i];
        }
        __syncthreads();   // next buffer complete, current buffer free for tile t+2
    }

    double d = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double s = wave_sum(acc0[r] + acc1[r]);
        const long row = row0 + r;
        if (row < rows) {
            if (lane == 0) Ap[row] = s;
            d += p_local[row] * s;   // same value in every lane
        }
    }
    if (lane == 0) red[w] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = red[0];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) s += red[i];
        partials[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// K2: one workgroup folds the per-workgroup partials in a fixed order.
// ------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partials, int n,
                                                          double *__restrict__ out, const int *__restrict__ done)
{
    if (done && *done) return;
    __shared__ double lds[4];
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) s += partials[(long)i * NV + v];
        s = block_sum<4>(s, lds);
        if (threadIdx.x == 0) out[v] = s;
    }
}

// gathered layout on every shard: [rank q][slot v], kSlots doubles per rank
__device__ __forceinline__ double sum_ranks(const double *__restrict__ gathered, int slot, int nranks)
{
    double s = gathered[slot];
    for (int q = 1; q < nranks; ++q) s += gathered[q * kSlots + slot];   // rank order, same on every shard
    return s;
}

// ------------------------------------------------------------------------------------------------
// K3: x += alpha p ; r -= alpha Ap ; partial r.r          (cg.cc:107-116)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_xr(int count, const double *__restrict__ p_local,
                                                    const double *__restrict__ Ap, double *__restrict__ x,
                                                    double *__restrict__ r, const Scalars *__restrict__ sc, int parity,
                                                    const double *__restrict__ gathered, int nranks,
                                                    double *__restrict__ partials)
{
    if (sc->done) return;
    __shared__ double lds[4];
    const double rsold = sc->rs[parity];
    const double conj = sum_ranks(gathered, kSlotConj, nranks);                 // MPI_Allreduce, cg.cc:106
    const double alpha = rsold / fmax(conj, rsold * kNearZero);      // cg.cc:107
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rr = 0.0;
    if (i < count) {
        x[i] = fma(alpha, p_local[i], x[i]);                          // cg.cc:110
        const double rn = fma(-alpha, Ap[i], r[i]);                   // cg.cc:113
        r[i] = rn;
        rr = rn * rn;                                                 // cg.cc:116
    }
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = rr;
}

// ------------------------------------------------------------------------------------------------
// K4: convergence test, beta, p = r + beta p, rsold <- rsnew   (cg.cc:117-132)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_p(int count, const double *__restrict__ r, double *__restrict__ p_local,
                                                   Scalars *__restrict__ sc, int parity, int k, double tol,
                                                   const double *__restrict__ gathered, int nranks)
{
    if (sc->done) return;
    const double rsold = sc->rs[parity];
    const double rsnew = sum_ranks(gathered, kSlotRr, nranks);                // MPI_Allreduce, cg.cc:117
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0);
    if (first) sc->rs[parity ^ 1] = rsnew;                           // becomes rsold of iteration k+1, cg.cc:132
    if (sqrt(rsnew) < tol) {                                         // cg.cc:120-121: break before the p update
        if (first) { sc->k_final = k; sc->done = 1; }
        return;
    }
    const double beta = rsnew / rsold;                               // cg.cc:124
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < count) p_local[i] = fma(beta, p_local[i], r[i]);         // cg.cc:127-129
}

// ------------------------------------------------------------------------------------------------
// setup / verification kernels (outside the iteration loop)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_residual(int count, const double *__restrict__ b,
                                                        const double *__restrict__ Ap, double *__restrict__ r,
                                                        double *__restrict__ p_local, double *__restrict__ partials)
{
    __shared__ double lds[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double rr = 0.0;
    if (i < count) {
        const double rv = b[i] - Ap[i];        // r_sub = b_sub - A_sub x, cg.cc:79-82
        r[i] = rv;
        p_local[i] = rv;                       // p_sub = r_sub, cg.cc:85
        rr = rv * rv;                          // rsold = r.p with p == r, cg.cc:91
    }
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = rr;
}

__global__ void k_set_rsold(Scalars *sc, const double *__restrict__ gathered, int nranks)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sc->rs[0] = sum_ranks(gathered, kSlotRr, nranks);   // cg.cc:92
        sc->rs[1] = sc->rs[0];
        sc->done = 0;
        sc->k_final = 0;
    }
}

__global__ __launch_bounds__(256) void k_debug_norms(int count, const double *__restrict__ Ax,
                                                      const double *__restrict__ b, const double *__restrict__ x,
                                                      double *__restrict__ partials)
{
    __shared__ double lds[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double e = 0.0, bb = 0.0, xx = 0.0;
    if (i < count) {
        const double d = Ax[i] - b[i];     // cg.cc:146-148
        e = d * d;
        bb = b[i] * b[i];
        xx = x[i] * x[i];
    }
    e = block_sum<4>(e, lds);
    bb = block_sum<4>(bb, lds);
    xx = block_sum<4>(xx, lds);
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = e;
        partials[3 * blockIdx.x + 1] = bb;
        partials[3 * blockIdx.x + 2] = xx;
    }
}

// generate_lap2d_matrix, cg.cc:159-188.  One thread writes 16 B; rows are 16-B aligned (lda even).
__global__ __launch_bounds__(256) void k_generate_lap2d(double *__restrict__ A, long lda, int size, int row0, int rows,
                                                         int inc)
{
    const long pairs_per_row = lda / 2;
    const long total = (long)rows * pairs_per_row;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long lr = t / pairs_per_row;
        const int j0 = (int)(t - lr * pairs_per_row) * 2;
        const int i = row0 + (int)lr;
        d2 v;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = j0 + e;
            double val = 0.0;                                               // cg.cc:178-180
            if (j < size) {
                if (j == i) val = 4.0;                                      // cg.cc:183
                else if (i > 0 && j == i - 1) val = -1.0;                   // cg.cc:182
                else if (i < size - 1 && j == i + 1) val = -1.0;            // cg.cc:184
                else if (i > inc && j == i - 1 - inc) val = -1.0;           // cg.cc:181
                else if (i < size - 1 - inc && j == i + 1 + inc) val = -1.0;// cg.cc:185
            }
            v[e] = val;
        }
        *reinterpret_cast<d2 *>(A + lr * lda + j0) = v;
    }
}

__global__ __launch_bounds__(256) void k_scatter_coo(double *__restrict__ A, long lda, int row0,
                                                      const int *__restrict__ I, const int *__restrict__ J,
                                                      const double *__restrict__ a, long nz)
{
    for (long z = (long)blockIdx.x * 256 + threadIdx.x; z < nz; z += (long)gridDim.x * 256)
        A[(long)(I[z] - row0) * lda + J[z]] = a[z];    // matrix.cc:17 (duplicates resolved on the host)
}

__global__ void k_loopback_gather(double *const *gathered_ptrs, const Scalars *const *scalar_ptrs, int nshards)
{
    const int t = threadIdx.x;
    if (t < nshards * nshards * kSlots) {
        const int dst = t / (nshards * kSlots);
        const int rem = t - dst * nshards * kSlots;
        const int src = rem / kSlots, v = rem - src * kSlots;
        gathered_ptrs[dst][src * kSlots + v] = scalar_ptrs[src]->local[v];
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

GemvPlan plan_gemv(int variant, int rows, int ncols)
{
    (void)ncols;
    GemvPlan pl{};
    pl.waves = 4;
    if (variant <= 0) {
        // default: column-split, as many rows per workgroup as still leaves >= 4 workgroups per CU
        pl.variant = 1;
        pl.nt = 1;
        if (rows >= 8192) { pl.R = 8; pl.U = 2; }
        else if (rows >= 2048) { pl.R = 4; pl.U = 4; }
        else { pl.R = 2; pl.U = 4; }
    } else {
        // explicit shape: variant*10000 + R*100 + U*10 + nt   (e.g. 10821, 20441, 11611)
        pl.variant = variant / 10000;
        pl.R = (variant / 100) % 100;
        pl.U = (variant / 10) % 10;
        pl.nt = variant % 10;
    }
    if (pl.variant == 2) {
        pl.rows_per_wg = pl.R * pl.waves;
    } else {
        pl.variant = 1;
        pl.rows_per_wg = pl.R;
    }
    pl.grid = ceil_div(rows, pl.rows_per_wg);
    return pl;
}

template <int R, int U, bool NT>
static hipError_t launch_gemv_shape(const GemvPlan &pl, const double *A, long lda, int rows, const double *p_full,
                                    const double *p_local, double *Ap, double *partials, const int *done, hipStream_t s)
{
    if (pl.variant == 2)
        hipLaunchKernelGGL((k_gemv_ldsp<R, U, 4, NT>), dim3(pl.grid), dim3(256), 0, s, A, lda, rows, p_full, p_local, Ap,
                           partials, done);
    else
        hipLaunchKernelGGL((k_gemv_colsplit<R, U, 4, NT>), dim3(pl.grid), dim3(256), 0, s, A, lda, rows, p_full, p_local,
                           Ap, partials, done);
    return hipGetLastError();
}

hipError_t launch_gemv(const GemvPlan &pl, const double *A, long lda, int rows, const double *p_full,
                       const double *p_local, double *Ap, double *partials, const int *done, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
#define CGX_SHAPE(r, u)                                                                                      \
    if (pl.R == r && pl.U == u) {                                                                            \
        return pl.nt ? launch_gemv_shape<r, u, true>(pl, A, lda, rows, p_full, p_local, Ap, partials, done, s) \
                     : launch_gemv_shape<r, u, false>(pl, A, lda, rows, p_full, p_local, Ap, partials, done, s); \
    }
    CGX_SHAPE(8, 2)
    CGX_SHAPE(8, 1)
    CGX_SHAPE(4, 4)
    CGX_SHAPE(4, 2)
    CGX_SHAPE(2, 4)
    CGX_SHAPE(2, 8)
    CGX_SHAPE(1, 8)
    CGX_SHAPE(16, 1)
#undef CGX_SHAPE
    return hipErrorInvalidValue;
}

hipError_t launch_reduce_partials(const double *partials, int n, double *out, const int *done, hipStream_t s)
{
    hipLaunchKernelGGL((k_reduce_partials<1>), dim3(1), dim3(256), 0, s, partials, n, out, done);
    return hipGetLastError();
}

hipError_t launch_reduce_partials3(const double *partials, int n, double *out3, hipStream_t s)
{
    hipLaunchKernelGGL((k_reduce_partials<3>), dim3(1), dim3(256), 0, s, partials, n, out3, (const int *)nullptr);
    return hipGetLastError();
}

int update_xr_grid(int count) { return count > 0 ? ceil_div(count, 256) : 1; }

hipError_t launch_update_xr(int count, const double *p_local, const double *Ap, double *x, double *r,
                            const Scalars *sc, int parity, const double *gathered, int nranks, double *partials,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_update_xr, dim3(update_xr_grid(count)), dim3(256), 0, s, count, p_local, Ap, x, r, sc, parity,
                       gathered, nranks, partials);
    return hipGetLastError();
}

hipError_t launch_update_p(int count, const double *r, double *p_local, Scalars *sc, int parity, int k, double tol,
                           const double *gathered, int nranks, hipStream_t s)
{
    hipLaunchKernelGGL(k_update_p, dim3(update_xr_grid(count)), dim3(256), 0, s, count, r, p_local, sc, parity, k, tol,
                       gathered, nranks);
    return hipGetLastError();
}

hipError_t launch_init_residual(int count, const double *b, const double *Ap, double *r, double *p_local,
                                double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_residual, dim3(update_xr_grid(count)), dim3(256), 0, s, count, b, Ap, r, p_local,
                       partials);
    return hipGetLastError();
}

hipError_t launch_set_rsold(Scalars *sc, const double *gathered, int nranks, hipStream_t s)
{
    hipLaunchKernelGGL(k_set_rsold, dim3(1), dim3(64), 0, s, sc, gathered, nranks);
    return hipGetLastError();
}

hipError_t launch_debug_norms(int count, const double *Ax, const double *b, const double *x, double *partials,
                              hipStream_t s)
{
    hipLaunchKernelGGL(k_debug_norms, dim3(update_xr_grid(count)), dim3(256), 0, s, count, Ax, b, x, partials);
    return hipGetLastError();
}

hipError_t launch_generate_lap2d(double *A, long lda, int size, int row0, int rows, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    const int inc = (int)floor(sqrt((double)size));   // cg.cc:175
    long total = (long)rows * (lda / 2);
    int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_generate_lap2d, dim3(grid), dim3(256), 0, s, A, lda, size, row0, rows, inc);
    return hipGetLastError();
}

hipError_t launch_scatter_coo(double *A, long lda, int row0, const int *I, const int *J, const double *a, long nz,
                              hipStream_t s)
{
    if (nz <= 0) return hipSuccess;
    int grid = (int)((nz + 255) / 256 < 2048 ? (nz + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_scatter_coo, dim3(grid), dim3(256), 0, s, A, lda, row0, I, J, a, nz);
    return hipGetLastError();
}

hipError_t launch_loopback_gather(double *const *gathered_ptrs, const Scalars *const *scalar_ptrs, int nshards,
                                  hipStream_t s)
{
    int threads = nshards * nshards * kSlots;
    if (threads > 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_loopback_gather, dim3(1), dim3(threads < 64 ? 64 : threads), 0, s, gathered_ptrs, scalar_ptrs,
                       nshards);
    return hipGetLastError();
}

}  // namespace cgx
