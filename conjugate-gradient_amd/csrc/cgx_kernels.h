// cgx_kernels.h -- launch interface of the CDNA4 (gfx950) kernels of the CG hot path.
// Host code in cgx_solver.cpp sees only these plain functions; all device code is in cgx_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace cgx {

// Device-resident scalar block of one shard.  Nothing in the iteration loop is read back by the
// host except `done` (polled every check_every iterations).
struct Scalars {
    double rs[2];        // rsold / rsnew ping-pong: iteration k reads rs[k&1], writes rs[(k+1)&1]   (cg.cc:91,116,132)
    double local[4];     // this shard's contributions to the reductions (all-gather send buffer, kSlots doubles)
    double dbg[4];       // global ||Ax-b||^2, ||b||^2, ||x||^2 (cg.cc:145-151)
    int    done;         // set by K4 when sqrt(rsnew) < tol (cg.cc:120-121); later kernels exit at once
    int    k_final;      // k of the converging iteration
    int    pad[2];
};

// Scalar exchange: every shard all-gathers its `local[kSlots]`; consumers sum slot v over ranks in rank
// order (bit-identical on every shard, which is what makes all ranks take the same break, cg.cc:117-121).
constexpr int kSlots = 4;
constexpr int kSlotConj = 0;   // p.Ap partial              (cg.cc:105-106); DEBUG: ||Ax-b||^2
constexpr int kSlotRr = 1;     // r.r partial               (cg.cc:116-117, 91-92); DEBUG: ||b||^2
constexpr int kSlotX = 2;      // DEBUG: ||x||^2            (cg.cc:151)
constexpr int kMaxRanks = 64;  // gathered[] holds kMaxRanks*kSlots doubles

struct GemvPlan {
    int variant;     // 1 = column-split (p from L2 to registers), 2 = row-split (p tiles staged in LDS)
    int R;           // rows per wave (variant 2) or per workgroup (variant 1)
    int U;           // column-step unroll
    int waves;       // waves per workgroup
    int nt;          // non-temporal loads of A
    int grid;        // workgroups
    int rows_per_wg;
};

// Choose the K1 shape for a shard of `rows` x `ncols` (variant 0 = default).
GemvPlan plan_gemv(int variant, int rows, int ncols);

// K1: Ap = A[rows x lda] * p ; partials[wg] = sum over the workgroup's rows of p_local[row]*Ap[row].
// `done` may be null.  cblas_dgemv + cblas_ddot of code/MPI/cg.cc:100-105.
hipError_t launch_gemv(const GemvPlan &plan, const double *A, long lda, int rows, const double *p_full,
                       const double *p_local, double *Ap, double *partials, const int *done, hipStream_t s);

// K2: out[0] = deterministic sum of partials[0..n).
hipError_t launch_reduce_partials(const double *partials, int n, double *out, const int *done, hipStream_t s);

// K3: alpha = rsold / max(conj, rsold*1e-14); x += alpha p; r -= alpha Ap; partials[wg] = sum r_i^2.
// conj = sum_{q<nranks} gathered[q].  cg.cc:107-116.
hipError_t launch_update_xr(int count, const double *p_local, const double *Ap, double *x, double *r,
                            const Scalars *sc, int parity, const double *gathered, int nranks,
                            double *partials, hipStream_t s);
int update_xr_grid(int count);

// K4: rsnew = sum gathered[q] -> rs[(k+1)&1]; if sqrt(rsnew) < tol: done=1, k_final=k, return;
// else beta = rsnew/rsold; p_local = r + beta*p_local.  cg.cc:117-132.
hipError_t launch_update_p(int count, const double *r, double *p_local, Scalars *sc, int parity, int k,
                           double tol, const double *gathered, int nranks, hipStream_t s);

// Initial residual pieces (cg.cc:79-92): r = b - Ap ; p_local = r ; partials[wg] = sum r_i^2.
hipError_t launch_init_residual(int count, const double *b, const double *Ap, double *r, double *p_local,
                                double *partials, hipStream_t s);
// rs[0] = rs[1] = sum over ranks of slot kSlotRr; done = 0.
hipError_t launch_set_rsold(Scalars *sc, const double *gathered, int nranks, hipStream_t s);

// DEBUG block (cg.cc:144-151): partials[3*wg + {0,1,2}] = sum (Ax-b)^2, b^2, x^2 over this shard's rows.
hipError_t launch_debug_norms(int count, const double *Ax, const double *b, const double *x, double *partials,
                              hipStream_t s);
hipError_t launch_reduce_partials3(const double *partials, int n, double *out3, hipStream_t s);

// generate_lap2d_matrix (cg.cc:159-188) for rows [row0,row0+rows) straight into device memory; pad columns = 0.
hipError_t launch_generate_lap2d(double *A, long lda, int size, int row0, int rows, hipStream_t s);

// Matrix::read scatter (matrix.cc:12-21): A[(I[z]-row0)*lda + J[z]] = a[z] for entries already filtered to this shard.
hipError_t launch_scatter_coo(double *A, long lda, int row0, const int *I, const int *J, const double *a, long nz,
                              hipStream_t s);

// Loopback "collective": copy local[kSlots] of every shard into gathered[] of every shard (<= 16 shards).
hipError_t launch_loopback_gather(double *const *gathered_ptrs, const Scalars *const *scalar_ptrs, int nshards,
                                  hipStream_t s);

}  // namespace cgx
