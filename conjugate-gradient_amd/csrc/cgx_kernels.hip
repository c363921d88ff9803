// cgx_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the dense fp64 CG hot path.
//
// The path restated here is CGSolver::solve's loop body, code/MPI/cg.cc:96-137 (reference file:line).
// One iteration k is TWO kernels and ONE exchange:
//
//   K1 gemv_fused   tail of iteration k-1:  rsnew = r.r                          cg.cc:116-117
//                                           if sqrt(rsnew) < tol: break          cg.cc:120-121
//                                           beta = rsnew/rsold                   cg.cc:124
//                                           p = r + beta p   (on the fly)        cg.cc:127-129
//                                           rsold = rsnew                        cg.cc:132
//                   head of iteration k:    Ap_sub = A_sub p                     cg.cc:100-102 (cblas_dgemv)
//                                           partials of p_sub.Ap_sub             cg.cc:105     (cblas_ddot)
//   [several ranks: Kp prefold_ap           the column pieces of K1's Ap added up; one p_sub.Ap_sub partial per
//                                           512-row chunk of the slice ("Chunks" below)          cg.cc:105]
//   -- exchange: all-gather of [Ap slice | p.Ap partials]                        cg.cc:106 (MPI_Allreduce) + 135-136
//   K3 update_xr    p.Ap = fixed-order sum of all ranks' partials; alpha         cg.cc:107
//                   x_sub += alpha p_sub                                         cg.cc:110
//                   r -= alpha Ap for ALL n rows (r is replicated); r.r          cg.cc:113,116
//   [CGX_COMM_P2P: Kp, the exchange and K3 are ONE kernel, k_update_xr_p2p (flag words) / k_update_xr_p2p_tagged]
//
// What travels between ranks is Ap, not p: r and p are replicated, every rank updates the whole r from the
// gathered Ap and reduces r.r over all n rows in the same fixed order, so r.r is bit-identical on every
// rank WITHOUT a second all-reduce, and p = r + beta p is formed by every rank inside the next K1 while it
// streams the columns anyway.  The reference's 2 all-reduces + 1 all-gather per iteration become one
// exchange; the arithmetic per element is the reference's (same formulas, fma), only the summation order of
// the dot products differs, which the reference does not fix either (OpenBLAS/MPI reduction order).
// None of this is derived from code/CUDA/cg.cu (thread-per-row-chunk kernels with atomicAdd); these are
// streaming kernels without floating-point atomics, deterministic for a fixed launch shape.
//
// Everything is HBM-bound (0.25 flop/byte): no MFMA.  What matters is 16 B/lane coalesced loads of A's
// rows, enough independent loads in flight per CU, p/r served from L2/LDS instead of HBM, and keeping
// every scalar on the device.
#include "cgx_kernels.h"
#include "cgx_device.h"

#include <hip/hip_ext.h>

namespace cgx {

enum { kPlain = 0, kFusedSingle = 1 };   // the fused form always reads the replicated, contiguous r

template <bool NT>
__device__ __forceinline__ d2 load_a(const double *ptr)
{
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2 *>(ptr));
    else return *reinterpret_cast<const d2 *>(ptr);
}

// ------------------------------------------------------------------------------------------------
// the exchanged residual (SegView) and the scalar sums over ranks
// ------------------------------------------------------------------------------------------------
// Owner of column c: q = min(c / n_loc, nranks-1), without an integer division: the host supplies the
// round-up magic number (div_magic, div_shift) for n_loc (seg_finalize), exact for 0 <= c < 2^31.
__device__ __forceinline__ int seg_owner(const SegView &sv, int c)
{
    if (sv.n_loc <= 0) return sv.nranks - 1;   // N < P: floor(N/P) = 0 rows everywhere but on the last rank (cg.cc:255-266)
    const int t = (sv.div_shift == 32) ? c : (int)(__umulhi((unsigned)c, sv.div_magic) >> sv.div_shift);
    return t < sv.nranks - 1 ? t : sv.nranks - 1;
}

__device__ __forceinline__ double seg_load(const SegView &sv, int c)   // r[c], c < n
{
    const int q = (sv.nranks > 1) ? seg_owner(sv, c) : 0;
    return sv.base[c + q * sv.seg_gap];   // q*S + (c - q*n_loc)
}

__device__ __forceinline__ double seg_sum_slot(const SegView &sv, int slot)
{
    double s = sv.base[sv.Sr + slot];
    for (int q = 1; q < sv.nranks; ++q) s += sv.base[(long)q * sv.S + sv.Sr + slot];   // rank order
    return s;
}

// gathered layout on every shard: [rank q][slot v], kSlots doubles per rank
__device__ __forceinline__ double sum_ranks(const double *__restrict__ gathered, int slot, int nranks)
{
    double s = gathered[slot];
    for (int q = 1; q < nranks; ++q) s += gathered[q * kSlots + slot];   // rank order, same on every shard
    return s;
}

// Tail of iteration k-1, evaluated redundantly (and identically) by every WAVE of K1(k).
// r.r = fixed-order fold of K3's per-workgroup partials (the tail of the replicated-r segment): every wave of every
// workgroup of every rank folds the same values the same way (lane-strided sums, then the shuffle butterfly), so the
// break decision is the same everywhere, and no in-kernel grid reduction (ticket + fences, ~4 us at the end of K3) is
// needed.  No LDS and no workgroup barrier: on gfx9 a barrier's release fence drains vmcnt, i.e. it would wait for every
// A load a kernel has already issued ahead of the head.  The head is cut in two so that a kernel can put its first
// A loads between the halves: head_issue sends out the head's own loads (done, rsold, up to 256 partials), head_finish
// consumes them -- loads return in order, so the wait in between covers the head's loads only.
struct IterHead {
    double beta;
    bool stop;
};

struct HeadLoads {
    double rsold;
    int done;
    double a[4];
};

__device__ __forceinline__ HeadLoads head_issue(const Scalars *sc, const SegView &sv, int k)
{
    HeadLoads hl;
    const int nparts = sv.S - sv.Sr, lane = threadIdx.x & 63;
    const double *part = sv.base + sv.Sr;
    hl.done = sc->done;                                  // converged earlier: the whole grid drains immediately
    hl.rsold = sc->rs[(k > 0 ? k - 1 : 0) & 1];          // stored by the previous K1: independent of the fold
    // unconditional loads (clamped index, value discarded by a select): a load behind a branch would make the number of
    // loads in flight path dependent, and the compiler then waits for ALL of them (vmcnt(0)) instead of counting
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        const double val = part[t < nparts ? t : nparts - 1];
        hl.a[u] = (t < nparts) ? val : 0.0;
    }
    return hl;
}

__device__ __forceinline__ IterHead head_finish(const HeadLoads &hl, Scalars *sc, const SegView &sv, int k, double tol)
{
    IterHead h{0.0, false};
    const int nparts = sv.S - sv.Sr, lane = threadIdx.x & 63;
    const double *part = sv.base + sv.Sr;
    double v = (hl.a[0] + hl.a[1]) + (hl.a[2] + hl.a[3]);
    for (int t = lane + 256; t < nparts; t += 256) {     // more than 256 partials: n > 65536
        const double a0 = part[t], a1 = (t + 64 < nparts) ? part[t + 64] : 0.0;
        const double a2 = (t + 128 < nparts) ? part[t + 128] : 0.0, a3 = (t + 192 < nparts) ? part[t + 192] : 0.0;
        v += (a0 + a1) + (a2 + a3);
    }
    const double rsnew = wave_sum(v);                                // r.r over all rows, cg.cc:116-117 (k==0: cg.cc:91-92)
    const bool first = (blockIdx.x == 0 && threadIdx.x == 0) && !hl.done;   // nothing is written once converged
    if (k == 0) {                                                    // p = r (cg.cc:85): beta = 0, p_old = 0
        if (first) { sc->rs[0] = rsnew; sc->rs[1] = rsnew; }
        return h;
    }
    if (first) sc->rs[k & 1] = rsnew;                                // rsold = rsnew, cg.cc:132
    if (sqrt(rsnew) < tol) {                                         // cg.cc:120-121: break before the p update
        if (first) { sc->k_final = k - 1; sc->done = 1; }
        h.stop = true;
        return h;
    }
    h.beta = rsnew / hl.rsold;                                       // cg.cc:124
    return h;
}

// Both halves back to back; *done = the flag as loaded.  All lanes of the wave must be active (shuffles).
__device__ __forceinline__ IterHead iteration_head(Scalars *sc, const SegView &sv, int k, double tol, int *done)
{
    const HeadLoads hl = head_issue(sc, sv, k);
    *done = hl.done;
    return head_finish(hl, sc, sv, k, tol);
}

// p_new for the column pair (c, c+1); pad columns (>= n) stay exactly 0.
// SINGLE (one shard): r is contiguous and zero padded up to lda, one 16-B load.
template <bool SINGLE>
__device__ __forceinline__ d2 make_p(const SegView &sv, double beta, d2 p_old, int c)
{
    d2 r;
    if constexpr (SINGLE) {
        r = *reinterpret_cast<const d2 *>(sv.base + c);
    } else {
        r.x = (c < sv.n) ? seg_load(sv, c) : 0.0;
        r.y = (c + 1 < sv.n) ? seg_load(sv, c + 1) : 0.0;
    }
    d2 p;
    p.x = fma(beta, p_old.x, r.x);                                   // cg.cc:127-129
    p.y = fma(beta, p_old.y, r.y);
    return p;
}

// ------------------------------------------------------------------------------------------------
// K1, variant 1: column-split.  One workgroup owns R consecutive rows; its WAVES waves split the
// columns in 1 KiB pieces (lane = 16 B), so every global_load_dwordx4 of A is a fully coalesced
// 1 KiB wave access and the workgroup sweeps WAVES KiB of each row per step.  The vector is loaded once
// per step (16 B/lane, L2 hit) and reused from registers by all R rows: vector traffic = 1/R of the A
// stream.  U steps are issued back to back: R*U independent 1 KiB loads in flight per wave.
// Epilogue: shuffle wave reduction, LDS cross-wave combine in fixed wave order, Ap store, and the
// fused p.Ap (cg.cc:105): one partial per workgroup, folded by K3 in a fixed order (no atomics).
// FUSED: the vector is p_new = r + beta p_old, formed in registers from two L2-resident streams; the
// workgroup whose turn it is (step index mod grid) also stores it, so p_new is written exactly once.
// ------------------------------------------------------------------------------------------------
// The default shape (8,2) is held to 128 VGPRs = 4 workgroups per CU (the 4096-workgroup grid of N=32768 then runs in
// exactly 4 rounds; at 3 per CU it needs 5.33 and loses ~1.2 %, measured).
// LIGHT: for grids with few rounds (the shards of a multi-GPU run: 4096 rows = 512 ... 1024 workgroups).  Whatever a
// workgroup does before its first load and after its last one is then paid nearly in full by the whole launch.  With
// 256 registers to spend, the first trip's loads go out before the iteration head is folded and the epilogue's two
// vector operands are fetched before the sweep.  (A two-trips-deep software pipeline of the sweep was built and
// measured in round 2: no faster -- 157.0 us against 157.2 on the 4096 x 32768 shard -- and dropped again; what limits
// this shape is the access pattern itself, tools/hbm_rows_bw.hip.)
// PART = false (one-round form only): nobody consumes this launch's per-workgroup p.Ap partials (chunked exchange: the
// consumer of Ap reduces one partial per 512-row chunk itself), so the epilogue's two operand loads, the product and the
// store are left out.
template <int R, int U, int WAVES, int MODE, bool LIGHT = false, bool PART = true>
__global__ __launch_bounds__(WAVES * 64, (LIGHT ? 2 : ((R == 8 && U == 2) ? 4 : 1))) void k_gemv_colsplit(const double *__restrict__ A, long lda, int ncols_all, int rows,
                                                               int row0_global, const double *__restrict__ v,
                                                               double *__restrict__ p_new, SegView sv,
                                                               double *__restrict__ Ap, double *partials,
                                                               Scalars *sc, int k, double tol, int split, long ap_stride)
{
    constexpr bool FUSED = MODE != kPlain;
    constexpr bool NT = true;   // A is streamed once: non-temporal loads keep p and r in L2 (+12 % measured)
    __shared__ double red[WAVES][R];

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    // split > 1 (shards of a multi-GPU run): the columns of a row group are cut into `split` pieces of whole trips, one
    // workgroup each, with piece = blockIdx % split.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
    // share one), so with split = 8 every workgroup of an XCD sweeps the SAME eighth of the columns: each XCD's L2 holds
    // and re-fetches per launch one eighth of p and r instead of all of both, and what the workgroups of an XCD wait for
    // in lock step is that much less.  Measured on the shards of N = 32768: 154.5 -> 152.0 us (P=8, split 8),
    // 305.4 -> 302.2 (P=4, split 4), 603.1 -> 600.2 (P=2, split 2); with piece = blockIdx / groups (pieces not tied to
    // XCDs) the same split was 1-2 us SLOWER than no split -- the placement is what pays, not the granularity.
    // Piece s writes its partial row sums to Ap + s * ap_stride; whoever consumes Ap adds the pieces in ascending order.
    // The p.Ap partial is linear in Ap and needs no combine.
    const int groups = (int)gridDim.x / split;
    const int piece = (int)blockIdx.x % split;
    const int group = (int)blockIdx.x / split;
    const long row0 = (long)group * R;
    int ncols = ncols_all;
    int c_first = 0;
    if (split > 1) {
        // Pieces are balanced to one 128-column unit (one wave's 1-KiB load): with whole trips of 1024 columns per piece the
        // last piece of N = 46340 had 4 trips against 6 for the others, and since a piece IS an XCD here, that XCD sat idle
        // for a third of the launch (311.5 us for the 5792 x 46340 shard, 0.862 of peak, against 0.896 at N = 32768).
        const int units = (ncols_all + 127) / 128;
        const int base = units / split, rem = units - base * split;
        const int u0 = piece * base + (piece < rem ? piece : rem);
        const int u1 = u0 + base + (piece < rem ? 1 : 0);
        c_first = u0 * 128;
        ncols = u1 * 128 < ncols_all ? u1 * 128 : ncols_all;
        Ap += (long)piece * ap_stride;
    }
    // ncols = n rounded up to even: the pad columns up to the pitch hold zeros in A and in the vectors and are skipped
    // (sweeping them cost wave 0 of every workgroup one more, fully exposed, memory round trip at the end of its rows)
    const double *rfull = sv.base;   // FUSED: the replicated r, contiguous and zero padded up to lda

    // Row bases are workgroup-uniform (SGPR pairs); the lane's position is ONE 32-bit byte offset, so each A load
    // is `global_load_dwordx4 v, v_off, s[base]` and the R row pointers cost no vector registers.
    const char *a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        long row = row0 + r;
        if (row > rows - 1) row = rows - 1;   // tail workgroup: re-read the last row, result discarded
        if (row < 0) row = 0;                 // shard without rows (N < P): stream the dummy row, store nothing
        a[r] = reinterpret_cast<const char *>(A + row * lda);
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }

    constexpr int kStep = WAVES * 128;   // doubles swept by the workgroup per step
    int c = c_first + w * 128 + lane * 2;
    // index of the 512-column block (of the whole row) this WAVE's 128 columns lie in; it advances by one per step.  A piece
    // need not start on a block boundary, so the waves of a workgroup may be in different blocks.
    int step = (c_first + w * 128) / kStep;
    // next block whose p_new this workgroup stores: block b belongs to the row group b mod groups (the wave of that group
    // which sweeps a 128-column unit of it stores that unit), so every column is stored exactly once
    int my_step = group;
    if (my_step < step) my_step += ((step - my_step + groups - 1) / groups) * groups;
    double beta = 0.0;
    double ep_v = 0.0, ep_r = 0.0;       // LIGHT: the epilogue's operands, fetched ahead of the sweep

    // one trip = U steps: all vector and A loads first (R*U + 2U independent 16-B loads in flight per lane) ...
    auto load_trip = [&](d2 (&pvx)[U], d2 (&rvx)[U], d2 (&avx)[U][R], int cc) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = (unsigned)(cc + u * kStep) * 8u;   // uniform base + 32-bit lane offset
            pvx[u] = *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(v) + off);
            if constexpr (FUSED) rvx[u] = *reinterpret_cast<const d2 *>(reinterpret_cast<const char *>(rfull) + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
                avx[u][r] = load_a<NT>(reinterpret_cast<const double *>(a[r] + (unsigned)(cc + u * kStep) * 8u));
    };
    // ... then p = r + beta p_old (cg.cc:127-129), its one store, and the FMAs.
    auto compute_trip = [&](d2 (&pvx)[U], d2 (&rvx)[U], d2 (&avx)[U][R], int cc, int st) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (FUSED) {
                pvx[u].x = fma(beta, pvx[u].x, rvx[u].x);
                pvx[u].y = fma(beta, pvx[u].y, rvx[u].y);
                if (st + u == my_step) {
                    *reinterpret_cast<d2 *>(reinterpret_cast<char *>(p_new) + (unsigned)(cc + u * kStep) * 8u) = pvx[u];
                    my_step += groups;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc0[r] = fma(avx[u][r].x, pvx[u].x, acc0[r]);
                acc1[r] = fma(avx[u][r].y, pvx[u].y, acc1[r]);
            }
        }
    };
    auto full = [&](int cc) { return cc + (U - 1) * kStep < ncols; };   // a whole trip fits (per lane)

    d2 pv[U], rv2[U];
    d2 av[U][R];
    {
        // The first trip's loads do not depend on the iteration head (done / r.r / beta).  Issue order: the head's own
        // loads, (LIGHT) the epilogue's two operands, the first trip -- then the head is finished while the first trip
        // is in flight: loads return in order, so the wait in front of the fold covers the head's loads only.  Every one
        // of these loads is unconditional for the lanes that have a whole first trip (a load behind a divergent branch
        // makes the count path dependent and the compiler then drains everything).
        // Hoisting needs the trip's registers during the head: only for the light shapes (with R*U = 16 the 80 extra
        // live registers push the 4-per-CU kernel past 128 VGPRs: the 4096-workgroup grid of N=32768 then runs 5.33
        // rounds instead of 4, -1.2 %) and for the one-round form, which has 256 registers.
        constexpr bool HOIST = LIGHT || R * U <= 8;
        HeadLoads hl{};
        if constexpr (FUSED) hl = head_issue(sc, sv, k);
        if constexpr (LIGHT && PART) {
            long er = row0 + (lane & (R - 1));
            if (er > rows - 1) er = rows - 1;
            if (er < 0) er = 0;
            int j = row0_global + (int)er;
            if (j > ncols_all - 1) j = ncols_all - 1;   // shard without rows (N < P): row0_global == n; the value is never used
            ep_v = v[j];
            if constexpr (FUSED) ep_r = seg_load(sv, j);
        }
        const bool first = HOIST && full(c);
        if (first) load_trip(pv, rv2, av, c);
        if constexpr (FUSED) {
            const IterHead h = head_finish(hl, sc, sv, k, tol);
            if (hl.done || h.stop) return;
            beta = h.beta;
        }
        // Keep all loads of a trip in flight: without this fence hipcc's occupancy-driven scheduler
        // re-serialises them as load / s_waitcnt vmcnt(0) / fma pairs (measured in the .s).
        __builtin_amdgcn_sched_barrier(0);
        if (first) {
            compute_trip(pv, rv2, av, c, step);
            c += U * kStep;
            step += U;
        }
        for (; full(c); c += U * kStep, step += U) {
            load_trip(pv, rv2, av, c);
            __builtin_amdgcn_sched_barrier(0);
            compute_trip(pv, rv2, av, c, step);
        }
    }
    for (; c < ncols; c += kStep, ++step) {   // remaining single steps (ncols is even, so c+1 < lda)
        d2 p1 = *reinterpret_cast<const d2 *>(v + c);
        if constexpr (FUSED) {
            const d2 r1 = *reinterpret_cast<const d2 *>(rfull + c);
            p1.x = fma(beta, p1.x, r1.x);
            p1.y = fma(beta, p1.y, r1.y);
            if (step == my_step) {
                *reinterpret_cast<d2 *>(p_new + c) = p1;
                my_step += groups;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            d2 a1 = load_a<NT>(reinterpret_cast<const double *>(a[r] + (unsigned)c * 8u));
            acc0[r] = fma(a1.x, p1.x, acc0[r]);
            acc1[r] = fma(a1.y, p1.y, acc1[r]);
        }
    }

    // the wave's R row sums: one shared reduction (see wave_sum_rows), not R butterflies -- in a launch with one round
    // of workgroups the epilogue of every workgroup is on the launch's critical path
#pragma unroll
    for (int r = 0; r < R; ++r) acc0[r] += acc1[r];
    {
        const int myrow = wave_sum_rows<R>(acc0, lane);
        if ((lane & (64 / R - 1)) == 0) red[w][myrow] = acc0[0];
    }
    __syncthreads();
    double d = 0.0;
    if (w == 0) {
        if (lane < R) {
            double s = red[0][lane];
#pragma unroll
            for (int i = 1; i < WAVES; ++i) s += red[i][lane];
            const long row = row0 + lane;
            if (row < rows) {
                Ap[row] = s;
                if constexpr (PART) {
                    const int j = row0_global + (int)row;
                    // (LIGHT: lane < R holds the operands of row row0 + lane, fetched ahead of the sweep)
                    double pl = LIGHT ? ep_v : v[j];
                    if constexpr (FUSED) pl = fma(beta, pl, LIGHT ? ep_r : seg_load(sv, j));   // same bits as the stored p_new[j]
                    d = pl * s;                                                 // cg.cc:105
                }
            }
        }
        if constexpr (PART) {
            d = group_sum<(R < 64 ? R : 64)>(d);   // only lanes 0..R-1 hold a term
            // One partial per workgroup; K3 folds all of them (all ranks') in a fixed order.  No ticket here:
            // 4096 workgroups taking a returning atomic on one word cost 3-10 % of K1 (measured).
            if (lane == 0) partials[blockIdx.x] = d;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K1, variant 2: row-split with LDS-staged vector tiles.  Each of the WAVES waves owns R rows (the
// workgroup WAVES*R rows) and all waves sweep the same columns, so the vector tile (TILE doubles) is
// fetched from L2 once per workgroup into LDS (double buffered, one barrier per tile) and read back
// with conflict-free ds_read_b128 (lane = 16 B).  L2->CU vector traffic = 1/(WAVES*R) of the A stream.
// FUSED: the staging threads form p_new = r + beta p_old on the way into LDS; the workgroup whose turn
// it is (tile index mod grid) also stores the tile to p_new.
// ------------------------------------------------------------------------------------------------
template <int R, int U, int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_gemv_ldsp(const double *__restrict__ A, long lda, int rows,
                                                           int row0_global, const double *__restrict__ v,
                                                           double *__restrict__ p_new, SegView sv,
                                                           double *__restrict__ Ap, double *partials,
                                                           Scalars *sc, int k, double tol)
{
    constexpr bool FUSED = MODE != kPlain;
    constexpr bool SINGLE = MODE == kFusedSingle;
    constexpr bool NT = true;
    double beta = 0.0;
    if constexpr (FUSED) {
        int done;
        const IterHead h = iteration_head(sc, sv, k, tol, &done);
        if (done || h.stop) return;
        beta = h.beta;
    }
    constexpr int TILE = 2048;                       // doubles of the vector per LDS buffer (16 KiB)
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
        if (row < 0) row = 0;
        a[r] = A + row * lda;
    }
    double acc0[R], acc1[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }

    const int ntiles = (ncols + TILE - 1) / TILE;
    int my_tile = (int)blockIdx.x;                   // next tile whose p_new this workgroup stores
    d2 stage[kPerThread];
    auto fetch_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const int c = t * TILE + (i * kThreads + (int)threadIdx.x) * 2;
            d2 x = (c < ncols) ? *reinterpret_cast<const d2 *>(v + c) : d2{0.0, 0.0};
            if constexpr (FUSED) {
                if (c < ncols) x = make_p<SINGLE>(sv, beta, x, c);
            }
            stage[i] = x;
        }
    };
    auto commit_tile = [&](int t, int buf) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i)
            *reinterpret_cast<d2 *>(&ptile[buf][(i * kThreads + (int)threadIdx.x) * 2]) = stage[i];
        if constexpr (FUSED) {
            if (t == my_tile) {
#pragma unroll
                for (int i = 0; i < kPerThread; ++i) {
                    const int c = t * TILE + (i * kThreads + (int)threadIdx.x) * 2;
                    if (c < ncols) *reinterpret_cast<d2 *>(p_new + c) = stage[i];
                }
                my_tile += (int)gridDim.x;
            }
        }
    };
    fetch_tile(0);
    commit_tile(0, 0);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        const int base = t * TILE;
        // issue the next tile's vector loads early; they land in registers while this tile streams A
        if (t + 1 < ntiles) fetch_tile(t + 1);
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
        if (t + 1 < ntiles) commit_tile(t + 1, buf ^ 1);
        __syncthreads();   // next buffer complete, current buffer free for tile t+2
    }

    double d = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        double s = wave_sum(acc0[r] + acc1[r]);
        const long row = row0 + r;
        if (row < rows) {
            if (lane == 0) Ap[row] = s;
            const int j = row0_global + (int)row;
            double pl = v[j];
            if constexpr (FUSED) pl = fma(beta, pl, seg_load(sv, j));
            d += pl * s;   // same value in every lane
        }
    }
    if (lane == 0) red[w] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = red[0];
#pragma unroll
        for (int i = 1; i < WAVES; ++i) tot += red[i];
        partials[blockIdx.x] = tot;
    }
}

// ------------------------------------------------------------------------------------------------
// Chunks.  Between K1 and the exchange a rank's Ap slice is handled in chunks of kChunkRows consecutive rows, one workgroup
// each (a thread owns a pair of rows, so every access is 16 B).  For its pair a thread adds K1's column pieces in ascending
// order (split == 1: the slice is already whole), and the workgroup reduces the chunk's part of p_sub . Ap_sub (cg.cc:105)
// in one fixed order.  What travels between ranks is then [Ap slice | one p.Ap partial per chunk]: n/512 partials in all,
// whatever the grid of K1 was -- K1's own per-workgroup partials (4096 per rank with the columns split 8 ways) are not
// folded by anybody on a multi-GPU run.  Used by k_prefold_ap (RCCL, separate mailbox exchange, loopback) and by the
// pushers of k_update_xr_p2p (fused exchange): the same code, the same bits on every transport.
// ------------------------------------------------------------------------------------------------
// Branch-free: every load is unconditional (clamped index, value dropped by a select), so that all of them -- up to
// kMaxSplit pieces and the two elements of p -- are in flight together; loads behind a branch or in a loop of unknown
// length are waited for one by one (seen in the ISA: s_waitcnt vmcnt(0) after each piece).
constexpr int kMaxSplit = 8;
__device__ __forceinline__ d2 chunk_pair(const double *__restrict__ parts, int split, long stride, int row, int Sr)
{
    const int rc = row < Sr ? row : Sr - 2;                           // Sr is even and >= 2, slices are 16-B aligned
    d2 v[kMaxSplit];
#pragma unroll
    for (int sp = 0; sp < kMaxSplit; ++sp)
        v[sp] = *reinterpret_cast<const d2 *>(parts + (sp < split ? sp : split - 1) * stride + rc);
    d2 a = v[0];
#pragma unroll
    for (int sp = 1; sp < kMaxSplit; ++sp) {                          // ascending piece order
        a.x = sp < split ? a.x + v[sp].x : a.x;
        a.y = sp < split ? a.y + v[sp].y : a.y;
    }
    if (row >= Sr) a = d2{0.0, 0.0};
    return a;
}

// the pair's two elements of p_sub (p_loc = p_new + row0; row0 may be odd: 8-B loads), 0 behind the last row
__device__ __forceinline__ d2 chunk_p(const double *__restrict__ p_loc, int row, int rows)
{
    const int last = rows > 0 ? rows - 1 : 0;                         // rows == 0: p_loc[0] is still inside p (zero pad)
    const double p0 = p_loc[row < rows ? row : last];
    const double p1 = p_loc[row + 1 < rows ? row + 1 : last];
    return d2{row < rows ? p0 : 0.0, row + 1 < rows ? p1 : 0.0};
}

template <int WAVES>
__device__ __forceinline__ double chunk_dot(d2 p, d2 a, double *lds)
{
    return block_sum<WAVES>(fma(p.y, a.y, p.x * a.x), lds);
}

// One workgroup per chunk, in front of the exchange: dst = the Ap slice of this rank's segment, tail = its chunk partials.
__global__ __launch_bounds__(256) void k_prefold_ap(const double *__restrict__ parts, int split, long stride, int rows, int Sr,
                                                     const double *__restrict__ p_loc, double *__restrict__ dst,
                                                     double *__restrict__ tail, const Scalars *sc)
{
    __shared__ double lds[4];
    // every load of the kernel is issued before the first wait: one memory round trip, not a chain of them
    const int done = sc->done;
    const int row = ((int)blockIdx.x * 256 + (int)threadIdx.x) * 2;
    const d2 p = chunk_p(p_loc, row, rows);
    const d2 a = chunk_pair(parts, split, stride, row, Sr);
    // converged earlier (uniform): nothing is stored -- as a predicate on the stores, not as an early return, which the
    // compiler turns into "wait for the flag, then start loading" (seen in the ISA)
    if (!done && split > 1 && row < Sr) *reinterpret_cast<d2 *>(dst + row) = a;
    const double d = chunk_dot<4>(p, a, lds);
    if (!done && threadIdx.x == 0) tail[blockIdx.x] = d;
}

// ------------------------------------------------------------------------------------------------
// K3: x_sub += alpha p_sub ; r -= alpha Ap (all n rows) ; r.r          (cg.cc:106-116)
// apv: the gathered segments [Ap slice | tail] of all ranks; the p.Ap partials of rank q are the
// tail_count doubles at tail_off of its segment.  rv: the replicated r, [r (lda) | scalars].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_xr(int n, int rows, int row0, const double *__restrict__ p_new,
                                                    SegView apv, int tail_off, int tail_count,
                                                    double *__restrict__ x, SegView rv, Scalars *sc, int parity,
                                                    double *partials)
{
    __shared__ double lds[4];
    double *r = rv.base;
    // Issue every load this thread needs before the first wait: the kernel is a chain of ~1 us memory round
    // trips (flag, partials, vectors), so they are overlapped instead of serialised.
    const int done = sc->done;
    const double rsold = sc->rs[parity];
    const int i = blockIdx.x * 256 + threadIdx.x;                     // global row
    const int li = i - row0;
    const bool in = i < n, own = in && li >= 0 && li < rows;
    double ap_i = 0.0, r_i = 0.0, p_i = 0.0, x_i = 0.0;
    if (in) { ap_i = seg_load(apv, i); r_i = r[i]; }
    if (own) { p_i = p_new[i]; x_i = x[li]; }
    // p.Ap = sum of every K1 workgroup partial of every rank (cblas_ddot + MPI_Allreduce, cg.cc:105-106), folded
    // in one fixed order by every workgroup of every rank: bit-identical everywhere.
    // One flat index over (rank, partial): with a few partials per rank (the chunk partials of a multi-GPU run: n/512 in
    // all) every thread has at most one or two loads, all in flight together -- a loop over the ranks with one load each was
    // a chain of P dependent round trips (5.9 us at P = 8 against 4.8 at P = 2 and 4, round 3).
    double cs = 0.0;
    {
        const int total = apv.nranks * tail_count;
        const double *tails = apv.base + apv.Sr + tail_off;
        auto at = [&](int f) {
            if (apv.nranks == 1) return tails[f];
            const int q = f / tail_count;
            return tails[(long)q * apv.S + (f - q * tail_count)];
        };
        for (int f = threadIdx.x; f < total; f += 4 * 256) {   // four independent loads in flight (clamped, not branched around)
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = f + u * 256;
                const double val = at(g < total ? g : total - 1);
                a[u] = g < total ? val : 0.0;
            }
            cs += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    if (done) return;   // converged earlier (uniform over the grid): nothing is written
    const double conj = block_sum<4>(cs, lds);
    const double alpha = safeguarded_alpha(rsold, conj);             // cg.cc:107
    double rr = 0.0;
    if (in) {
        const double rn = fma(-alpha, ap_i, r_i);                     // cg.cc:113, for every row (r is replicated)
        r[i] = rn;
        rr = rn * rn;                                                 // cg.cc:116
    }
    if (own) x[li] = fma(alpha, p_i, x_i);                            // cg.cc:110, own rows only
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) r[rv.Sr + blockIdx.x] = rr;   // one r.r partial per workgroup, folded by the next K1's head
}

// Loop ran out after k iterations: the tail of iteration k-1 that the next K1 would have done (same code,
// same bits), or, for k == 0, the rsold of cg.cc:91-92.
__global__ __launch_bounds__(256) void k_close_iteration(Scalars *sc, SegView sv, int k, double tol)
{
    int done;
    (void)iteration_head(sc, sv, k, tol, &done);
}

// ------------------------------------------------------------------------------------------------
// K2, stand-alone form (setup / verification only): one workgroup folds partials in a fixed order.
// ------------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void k_reduce_partials(const double *__restrict__ partials, int n,
                                                          double *__restrict__ out)
{
    __shared__ double lds[4];
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) s += partials[(long)i * NV + v];
        s = block_sum<4>(s, lds);
        if (threadIdx.x == 0) out[v] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// setup / verification kernels (outside the iteration loop)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_init_residual(int n, const double *__restrict__ b_full, SegView apv,
                                                        SegView rv, double *__restrict__ partials)
{
    __shared__ double lds[4];
    double *r = rv.base;
    double rr = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {   // one trip up to 256 Ki rows
        const double rvv = b_full[i] - seg_load(apv, (int)i);   // r = b - A x, cg.cc:79-82 (all rows: r is replicated)
        r[i] = rvv;
        rr += rvv * rvv;                                        // rsold = r.p with p == r, cg.cc:85,91
    }
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) r[rv.Sr + blockIdx.x] = rr;   // same slots as K3's partials: K1(0) folds them (cg.cc:91-92)
}

// dst[0..count) = src[0..count); either side may be pinned host memory (x0 in / x out of a solve: a kernel instead of
// a copy-engine transfer, whose first use in a process costs ~8 ms -- inside the reference's timing window).
// cgx_solve_begin for a ZERO initial guess in one kernel (launch_solve_begin_zero, cgx_kernels.h): x = 0, r = b, the r.r partials
// behind r in k_init_residual's grouping (one per workgroup), p buffers / exchanged segments / scalar block zeroed, error word down.
__global__ __launch_bounds__(256) void k_solve_begin_zero(int n, long lda, const double *__restrict__ b_full, double *__restrict__ x,
                                                           SegView rv, double *__restrict__ p0, double *__restrict__ p1,
                                                           double *__restrict__ apg, long apg_count, Scalars *sc, int *err)
{
    __shared__ double lds[4];
    double *r = rv.base;
    double rr = 0.0;
    const long top = lda > apg_count ? lda : apg_count;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < top; i += (long)gridDim.x * 256) {
        if (i < lda) {
            const double bi = i < n ? b_full[i] : 0.0;          // r = b - A 0 = b, cg.cc:79-82
            x[i] = 0.0;
            p0[i] = 0.0;
            p1[i] = 0.0;
            r[i] = bi;
            rr += bi * bi;                                      // rsold = r.p with p == r, cg.cc:85,91
        }
        if (i < apg_count) apg[i] = 0.0;
    }
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) r[rv.Sr + blockIdx.x] = rr;
    if (blockIdx.x == 0) {
        if (threadIdx.x < sizeof(Scalars) / sizeof(double)) reinterpret_cast<double *>(sc)[threadIdx.x] = 0.0;
        if (threadIdx.x == 0 && err) *err = 0;
    }
}

// cgx_solve_end behind the verification GEMV in one kernel (launch_solve_end): one workgroup, fixed order of summation.
__global__ __launch_bounds__(1024) void k_solve_end(int n, const double *__restrict__ Ax, const double *__restrict__ b,
                                                     const double *__restrict__ x, const Scalars *sc, double *__restrict__ out)
{
    __shared__ double lds[3][16];
    double e = 0.0, bb = 0.0, xx = 0.0;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const double xi = x[i], bi = b[i], d = Ax[i] - bi;      // cg.cc:146-151
        out[i] = xi;
        e += d * d;
        bb += bi * bi;
        xx += xi * xi;
    }
    e = wave_sum(e);
    bb = wave_sum(bb);
    xx = wave_sum(xx);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        lds[0][w] = e;
        lds[1][w] = bb;
        lds[2][w] = xx;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = lds[threadIdx.x][0];
        for (int i = 1; i < 16; ++i) s += lds[threadIdx.x][i];
        out[n + threadIdx.x] = s;
    }
    if (threadIdx.x == 3) {
        out[n + 3] = sc->rs[0];
        out[n + 4] = sc->rs[1];
    }
}

__global__ __launch_bounds__(256) void k_copy_doubles(double *__restrict__ dst, const double *__restrict__ src, long count)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void k_unpack_segments(SegView sv, double *__restrict__ v_full, long lda)
{
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < lda; c += (long)gridDim.x * 256)
        v_full[c] = (c < sv.n) ? seg_load(sv, (int)c) : 0.0;
}

__global__ __launch_bounds__(256) void k_debug_norms(int count, const double *__restrict__ Ax,
                                                      const double *__restrict__ b, const double *__restrict__ x,
                                                      double *__restrict__ partials)
{
    __shared__ double lds[4];
    double e = 0.0, bb = 0.0, xx = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long)gridDim.x * 256) {
        const double d = Ax[i] - b[i];     // cg.cc:146-148
        e += d * d;
        bb += b[i] * b[i];
        xx += x[i] * x[i];
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

// One entry of generate_lap2d_matrix, cg.cc:178-185 (0 <= i, j < size).
__device__ __forceinline__ double lap2d_entry(int size, int inc, long i, long j)
{
    if (j == i) return 4.0;                                      // cg.cc:183
    if (i > 0 && j == i - 1) return -1.0;                        // cg.cc:182
    if (i < size - 1 && j == i + 1) return -1.0;                 // cg.cc:184
    if (i > inc && j == i - 1 - inc) return -1.0;                // cg.cc:181
    if (i < size - 1 - inc && j == i + 1 + inc) return -1.0;     // cg.cc:185
    return 0.0;                                                  // cg.cc:178-180
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
            v[e] = (j < size) ? lap2d_entry(size, inc, i, j) : 0.0;
        }
        *reinterpret_cast<d2 *>(A + lr * lda + j0) = v;
    }
}

// TEST PROBE (cgx_probe_fill_matrix_hash): the row block filled with hash_entry -- dense, incompressible data for the parity
// and rate checks of K1 at the BASELINE sizes.  Same store pattern as the generator: one thread writes 16 B.
__global__ __launch_bounds__(256) void k_fill_hash(double *__restrict__ A, long lda, int n, int row0, int rows,
                                                    unsigned long long seed_mixed, int symmetric, double diag)
{
    const long pairs_per_row = lda / 2;
    const long total = (long)rows * pairs_per_row;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long lr = t / pairs_per_row;
        const int j0 = (int)(t - lr * pairs_per_row) * 2;
        const long i = (long)row0 + lr;
        d2 v;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = j0 + e;
            v[e] = (j < n) ? hash_entry(seed_mixed, symmetric, diag, i, j) : 0.0;
        }
        *reinterpret_cast<d2 *>(A + lr * lda + j0) = v;
    }
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

// K3 for vectors longer than 256 * kMaxVectorGrid rows (banded storage only: a dense matrix of that size does not
// exist): the same arithmetic as k_update_xr with the workgroups striding over the rows, so that the number of r.r
// partials every K1 workgroup folds stays bounded.
__global__ __launch_bounds__(256) void k_update_xr_strided(int n, int rows, int row0, const double *__restrict__ p_new,
                                                            SegView apv, int tail_off, int tail_count,
                                                            double *__restrict__ x, SegView rv, Scalars *sc, int parity)
{
    __shared__ double lds[4];
    double *r = rv.base;
    const int done = sc->done;
    const double rsold = sc->rs[parity];
    double cs = 0.0;
    for (int q = 0; q < apv.nranks; ++q) {
        const double *tail = apv.base + (long)q * apv.S + apv.Sr + tail_off;
        for (int j = threadIdx.x; j < tail_count; j += 256) cs += tail[j];
    }
    if (done) return;
    const double conj = block_sum<4>(cs, lds);
    const double alpha = safeguarded_alpha(rsold, conj);             // cg.cc:107
    double rr = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const double rn = fma(-alpha, seg_load(apv, (int)i), r[i]);   // cg.cc:113
        r[i] = rn;
        rr += rn * rn;                                                // cg.cc:116
        const long li = i - row0;
        if (li >= 0 && li < rows) x[li] = fma(alpha, p_new[i], x[li]);   // cg.cc:110
    }
    rr = block_sum<4>(rr, lds);
    if (threadIdx.x == 0) r[rv.Sr + blockIdx.x] = rr;
}

// ------------------------------------------------------------------------------------------------
// Banded storage (opt-in fast path; NOT the reference's dense GEMV contract).
// K1 on the diagonals of the row block: one thread per row, diagonals in ascending offset order (= ascending
// column order), every load coalesced: vals[t][i..i+63] and p[g+off .. g+off+63].  Same head, same p_new, same
// one-partial-per-workgroup contract as the dense K1, so K3, the exchange and the host loop are shared.
// Out-of-range columns are clamped instead of branched around: their stored value is exactly 0.
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(8))) d2u {   // two consecutive doubles at an 8-B aligned address: one
    double x, y;                                     // global_load_dwordx4 (unaligned access mode is on under HSA)
};

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k_spmv_dia(DiaView dv, int rows, int row0_global, int n, long lda,
                                                   const double *__restrict__ v, double *__restrict__ p_new, SegView sv,
                                                   double *__restrict__ Ap, double *__restrict__ partials, Scalars *sc,
                                                   int k, double tol)
{
    constexpr bool FUSED = MODE != kPlain;
    __shared__ double lds[4];
    const double *rfull = sv.base;   // FUSED: the replicated r, zero padded up to lda
    double beta = 0.0;
    if constexpr (FUSED) {
        int done;
        const IterHead h = iteration_head(sc, sv, k, tol, &done);
        if (done || h.stop) return;
        beta = h.beta;
        // p_new = r + beta p_old (cg.cc:127-129) for the columns that are not rows of this shard (other ranks' rows
        // and the pad); the shard's own rows are stored by the row loop below.
        const long before = row0_global, after = lda - ((long)row0_global + rows);
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < before + after; c += (long)gridDim.x * 256) {
            const long cc = c < before ? c : c - before + row0_global + rows;
            p_new[cc] = fma(beta, v[cc], rfull[cc]);
        }
    }
    // Two consecutive rows per thread, so that the diagonals (streamed once: non-temporal), the vectors, Ap and p_new
    // all move as 16-B pieces.  Row i+1 needs no guard: behind the last row the diagonals hold zeros (ld is even and
    // zero filled), the vectors are zero padded up to lda, and the Ap slice is padded to an even count.
    // CH diagonals are in flight at a time: all their loads are issued before the first FMA (see K1 above).
    double d = 0.0;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2; i < rows; i += (long)gridDim.x * 512) {
        const int g = row0_global + (int)i;
        const d2u pv = *reinterpret_cast<const d2u *>(v + g);
        d2u pr{0.0, 0.0};
        if constexpr (FUSED) pr = *reinterpret_cast<const d2u *>(rfull + g);
        double acc0 = 0.0, acc1 = 0.0;
        for (int t0 = 0; t0 < dv.ndiag; t0 += CH) {
            d2u qv[CH], qr[CH];
            d2 a[CH];
            bool low[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int t = t0 + u < dv.ndiag ? t0 + u : dv.ndiag - 1;   // tail of the last chunk: re-load, not used
                const int j = g + dv.off[t];                              // column of row g; row g+1 has j+1
                const int jb = j < 0 ? 0 : (j > n - 1 ? n - 1 : j);       // outside [0,n): the stored value is 0
                low[u] = j < 0;                                           // j == -1: row g+1 needs column 0 = pair.x
                qv[u] = *reinterpret_cast<const d2u *>(v + jb);
                if constexpr (FUSED) qr[u] = *reinterpret_cast<const d2u *>(rfull + jb);
                a[u] = load_a<true>(dv.vals + t * dv.ld + i);            // ld and i are even: 16-B aligned
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (t0 + u < dv.ndiag) {
                    double q0 = qv[u].x, q1 = low[u] ? qv[u].x : qv[u].y;
                    if constexpr (FUSED) {
                        q0 = fma(beta, q0, qr[u].x);                      // same bits as the stored p_new[j]
                        q1 = fma(beta, q1, low[u] ? qr[u].x : qr[u].y);
                    }
                    acc0 = fma(a[u].x, q0, acc0);                         // cg.cc:100-102
                    acc1 = fma(a[u].y, q1, acc1);
                }
            }
        }
        double p0 = pv.x, p1 = pv.y;
        if constexpr (FUSED) {
            p0 = fma(beta, p0, pr.x);
            p1 = fma(beta, p1, pr.y);
            *reinterpret_cast<d2u *>(p_new + g) = d2u{p0, p1};            // column g+1 beyond the block: same value as
        }                                                                 // the pre-pass stores (r is replicated)
        d2 out;
        out.x = acc0;
        out.y = acc1;
        *reinterpret_cast<d2 *>(Ap + i) = out;
        d = fma(p0, acc0, d);                                             // cg.cc:105
        d = fma(p1, acc1, d);                                             // row beyond the block: acc1 == 0
    }
    d = block_sum<4>(d, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = d;
}

// ------------------------------------------------------------------------------------------------
// K1b with the vector staged in LDS windows (round 2).  The direct form above loads, per pair of rows and per diagonal,
// 16 B of p_old and 16 B of r straight from L2 (neighbouring threads overlap almost completely: 2.0 x the minimal number
// of L2 requests, measured) and forms p = r + beta p_old once per use.  Here a workgroup owns a tile of 512 consecutive
// rows; diagonals whose offsets lie within kWinGap of each other share a WINDOW of columns
//     [tile + lo, tile + 512 + hi)                      (for the 5-point inputs: three windows, -inc-1 | -1,0,+1 | +inc+1)
// and each window is fetched ONCE per tile with aligned 16-B loads, combined (p = fma(beta, p_old, r), cg.cc:127-129)
// on its way into LDS and read back by every row as two neighbouring doubles.  The diagonal values of the tile are
// requested before the staging so that they are in flight behind it.  Same arithmetic, same order of the row sums, same
// bits as the direct form.
// ------------------------------------------------------------------------------------------------
constexpr int kWinGap = 256;        // widest spread of offsets inside one window
constexpr int kMaxWindows = 8;      // more windows than that: the direct form runs
struct DiaWindows {
    int nwin;
    int start[kMaxWindows];         // first column of the window relative to the tile's first row (even + parity fix)
    int len[kMaxWindows];           // doubles (even)
    int base[kMaxWindows];          // position of the window in LDS (doubles, even)
    short idx[kMaxDiags];           // per diagonal: LDS index of column (row + off) for the tile's first row
    int total;                      // doubles of LDS in all
};

template <int MODE, int CH>
__global__ __launch_bounds__(256) void k_spmv_dia_lds(DiaView dv, DiaWindows dw, int rows, int row0_global, int n, long lda,
                                                       const double *__restrict__ v, double *__restrict__ p_new, SegView sv,
                                                       double *__restrict__ Ap, double *__restrict__ partials, Scalars *sc,
                                                       int k, double tol)
{
    constexpr bool FUSED = MODE != kPlain;
    extern __shared__ __attribute__((aligned(16))) double win[];
    __shared__ double lds[4];
    const double *rfull = sv.base;   // FUSED: the replicated r, zero padded up to lda
    double beta = 0.0;
    if constexpr (FUSED) {
        int done;
        const IterHead h = iteration_head(sc, sv, k, tol, &done);
        if (done || h.stop) return;
        beta = h.beta;
        // p_new for the columns that are not rows of this shard: as in the direct form
        const long before = row0_global, after = lda - ((long)row0_global + rows);
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < before + after; c += (long)gridDim.x * 256) {
            const long cc = c < before ? c : c - before + row0_global + rows;
            p_new[cc] = fma(beta, v[cc], rfull[cc]);
        }
    }
    const int tid = threadIdx.x;
    const int ntiles = (rows + 511) / 512;
    double d = 0.0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long i = (long)tile * 512 + 2 * tid;       // this thread's two local rows i, i+1 (may lie behind the block)
        const int g0 = row0_global + tile * 512;          // global index of the tile's first row
        const bool live = i < rows;
        const long il = live ? i : 0;                     // behind the block: re-read row 0, nothing is stored
        const int g = row0_global + (int)il;
        // the tile's diagonal values and own-row vector pieces first: in flight while the windows are staged
        d2 a[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int t = u < dv.ndiag ? u : dv.ndiag - 1;
            a[u] = load_a<true>(dv.vals + t * dv.ld + il);
        }
        const d2u pv = *reinterpret_cast<const d2u *>(v + g);
        d2u pr{0.0, 0.0};
        if constexpr (FUSED) pr = *reinterpret_cast<const d2u *>(rfull + g);
        // stage the windows: aligned pairs of columns, clamped at the ends of the vectors
        for (int w = 0; w < dw.nwin; ++w) {
            const int c0 = g0 + dw.start[w];              // even
            double *dst = win + dw.base[w];
            for (int q = tid; q < dw.len[w] / 2; q += 256) {
                const int c = c0 + 2 * q;
                double x0 = 0.0, x1 = 0.0, r0 = 0.0, r1 = 0.0;
                if (c >= 0 && c + 1 < lda) {
                    const d2 xv = *reinterpret_cast<const d2 *>(v + c);
                    x0 = xv.x;
                    x1 = xv.y;
                    if constexpr (FUSED) {
                        const d2 rv = *reinterpret_cast<const d2 *>(rfull + c);
                        r0 = rv.x;
                        r1 = rv.y;
                    }
                } else {                                   // first / last tile only; a column outside [0,n) meets a stored 0
                    if (c >= 0 && c < lda) { x0 = v[c]; if constexpr (FUSED) r0 = rfull[c]; }
                    if (c + 1 >= 0 && c + 1 < lda) { x1 = v[c + 1]; if constexpr (FUSED) r1 = rfull[c + 1]; }
                }
                if constexpr (FUSED) {
                    x0 = fma(beta, x0, r0);                // same bits as the stored p_new[c]
                    x1 = fma(beta, x1, r1);
                }
                d2 o;
                o.x = x0;
                o.y = x1;
                *reinterpret_cast<d2 *>(dst + 2 * q) = o;
            }
        }
        __syncthreads();
        double acc0 = 0.0, acc1 = 0.0;
        for (int t0 = 0; t0 < dv.ndiag; t0 += CH) {
            if (t0 > 0) {                                  // more than CH diagonals: the next chunk's values
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const int t = t0 + u < dv.ndiag ? t0 + u : dv.ndiag - 1;
                    a[u] = load_a<true>(dv.vals + t * dv.ld + il);
                }
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (t0 + u < dv.ndiag) {
                    const double *q = win + dw.idx[t0 + u] + 2 * tid;
                    acc0 = fma(a[u].x, q[0], acc0);        // cg.cc:100-102, ascending columns
                    acc1 = fma(a[u].y, q[1], acc1);
                }
            }
        }
        if (live) {
            double p0 = pv.x, p1 = pv.y;
            if constexpr (FUSED) {
                p0 = fma(beta, p0, pr.x);
                p1 = fma(beta, p1, pr.y);
                *reinterpret_cast<d2u *>(p_new + g) = d2u{p0, p1};
            }
            d2 out;
            out.x = acc0;
            out.y = acc1;
            *reinterpret_cast<d2 *>(Ap + i) = out;
            d = fma(p0, acc0, d);                          // cg.cc:105
            d = fma(p1, acc1, d);
        }
        __syncthreads();                                   // the windows are free for the next tile
    }
    d = block_sum<4>(d, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = d;
}

// generate_lap2d_matrix (cg.cc:159-188) straight into banded storage.
__global__ __launch_bounds__(256) void k_dia_generate_lap2d(double *__restrict__ vals, DiaView dv, int size, int row0,
                                                             int rows, int inc)
{
    const long total = (long)dv.ndiag * rows;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int t = (int)(idx / rows);
        const long i = idx - (long)t * rows;
        const long g = row0 + i, j = g + dv.off[t];
        vals[t * dv.ld + i] = (j >= 0 && j < size) ? lap2d_entry(size, inc, g, j) : 0.0;
    }
}

// Which diagonals of a dense row block hold a non-zero?  flags[(j - i_global) + n - 1] = 1 (same value from every
// writer, so the race is benign).
__global__ __launch_bounds__(256) void k_dia_mark(const double *__restrict__ A, long lda, int n, int row0, int rows,
                                                   unsigned char *__restrict__ flags)
{
    const long total = (long)rows * n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long i = idx / n, j = idx - i * n;
        if (A[i * lda + j] != 0.0) flags[j - (row0 + i) + (n - 1)] = 1;
    }
}

__global__ __launch_bounds__(256) void k_dia_pack(const double *__restrict__ A, long lda, int n, int row0, int rows,
                                                   double *__restrict__ vals, DiaView dv)
{
    const long total = (long)dv.ndiag * rows;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int t = (int)(idx / rows);
        const long i = idx - (long)t * rows;
        const long j = row0 + i + dv.off[t];
        vals[t * dv.ld + i] = (j >= 0 && j < n) ? A[i * lda + j] : 0.0;
    }
}

// Matrix::read on the device (see cgx_kernels.h, launch_coo_assign).  Element of assignment (i,j): its index in the
// block's own storage, or -1 if the row is not local (or, banded, the diagonal is not stored: cannot happen for
// offsets collected from the same entries).
template <bool BANDED>
__device__ __forceinline__ long coo_cell(long lda, const DiaView &dv, int n, int row0, int rows, int i, int j)
{
    const int li = i - row0;
    if (li < 0 || li >= rows) return -1;
    if constexpr (!BANDED) {
        return (long)li * lda + j;
    } else {
        const int off = j - i;
        int lo = 0, hi = dv.ndiag - 1;
        while (lo < hi) {                       // offsets are ascending, at most 64
            const int mid = (lo + hi) >> 1;
            if (dv.off[mid] < off) lo = mid + 1;
            else hi = mid;
        }
        if (dv.ndiag <= 0 || dv.off[lo] != off) return -1;
        return (long)lo * dv.ld + li;
    }
}

// PASS 0: claim, 1: resolve, 2: write.  Sequence of assignment (z, mirror) = 2z + mirror + 1 (0 = untouched).
template <bool BANDED, int PASS>
__global__ __launch_bounds__(256) void k_coo_assign(double *__restrict__ cells, long lda, DiaView dv, int n, int row0,
                                                     int rows, const int *__restrict__ I, const int *__restrict__ J,
                                                     const double *__restrict__ a, long nz, int sym,
                                                     unsigned char *__restrict__ win)
{
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(cells);
    for (long z = (long)blockIdx.x * 256 + threadIdx.x; z < nz; z += (long)gridDim.x * 256) {
        const int i0 = I[z], j0 = J[z];
        for (int mir = 0; mir <= (sym ? 1 : 0); ++mir) {
            const long c = coo_cell<BANDED>(lda, dv, n, row0, rows, mir ? j0 : i0, mir ? i0 : j0);
            if (c < 0) continue;
            const unsigned long long seq = 2ull * (unsigned long long)z + mir + 1;
            if constexpr (PASS == 0) atomicMax(bits + c, seq);                       // matrix.cc:17-20, last one wins
            else if constexpr (PASS == 1) win[2 * z + mir] = bits[c] == seq;
            else if (win[2 * z + mir]) cells[c] = a[z];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Direct peer exchange over xGMI: replaces one RCCL all-gather launch.
// Protocol per (channel, epoch), placement independent:
//   producer: plain 16-B stores of the payload into the PEER's mailbox slot [epoch&1][my rank]; every storing
//             thread fences at system scope; workgroup barrier; one lane stores flag = epoch with a system-scope
//             release atomic into the peer's flag word [channel][my rank];
//   consumer: one lane polls its OWN flag word [channel][peer] with system-scope acquire loads (bounded by the
//             100 MHz wall clock), workgroup barrier, every thread fences (acquire, system) and reads the slot
//             with system-scope loads (never served from a stale cache line), then plain-stores into the
//             ordinary working buffer the next kernel reads.
// Two parities suffice: a rank can start epoch e+2 only after every peer delivered e+1, which a peer does only
// after the kernel that consumed epoch e on that peer has finished (stream order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mailbox_allgather(MailboxView mv, int chan, unsigned long long epoch,
                                                            const double *__restrict__ src, int count, int tail_off,
                                                            int tail_n, double *__restrict__ dst, long dst_stride,
                                                            int sum_off, int copy_self, long long timeout_ticks, int *err)
{
    const int peer = blockIdx.x;
    const int tid = threadIdx.x;
    // an earlier wait expired: do not wait again, let the host see it.  Decided per workgroup (another workgroup may
    // raise the word while this one starts), so that no thread is left alone at a barrier.
    if (__syncthreads_or(__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) return;
    __shared__ double lds[4];
    double tail_sum = 0.0;
    if (tail_n > 0) {
        // fold my tail (the K1 partials) in a fixed order and ship ONE double behind the main payload (every
        // workgroup computes the same bits): the local half of MPI_Allreduce (cg.cc:106) rides in the exchange
        double v = 0.0;
        for (int i = tid; i < tail_n; i += 256) v += src[tail_off + i];
        tail_sum = block_sum<4>(v, lds);
    }
    if (peer == mv.rank) {
        if (copy_self)
            for (int i = tid; i < count; i += 256) dst[(long)mv.rank * dst_stride + i] = src[i];
        if (tail_n > 0 && tid == 0) dst[(long)mv.rank * dst_stride + sum_off] = tail_sum;
        return;
    }
    const int parity = (int)(epoch & 1);
    const long slot = mv.slot_bytes[chan];
    // ---- push my contribution into the peer's mailbox: [count doubles | tail sum] ----------------------------
    {
        double *out = reinterpret_cast<double *>(mv.base[peer] + mv.data_off[chan] +
                                                 ((long)parity * mv.nranks + mv.rank) * slot);
        const int pairs = count >> 1;
        for (int i = tid; i < pairs; i += 256)
            *reinterpret_cast<d2 *>(out + 2 * i) = *reinterpret_cast<const d2 *>(src + 2 * i);
        if ((count & 1) && tid == 0) out[count - 1] = src[count - 1];
        if (tail_n > 0 && tid == 0) out[count] = tail_sum;
        __threadfence_system();   // every storing wave: write back, wait for its own stores (vmcnt is per wave), before the barrier
        __syncthreads();
        if (tid == 0) {
            unsigned long long *flag = reinterpret_cast<unsigned long long *>(
                mv.base[peer] + ((long)chan * kMaxRanks + mv.rank) * kP2pFlagStride);
            // the same tail as chunk_publish: release fence, its own wait (the compiler drops the one behind buffer_wbl2 when it
            // can prove this wave has nothing outstanding -- harmless here, every wave has drained above, but not left to that),
            // then the flag as a relaxed system-scope store
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- wait for the peer's contribution in MY mailbox, then copy it out ------------------------------------
    __shared__ int s_ok;
    if (tid == 0) {
        const unsigned long long *flag = reinterpret_cast<const unsigned long long *>(
            mv.base[mv.rank] + ((long)chan * kMaxRanks + peer) * kP2pFlagStride);
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {   // relaxed polls, ONE acquire after
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > timeout_ticks) {   // every spin is bounded: give up, tell the host
                ok = 0;
                atomicExch(err, 1);
                break;
            }
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // system scope
    {
        const unsigned long long *in = reinterpret_cast<const unsigned long long *>(
            mv.base[mv.rank] + mv.data_off[chan] + ((long)parity * mv.nranks + peer) * slot);
        double *out = dst + (long)peer * dst_stride;
        const int total = count + (tail_n > 0 ? 1 : 0);
        // 8 independent loads in flight per thread: the slot is read straight from memory, never from a cache
        for (int base = 0; base < total; base += 8 * 256) {
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + tid;
                v[u] = (i < total) ? __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = base + u * 256 + tid;
                if (i < count) out[i] = __longlong_as_double((long long)v[u]);
                else if (i == count && tail_n > 0) out[sum_off] = __longlong_as_double((long long)v[u]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K3 with the exchange inside (CGX_COMM_P2P): iteration = K1 + this kernel, nothing else.
// The (peer, chunk) pairs -- P x cpr of them, about as many as the kernel has workgroups -- are dealt over the grid: a
// workgroup adds K1's column pieces for ONE chunk of kChunkRows rows, reduces the chunk's p.Ap partial, stores
// [chunk of Ap | partial] into ONE peer's mailbox slot (its own included: own rows take the same road as everybody
// else's) and raises that peer's flag word (me, chunk): one pass of independent loads, one 16-B store per thread, one fence,
// one flag.  (Round 2 had P workgroups push a whole slice each: 8 dependent rounds of loads per pusher, and every
// workgroup folded all of the rank's K1 partials -- 4096 with the columns split 8 ways: 13.4 us against 6.7 unsplit.)
// Then EVERY workgroup waits (bounded) for all P x cpr flags in its own mailbox, lane f polling word f, and reads what
// it needs straight from the slots with system-scope loads: the P x cpr partials, folded in one fixed order (bit-identical
// on every rank, MPI_Allreduce cg.cc:106), and per thread the one Ap element of its row.  No copy-out pass, no kernel
// boundary between exchange and update.  All workgroups must be co-resident (the host checks the occupancy): pushers
// never wait before they have pushed, so every flag a workgroup polls is raised by a workgroup that is running.
// ------------------------------------------------------------------------------------------------
// The exchange half of that kernel as device functions, so that the self-test of the transport (k_chunk_exchange_selftest,
// the gate in front of every P2P run) executes EXACTLY the stores, fences, flag words, polls and loads of the production
// kernel, not a look-alike.
struct ChunkItem {      // one (peer, chunk) pair of this workgroup, loads issued
    int peer, c, row;
    d2 a, pp;
};

__device__ __forceinline__ ChunkItem chunk_fetch(int pr, int cpr, const double *__restrict__ ap_src, int split, long part_stride,
                                                 int Sr, const double *__restrict__ p_loc, int rows)
{
    ChunkItem it;
    it.peer = pr / cpr;
    it.c = pr - it.peer * cpr;
    it.row = (it.c * 256 + (int)threadIdx.x) * 2;                     // this thread's pair of rows of MY slice
    it.pp = chunk_p(p_loc, it.row, rows);
    it.a = chunk_pair(ap_src, split, part_stride, it.row, Sr);
    return it;
}

// [chunk of Ap | its p.Ap partial] into the peer's slot, then the peer's flag word (me, chunk).
// Release, the producer form of the guide (MI355X_MICROARCH.md, "Valid forms"): EVERY storing wave waits for its own stores
// (`s_waitcnt vmcnt(0)`: the counter is per wave, and neither a barrier nor a workgroup-scope release drains it), then the
// workgroup barrier, then lane 0: one system-scope release fence, its own vmcnt(0), and the flag as a relaxed system-scope
// store.  (Round 3 had only the barrier and a release STORE by lane 0: in the ISA waves 1-3 went from their
// global_store_dwordx4 straight to s_barrier, so over xGMI the flag could have passed their part of the chunk -- the
// one-GPU self-test cannot see that; found by review)
__device__ __forceinline__ void chunk_publish(const MailboxView &mv, int chan, unsigned long long epoch, int cpr, int Sr,
                                              const ChunkItem &it, double *lds)
{
    const int P = mv.nranks, me = mv.rank, par = (int)(epoch & 1);
    const double d = chunk_dot<4>(it.pp, it.a, lds);                  // local part of MPI_Allreduce(p.Ap), cg.cc:105-106
    double *out = reinterpret_cast<double *>(mv.base[it.peer] + mv.data_off[chan] + ((long)par * P + me) * mv.slot_bytes[chan]);
    if (it.row < Sr) *reinterpret_cast<d2 *>(out + it.row) = it.a;
    if (threadIdx.x == 0) out[Sr + it.c] = d;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(mv.base[it.peer] + mv.cflag_off) + (me * cpr + it.c), epoch,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Lane f waits for flag word f = (source rank, chunk) of MY mailbox (bounded by the 100 MHz wall clock), then ONE
// system-scope acquire per polling wave, drained before the caller's barrier releases the other waves (the consumer form of
// the guide: relaxed polls -> one acquire -> s_waitcnt vmcnt(0) -> barrier -> loads).  Every load of handed-off bytes
// afterwards is a system-scope load as well, so nothing rests on one mechanism alone when the peer's stores arrive over
// xGMI instead of from a process on the same GPU.  Returns 0 in the lanes whose wait expired.
__device__ __forceinline__ int chunk_wait_all(const MailboxView &mv, unsigned long long epoch, int npairs, long long timeout_ticks,
                                              int *err)
{
    int ok = 1;
    const unsigned long long *flags = reinterpret_cast<const unsigned long long *>(mv.base[mv.rank] + mv.cflag_off);
    for (int f = threadIdx.x; f < npairs; f += 256) {
        const long long t0 = wall_clock64();
        // relaxed polls (an acquire per poll would invalidate caches every time round: 2-3x slower per hop)
        while (__hip_atomic_load(flags + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (wall_clock64() - t0 > timeout_ticks) {
                ok = 0;
                atomicExch(err, 1);
                break;
            }
        }
    }
    if (mv.acquire && ((int)threadIdx.x & ~63) < npairs) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    return ok;
}

__device__ __forceinline__ const unsigned long long *chunk_slot(const MailboxView &mv, int chan, unsigned long long epoch, int q)
{
    return reinterpret_cast<const unsigned long long *>(mv.base[mv.rank] + mv.data_off[chan]) +
           ((long)(epoch & 1) * mv.nranks + q) * (mv.slot_bytes[chan] / 8);
}

// this thread's share of all ranks' chunk partials, one fixed order (the caller block-reduces)
__device__ __forceinline__ double chunk_read_partials(const MailboxView &mv, int chan, unsigned long long epoch, int cpr, int Sr,
                                                      int npairs)
{
    double cs = 0.0;
    for (int f = threadIdx.x; f < npairs; f += 256) {
        const int q = f / cpr, c = f - q * cpr;
        cs += __longlong_as_double((long long)__hip_atomic_load(chunk_slot(mv, chan, epoch, q) + Sr + c, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_SYSTEM));
    }
    return cs;
}

__device__ __forceinline__ double chunk_read_ap(const MailboxView &mv, int chan, unsigned long long epoch, int q, int off)
{
    return __longlong_as_double((long long)__hip_atomic_load(chunk_slot(mv, chan, epoch, q) + off, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_SYSTEM));
}

__global__ __launch_bounds__(256) void k_update_xr_p2p(int n, int rows, int row0, const double *__restrict__ p_new,
                                                        SegView apv, int cpr, MailboxView mv, int chan,
                                                        unsigned long long epoch, double *__restrict__ x, SegView rv,
                                                        Scalars *sc, int parity_rs, long long timeout_ticks, int *err,
                                                        const double *__restrict__ ap_src, int split, long part_stride)
{
    __shared__ double lds[4];
    double *r = rv.base;
    const int tid = threadIdx.x, P = mv.nranks;
    const int done = sc->done;
    // (an atomic load, not a volatile one: the compiler waits for a volatile load on the spot -- a whole memory round trip
    // at the top of the kernel with nothing else in flight, seen in the ISA)
    const int had_err = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double rsold = sc->rs[parity_rs];
    const int i = blockIdx.x * 256 + tid;   // global row
    const int li = i - row0;
    const bool in = i < n, own = in && li >= 0 && li < rows;
    double r_i = 0.0, p_i = 0.0, x_i = 0.0;
    if (in) r_i = r[i];
    if (own) { p_i = p_new[i]; x_i = x[li]; }
    const int npairs = P * cpr;
    // the first (peer, chunk) pair of this workgroup: its loads go out with the ones above, ahead of the first wait
    int pr = blockIdx.x;
    ChunkItem it{};
    if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_new + row0, rows);   // uniform per workgroup
    // `done` is identical on every rank (r.r is bit-identical), so either all ranks exchange or none does
    if (__syncthreads_or(done | had_err)) return;
    while (pr < npairs) {
        chunk_publish(mv, chan, epoch, cpr, apv.Sr, it, lds);
        pr += gridDim.x;
        if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_new + row0, rows);
    }
    const int ok = chunk_wait_all(mv, epoch, npairs, timeout_ticks, err);
    if (!__syncthreads_and(ok)) return;
    // the row's Ap element first, the partials behind it: both loads are in flight together (the other way round the
    // partial is consumed -- waited for -- before the Ap load is even issued: one more memory round trip for wave 0)
    double ap_i = 0.0;
    if (in) {
        const int q = (P > 1) ? seg_owner(apv, i) : 0;
        ap_i = chunk_read_ap(mv, chan, epoch, q, i - q * apv.n_loc);
    }
    const double cs = chunk_read_partials(mv, chan, epoch, cpr, apv.Sr, npairs);
    const double conj = block_sum<4>(cs, lds);                       // bit-identical on every rank (cg.cc:106)
    const double alpha = safeguarded_alpha(rsold, conj);             // cg.cc:107
    double rr = 0.0;
    if (in) {
        const double rn = fma(-alpha, ap_i, r_i);                     // cg.cc:113
        r[i] = rn;
        rr = rn * rn;                                                 // cg.cc:116
    }
    if (own) x[li] = fma(alpha, p_i, x_i);                            // cg.cc:110
    rr = block_sum<4>(rr, lds);
    if (tid == 0) r[rv.Sr + blockIdx.x] = rr;
}

// ------------------------------------------------------------------------------------------------
// Tagged words: the exchange without flags and without fences (cfg.p2p_tagged; the transport self-test decides whether a
// node may use it).  Every double travels as two 8-byte words {32 bits of the value | 32 bits of the epoch}, each written
// with ONE relaxed system-scope atomic store and read with relaxed system-scope atomic loads.  A reader that sees the
// epoch in a word knows that the word's other half belongs to the same store (8-byte atomics are single-copy atomic), and
// it needs nothing else: no word depends on the order in which any other word arrives, so there is no release, no flag,
// no acquire, and no barrier between "the data is there" and "use it" -- the chain is store -> (xGMI) -> the poll that
// hits.  A slot still holds the words of epoch e-2 until they are overwritten: the tag is compared for equality.
// What a tagged slot can hold, and why none of it passes for the current epoch (round 4):
//   * zeros -- the mailbox is zero-filled when it is created, and the tagged region is zero-filled again whenever it is laid
//     out anew (a new problem geometry, the self-test): p2p_tag() is never 0;
//   * tagged words of epoch e-2, e-4, ... of the SAME layout -- every position a reader looks at is rewritten in every
//     epoch of its parity, so the newest stale tag is that of e-2, and p2p_tag(e) != p2p_tag(e-2) for every e;
//   * nothing else: the plain-double all-gathers of the set-up and verification phases (k_mailbox_allgather: x0 / x /
//     the initial Ap) have a slot region and an epoch counter of their own in tagged mode (channel 0), so no plain double is
//     ever stored where a tagged reader polls.  (Round 3 shared channel 1 and relied on "no finite double looks like a tag
//     during the first 2^19 epochs of a context".)
// Twice the bytes on the wire (64 KiB per rank at N = 32768): irrelevant for a latency-bound exchange.
// ------------------------------------------------------------------------------------------------
// polls both words until they carry `tag` (bounded by the wall clock); *ok = 0 if the wait expired
__device__ __forceinline__ double tagged_load(const unsigned long long *src, unsigned tag, long long timeout_ticks, int *err, int *ok)
{
    const long long t0 = wall_clock64();
    unsigned long long w0, w1;
    for (;;) {
        w0 = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        w1 = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(w0 >> 32) == tag && (unsigned)(w1 >> 32) == tag) break;
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > timeout_ticks) {
            *ok = 0;
            atomicExch(err, 1);
            break;
        }
    }
    return __longlong_as_double((long long)((w0 & 0xffffffffull) | (w1 << 32)));
}

__device__ __forceinline__ unsigned long long *tagged_slot(const MailboxView &mv, int owner, int chan, unsigned long long epoch, int q)
{
    return reinterpret_cast<unsigned long long *>(mv.base[owner] + mv.data_off[chan] +
                                                  ((long)(epoch & 1) * mv.nranks + q) * mv.slot_bytes[chan]);
}

// The tagged-word form of k_update_xr_p2p: same pairs, same chunk arithmetic (chunk_pair / chunk_dot: the same bits), same
// fold order of the partials; only how the bytes are handed over differs.
template <bool SELFTEST>
__global__ __launch_bounds__(256) void k_update_xr_p2p_tagged(int n, int rows, int row0, const double *__restrict__ p_new,
                                                               SegView apv, int cpr, MailboxView mv, int chan,
                                                               unsigned long long epoch, double *__restrict__ x, SegView rv,
                                                               Scalars *sc, int parity_rs, long long timeout_ticks, int *err,
                                                               const double *__restrict__ ap_src, int split, long part_stride,
                                                               double *__restrict__ vals, double *__restrict__ sums)
{
    __shared__ double lds[4];
    const int tid = threadIdx.x, P = mv.nranks, me = mv.rank;
    const unsigned tag = p2p_tag(epoch);
    int done = 0;
    double rsold = 0.0, r_i = 0.0, p_i = 0.0, x_i = 0.0;
    const int had_err = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int i = blockIdx.x * 256 + tid;   // global row
    const int li = i - row0;
    const bool in = i < n, own = in && li >= 0 && li < rows;
    if constexpr (!SELFTEST) {
        done = sc->done;
        rsold = sc->rs[parity_rs];
        if (in) r_i = rv.base[i];
        if (own) { p_i = p_new[i]; x_i = x[li]; }
    }
    const int npairs = P * cpr;
    int pr = blockIdx.x;
    ChunkItem it{};
    if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_new + row0, rows);
    if (__syncthreads_or(done | had_err)) return;
    while (pr < npairs) {
        const double d = chunk_dot<4>(it.pp, it.a, lds);
        unsigned long long *out = tagged_slot(mv, it.peer, chan, epoch, me);
        {
            // A lane holds the pair of rows (2L, 2L+1) of its wave's 128 rows; transposed through the wave so that one store
            // instruction covers 64 consecutive elements = 1 KiB without holes (whole 64-byte requests instead of half-masked
            // ones): lane L stores element L, then element 64 + L.
            const int lane = tid & 63, src = lane >> 1;
            const bool odd = (lane & 1) != 0;
            const double x1 = __shfl(it.a.x, src, 64), y1 = __shfl(it.a.y, src, 64);
            const double x2 = __shfl(it.a.x, 32 + src, 64), y2 = __shfl(it.a.y, 32 + src, 64);
            const int row_a = it.row - 2 * lane + lane, row_b = row_a + 64;
            if (row_a < apv.Sr) tagged_store(out + 2 * row_a, odd ? y1 : x1, tag);
            if (row_b < apv.Sr) tagged_store(out + 2 * row_b, odd ? y2 : x2, tag);
        }
        if (tid == 0) tagged_store(out + 2 * (apv.Sr + it.c), d, tag);
        pr += gridDim.x;
        if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_new + row0, rows);
    }
    // Every thread polls the two words of its own Ap element and -- the first P*cpr threads -- of one chunk partial, both in
    // the same loop: all four loads of a round are in flight together (one after the other, the first wave of every
    // workgroup paid two memory round trips where one does).
    int ok = 1;
    double cs = 0.0, ap_i = 0.0;
    {
        const unsigned long long *wa = nullptr, *wp = nullptr;
        if (in) {
            const int q = (P > 1) ? seg_owner(apv, i) : 0;
            wa = tagged_slot(mv, me, chan, epoch, q) + 2 * (i - q * apv.n_loc);
        }
        if (tid < npairs) {
            const int q = tid / cpr, c = tid - q * cpr;
            wp = tagged_slot(mv, me, chan, epoch, q) + 2 * (apv.Sr + c);
        }
        const unsigned long long *dummy = tagged_slot(mv, me, chan, epoch, 0);   // a mapped address for the loads nobody needs
        bool need_a = wa != nullptr, need_p = wp != nullptr;
        const long long t0 = wall_clock64();
        while (need_a || need_p) {
            const unsigned long long *pa = need_a ? wa : dummy, *pp = need_p ? wp : dummy;
            const unsigned long long a0 = __hip_atomic_load(pa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long a1 = __hip_atomic_load(pa + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long p0 = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long p1 = __hip_atomic_load(pp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (need_a && (unsigned)(a0 >> 32) == tag && (unsigned)(a1 >> 32) == tag) {
                ap_i = __longlong_as_double((long long)((a0 & 0xffffffffull) | (a1 << 32)));
                need_a = false;
            }
            if (need_p && (unsigned)(p0 >> 32) == tag && (unsigned)(p1 >> 32) == tag) {
                cs = __longlong_as_double((long long)((p0 & 0xffffffffull) | (p1 << 32)));
                need_p = false;
            }
            if (!(need_a || need_p)) break;
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout_ticks) {               // bounded: give up, tell the host
                ok = 0;
                atomicExch(err, 1);
                break;
            }
        }
        for (int f = tid + 256; f < npairs; f += 256) {              // more than 256 partials (n > 131072): the rest, same order
            const int q = f / cpr, c = f - q * cpr;
            cs += tagged_load(tagged_slot(mv, me, chan, epoch, q) + 2 * (apv.Sr + c), tag, timeout_ticks, err, &ok);
        }
    }
    if (!__syncthreads_and(ok)) return;
    const double conj = block_sum<4>(cs, lds);                       // bit-identical on every rank (cg.cc:106)
    if constexpr (SELFTEST) {
        if (in) vals[i] = ap_i;
        if (tid == 0) sums[blockIdx.x] = conj;
    } else {
        const double alpha = safeguarded_alpha(rsold, conj);         // cg.cc:107
        double rr = 0.0;
        if (in) {
            const double rn = fma(-alpha, ap_i, r_i);                 // cg.cc:113
            rv.base[i] = rn;
            rr = rn * rn;                                             // cg.cc:116
        }
        if (own) x[li] = fma(alpha, p_i, x_i);                        // cg.cc:110
        rr = block_sum<4>(rr, lds);
        if (tid == 0) rv.base[rv.Sr + blockIdx.x] = rr;
    }
}

// The exchange of k_update_xr_p2p alone, on a pattern: every thread stores the Ap element it read for its row to
// vals[i], every workgroup the folded partials to sums[blockIdx.x].  cgx_p2p_selftest compares both with what every rank
// must have sent.
__global__ __launch_bounds__(256) void k_chunk_exchange_selftest(int n, int rows, int row0, const double *__restrict__ p_like,
                                                                  SegView apv, int cpr, MailboxView mv, int chan,
                                                                  unsigned long long epoch, long long timeout_ticks, int *err,
                                                                  const double *__restrict__ ap_src, int split, long part_stride,
                                                                  double *__restrict__ vals, double *__restrict__ sums)
{
    __shared__ double lds[4];
    const int tid = threadIdx.x, P = mv.nranks;
    const int had_err = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int i = blockIdx.x * 256 + tid;
    const int npairs = P * cpr;
    int pr = blockIdx.x;
    ChunkItem it{};
    if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_like + row0, rows);
    if (__syncthreads_or(had_err)) return;
    while (pr < npairs) {
        chunk_publish(mv, chan, epoch, cpr, apv.Sr, it, lds);
        pr += gridDim.x;
        if (pr < npairs) it = chunk_fetch(pr, cpr, ap_src, split, part_stride, apv.Sr, p_like + row0, rows);
    }
    const int ok = chunk_wait_all(mv, epoch, npairs, timeout_ticks, err);
    if (!__syncthreads_and(ok)) return;
    const double cs = chunk_read_partials(mv, chan, epoch, cpr, apv.Sr, npairs);
    if (i < n) {
        const int q = (P > 1) ? seg_owner(apv, i) : 0;
        vals[i] = chunk_read_ap(mv, chan, epoch, q, i - q * apv.n_loc);
    }
    const double conj = block_sum<4>(cs, lds);
    if (tid == 0) sums[blockIdx.x] = conj;
}

// ------------------------------------------------------------------------------------------------
// host-side launchers
// ------------------------------------------------------------------------------------------------
static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

void seg_finalize(SegView *sv)
{
    sv->seg_gap = sv->S - sv->n_loc;
    // Round-up magic for dividends c < 2^31 (Granlund-Montgomery): s = ceil(log2 d), m = floor(2^(31+s)/d) + 1
    // fits 32 bits and floor(c/d) == (c*m) >> (31+s) == umulhi(c,m) >> (s-1) exactly.  d == 1 is flagged by
    // div_shift == 32 (quotient = c); n_loc == 0 (N < P: every column belongs to the last rank) is handled by
    // seg_owner itself and needs no magic.
    const unsigned d = sv->n_loc > 0 ? (unsigned)sv->n_loc : 1u;
    if (d == 1) {
        sv->div_magic = 0;
        sv->div_shift = 32;
        return;
    }
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    sv->div_magic = (unsigned)(((1ull << (31 + s)) / d) + 1);
    sv->div_shift = s - 1;
}

GemvPlan plan_gemv(int variant, int rows, int n, long lda, bool allow_split)
{
    GemvPlan pl{};
    pl.split = 1;
    const int ncols = (int)lda;
    pl.ncols = (n + 1) & ~1;
    if (pl.ncols > ncols) pl.ncols = ncols;
    pl.waves = 4;
    if (variant <= 0) {
        // default: column-split; the shape comes from measurements on MI355X (profiles/r01_k1_*, tools/ab_k1.py,
        // tools/small_n.py, tools/shard_plain.py):
        //  - blocks that stream from HBM (more than the 256 MiB of Infinity Cache): 8 rows per workgroup, two
        //    steps in flight per wave -- best or tied from 6144^2 up to 32768^2 and on 4096 x 32768 shards;
        //  - blocks that fit in the Infinity Cache (N <= ~5700 on one GPU): the launch is latency-bound, not
        //    bandwidth-bound, and twice as many 4-row workgroups are 8-11 % faster (N = 2048 ... 5120).
        pl.variant = 1;
        pl.nt = 1;
        const double block_bytes = 8.0 * (double)rows * (double)ncols;
        //  - row blocks of 8192 rows or fewer that still stream from HBM (the shards of a 4- and 8-GPU run of N = 32768):
        //    the one-round form with 4 rows x 4 steps, 155.5 us against 157.8 on 4096 x 32768 and 306.7 against 309.2
        //    on 8192 x 32768 (profiles/r02_k1_shards/); up to 16384 rows the one-round form of (8,2): 605.2 against 608.5.
        if (rows >= 2048 && block_bytes > 256.0 * 1024 * 1024) {
            if (rows <= 16384 && allow_split) {
                // multi-rank run: whoever consumes Ap adds the column pieces itself (k_prefold_ap or the fused P2P update) and
                // nobody folds K1's own partials, so their number does not matter: (8,2) one-round form, 8 pieces = one per
                // XCD.  Round 3, rocprofv3, loopback shards of N=32768: 153.3 / 302.8 / 603.0 us at P = 8 / 4 / 2 against
                // 156.1 / 307.2 / 604.9 unsplit; 4 pieces at P=4: 304.1, 2 or 4 at P=2: 604.0 / 605.8.
                pl.R = 8; pl.U = 2; pl.light = 1;
                pl.split = 8;
            }
            else if (rows <= 8192) { pl.R = 4; pl.U = 4; pl.light = 1; }
            else if (rows <= 16384) { pl.R = 8; pl.U = 2; pl.light = 1; }
            else { pl.R = 8; pl.U = 2; }
        }
        else if (rows >= 512) { pl.R = 4; pl.U = 2; }
        else { pl.R = 2; pl.U = 4; }
    } else {
        // explicit shape: variant*10000 + R*100 + U*10 + d   (e.g. 10821, 20441, 11611); d = 2 selects the one-round
        // form of the column-split kernel (10822, 10842, 10442, 10482), any other digit the ordinary one
        pl.variant = variant / 10000;
        pl.R = (variant / 100) % 100;
        pl.U = (variant / 10) % 10;
        pl.nt = 1;
        const int d = variant % 10;       // 2: the one-round form; 3 / 4 / 5: the same with the columns split 2 / 4 / 8 ways
        pl.light = d >= 2 && d <= 5;
        pl.split = d == 3 ? 2 : (d == 4 ? 4 : (d == 5 ? 8 : 1));
    }
    if (pl.variant == 2) {
        pl.rows_per_wg = pl.R * pl.waves;
    } else {
        pl.variant = 1;
        pl.rows_per_wg = pl.R;
    }
    pl.grid = ceil_div(rows, pl.rows_per_wg);
    if (pl.grid < 1) pl.grid = 1;   // a shard without rows still runs the iteration head and stores p
    if (pl.split < 1 || pl.variant != 1 || !pl.light) pl.split = 1;
    pl.grid *= pl.split;
    if (pl.variant != 1) pl.light = 0;   // (an explicit one-round shape is honoured at any grid: it is correct, just not one round)
    return pl;
}

namespace {

struct GemvArgs {
    const double *A;
    long lda;
    int rows, row0;
    const double *v;
    double *p_new;
    SegView sv;
    double *Ap, *partials;
    Scalars *sc;
    int k;
    double tol;
    hipEvent_t e0 = nullptr, e1 = nullptr;   // optional: bound to the dispatch (kernel begin / end)
    int split = 1;                           // one-round form only: column pieces per row group
    long ap_stride = 0;
};

template <int R, int U, int MODE>
hipError_t launch_shape(const GemvPlan &pl, const GemvArgs &g, hipStream_t s)
{
    if (pl.variant == 2)
        hipExtLaunchKernelGGL((k_gemv_ldsp<R, U, 4, MODE>), dim3(pl.grid / pl.split), dim3(256), 0, s, g.e0, g.e1, 0, g.A, g.lda, g.rows,
                              g.row0, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol);
    else
        hipExtLaunchKernelGGL((k_gemv_colsplit<R, U, 4, MODE>), dim3(pl.grid / pl.split), dim3(256), 0, s, g.e0, g.e1, 0, g.A, g.lda,
                              pl.ncols, g.rows, g.row0, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol, 1, 0L);
    return hipGetLastError();
}

template <int R, int U, int MODE>
hipError_t launch_light(const GemvPlan &pl, const GemvArgs &g, hipStream_t s)
{
    if (g.partials)
        hipExtLaunchKernelGGL((k_gemv_colsplit<R, U, 4, MODE, true, true>), dim3(pl.grid / pl.split * g.split), dim3(256), 0, s, g.e0, g.e1, 0, g.A, g.lda,
                              pl.ncols, g.rows, g.row0, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol, g.split, g.ap_stride);
    else   // nobody folds this launch's per-workgroup p.Ap partials (chunked exchange)
        hipExtLaunchKernelGGL((k_gemv_colsplit<R, U, 4, MODE, true, false>), dim3(pl.grid / pl.split * g.split), dim3(256), 0, s, g.e0, g.e1, 0, g.A, g.lda,
                              pl.ncols, g.rows, g.row0, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol, g.split, g.ap_stride);
    return hipGetLastError();
}

template <int MODE>
hipError_t dispatch_gemv(const GemvPlan &pl, const GemvArgs &g, hipStream_t s)
{
    if (pl.light && pl.variant == 1) {
        if (pl.R == 8 && pl.U == 2) return launch_light<8, 2, MODE>(pl, g, s);
        if (pl.R == 8 && pl.U == 4) return launch_light<8, 4, MODE>(pl, g, s);
        if (pl.R == 4 && pl.U == 4) return launch_light<4, 4, MODE>(pl, g, s);
        if (pl.R == 2 && pl.U == 8) return launch_light<2, 8, MODE>(pl, g, s);
        if (pl.R == 16 && pl.U == 1) return launch_light<16, 1, MODE>(pl, g, s);
        return hipErrorInvalidValue;
    }
#define CGX_SHAPE(r, u) \
    if (pl.R == r && pl.U == u) return launch_shape<r, u, MODE>(pl, g, s);
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

}  // namespace

hipError_t launch_gemv_plain(const GemvPlan &pl, const double *A, long lda, int rows, const double *v_full,
                             const double *v_local, double *Ap, double *partials, Scalars *sc, hipStream_t s)
{
    GemvArgs g{A, lda, rows, (int)(v_local - v_full), v_full, nullptr, SegView{}, Ap, partials, sc, 0, 0.0};
    return dispatch_gemv<kPlain>(pl, g, s);
}

hipError_t launch_gemv_fused(const GemvPlan &pl, const double *A, long lda, int rows, int row0, const double *p_old,
                             double *p_new, SegView seg, double *Ap, double *partials, Scalars *sc, int k, double tol,
                             hipStream_t s, hipEvent_t e_start, hipEvent_t e_stop, long ap_stride)
{
    GemvArgs g{A, lda, rows, row0, p_old, p_new, seg, Ap, partials, sc, k, tol, e_start, e_stop};
    g.split = pl.split;
    g.ap_stride = ap_stride;
    if (pl.split > 1 && (!pl.light || ap_stride <= 0)) return hipErrorInvalidValue;
    return dispatch_gemv<kFusedSingle>(pl, g, s);
}

int update_xr_grid(int count)
{
    const int g = count > 0 ? ceil_div(count, 256) : 1;
    return g < kMaxVectorGrid ? g : kMaxVectorGrid;
}

hipError_t launch_update_xr(int n, int rows, int row0, const double *p_new, SegView apv, int tail_off, int tail_count,
                            double *x, SegView rv, Scalars *sc, int parity, double *partials, hipStream_t s, hipEvent_t e0,
                            hipEvent_t e1)
{
    if ((long)update_xr_grid(n) * 256 >= n)   // one row per thread: every dense problem
        hipExtLaunchKernelGGL(k_update_xr, dim3(update_xr_grid(n)), dim3(256), 0, s, e0, e1, 0, n, rows, row0, p_new, apv, tail_off,
                              tail_count, x, rv, sc, parity, partials);
    else
        hipExtLaunchKernelGGL(k_update_xr_strided, dim3(update_xr_grid(n)), dim3(256), 0, s, e0, e1, 0, n, rows, row0, p_new, apv,
                              tail_off, tail_count, x, rv, sc, parity);
    return hipGetLastError();
}

GemvPlan plan_dia(int rows, int variant)
{
    GemvPlan pl{};
    pl.variant = 3;
    pl.split = 1;
    pl.R = 1;
    pl.U = 1;
    pl.waves = 4;
    pl.rows_per_wg = 512;   // two consecutive rows per thread
    pl.grid = rows > 0 ? ceil_div(rows, 512) : 1;
    if (pl.grid > 2048) pl.grid = 2048;   // above that the workgroups stride: K3 folds at most 2048 partials per rank (1024..8192: same speed, measured)
    // light = the LDS-window form of K1b.  30001 / 30002 force the direct / the window form.  Default by size, from
    // tools/banded_bench.py: the window form is slower up to 2^24 rows (258 against 246 us there: two barriers per
    // tile, and the direct form's overlapping vector loads were never the limiter -- both sit at ~90 % of what the
    // memory system gives this mix of 7 read and 2 write streams, tools/hbm_mix_bw.hip) and 5 % faster at 2^26.
    pl.light = variant == 30002 ? 1 : (variant == 30001 ? 0 : (rows >= (1 << 25) ? 1 : 0));
    return pl;
}

namespace {

struct DiaArgs {
    DiaView dv;
    int rows, row0, n;
    long lda;
    const double *v;
    double *p_new;
    SegView sv;
    double *Ap, *partials;
    Scalars *sc;
    int k;
    double tol;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

// Group the (ascending) offsets into windows; false if they do not fit kMaxWindows.  parity = parity of the first
// global row of a tile (tiles start at multiples of 512 behind row0): window starts are made even in GLOBAL columns.
bool make_windows(const DiaView &dv, int row0_parity, DiaWindows *dw)
{
    dw->nwin = 0;
    dw->total = 0;
    int t = 0;
    while (t < dv.ndiag) {
        if (dw->nwin == kMaxWindows) return false;
        const int lo = dv.off[t];
        int hi = lo, t1 = t;
        while (t1 + 1 < dv.ndiag && dv.off[t1 + 1] - lo <= kWinGap) hi = dv.off[++t1];
        const int w = dw->nwin++;
        int start = lo;
        if (((start + row0_parity) & 1) != 0) --start;           // even global column
        int len = 512 + hi - start;                               // rows i, i+1 <= tile + 511, columns up to tile + 511 + hi
        len = (len + 1) & ~1;
        dw->start[w] = start;
        dw->len[w] = len;
        dw->base[w] = dw->total;
        dw->total += len;
        for (int u = t; u <= t1; ++u) dw->idx[u] = (short)(dw->base[w] + dv.off[u] - start);
        t = t1 + 1;
    }
    return dw->total * 8 <= 60 * 1024;
}

template <int MODE, int CH>
hipError_t launch_dia_chunk(const GemvPlan &pl, const DiaArgs &g, hipStream_t s)
{
    if (pl.light) {   // LDS windows
        DiaWindows dw;
        if (make_windows(g.dv, g.row0 & 1, &dw)) {
            hipExtLaunchKernelGGL((k_spmv_dia_lds<MODE, CH>), dim3(pl.grid), dim3(256), (unsigned)dw.total * 8u, s, g.e0, g.e1, 0,
                                  g.dv, dw, g.rows, g.row0, g.n, g.lda, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol);
            return hipGetLastError();
        }
    }
    hipExtLaunchKernelGGL((k_spmv_dia<MODE, CH>), dim3(pl.grid), dim3(256), 0, s, g.e0, g.e1, 0, g.dv, g.rows, g.row0, g.n,
                          g.lda, g.v, g.p_new, g.sv, g.Ap, g.partials, g.sc, g.k, g.tol);
    return hipGetLastError();
}

// Diagonals in flight per chunk: all of them up to 8; above that the chunk size in 5..8 that wastes the fewest
// re-loads in the last chunk.
int dia_chunk(int ndiag)
{
    if (ndiag <= 8) return ndiag < 1 ? 1 : ndiag;
    int best = 8, waste = (8 - ndiag % 8) % 8;
    for (int ch = 7; ch >= 5; --ch) {
        const int w = (ch - ndiag % ch) % ch;
        if (w < waste) { best = ch; waste = w; }
    }
    return best;
}

template <int MODE>
hipError_t dispatch_dia(const GemvPlan &pl, const DiaArgs &g, hipStream_t s)
{
    switch (dia_chunk(g.dv.ndiag)) {
    case 1: return launch_dia_chunk<MODE, 1>(pl, g, s);
    case 2: return launch_dia_chunk<MODE, 2>(pl, g, s);
    case 3: return launch_dia_chunk<MODE, 3>(pl, g, s);
    case 4: return launch_dia_chunk<MODE, 4>(pl, g, s);
    case 5: return launch_dia_chunk<MODE, 5>(pl, g, s);
    case 6: return launch_dia_chunk<MODE, 6>(pl, g, s);
    case 7: return launch_dia_chunk<MODE, 7>(pl, g, s);
    default: return launch_dia_chunk<MODE, 8>(pl, g, s);
    }
}

}  // namespace

hipError_t launch_spmv_dia_plain(const GemvPlan &pl, const DiaView &dv, int rows, int row0, int n, long lda,
                                 const double *v_full, double *Ap, double *partials, Scalars *sc, hipStream_t s)
{
    return dispatch_dia<kPlain>(pl, DiaArgs{dv, rows, row0, n, lda, v_full, nullptr, SegView{}, Ap, partials, sc, 0, 0.0}, s);
}

hipError_t launch_spmv_dia_fused(const GemvPlan &pl, const DiaView &dv, int rows, int row0, int n, long lda,
                                 const double *p_old, double *p_new, SegView seg, double *Ap, double *partials,
                                 Scalars *sc, int k, double tol, hipStream_t s, hipEvent_t e_start, hipEvent_t e_stop)
{
    return dispatch_dia<kFusedSingle>(
        pl, DiaArgs{dv, rows, row0, n, lda, p_old, p_new, seg, Ap, partials, sc, k, tol, e_start, e_stop}, s);
}

int lap2d_offsets(int size, int *off)
{
    const int inc = (int)floor(sqrt((double)size));   // cg.cc:175
    const int cand[5] = {-(inc + 1), -1, 0, 1, inc + 1};
    int nd = 0;
    for (int c : cand)
        if (c > -size && c < size) off[nd++] = c;
    return nd;
}

static inline int capped_grid(long total, int cap)
{
    const long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : (g < cap ? g : cap));
}

hipError_t launch_dia_generate_lap2d(double *vals, const DiaView &dv, int size, int row0, int rows, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    const int inc = (int)floor(sqrt((double)size));
    hipLaunchKernelGGL(k_dia_generate_lap2d, dim3(capped_grid((long)dv.ndiag * rows, 8192)), dim3(256), 0, s, vals, dv, size,
                       row0, rows, inc);
    return hipGetLastError();
}

hipError_t launch_dia_mark(const double *A, long lda, int n, int row0, int rows, unsigned char *flags, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dia_mark, dim3(capped_grid((long)rows * n, 16384)), dim3(256), 0, s, A, lda, n, row0, rows, flags);
    return hipGetLastError();
}

hipError_t launch_dia_pack(const double *A, long lda, int n, int row0, int rows, double *vals, const DiaView &dv,
                           hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_dia_pack, dim3(capped_grid((long)dv.ndiag * rows, 8192)), dim3(256), 0, s, A, lda, n, row0, rows,
                       vals, dv);
    return hipGetLastError();
}

hipError_t launch_update_xr_p2p(int n, int rows, int row0, const double *p_new, SegView apv, int cpr,
                                const MailboxView &mv, int chan, unsigned long long epoch, double *x, SegView rv, Scalars *sc,
                                int parity, long long timeout_ticks, int *err, hipStream_t s, const double *ap_src, int split,
                                long stride, hipEvent_t e0, hipEvent_t e1)
{
    if (cpr != chunks_per_rank(apv.Sr) || (long)mv.nranks * cpr > kMaxChunkFlags || split < 1 || split > kMaxSplit)
        return hipErrorInvalidValue;
    if (mv.tagged) {
        hipExtLaunchKernelGGL((k_update_xr_p2p_tagged<false>), dim3(update_xr_grid(n)), dim3(256), 0, s, e0, e1, 0, n, rows, row0, p_new,
                              apv, cpr, mv, chan, epoch, x, rv, sc, parity, timeout_ticks, err, ap_src, split, stride,
                              (double *)nullptr, (double *)nullptr);
        return hipGetLastError();
    }
    hipExtLaunchKernelGGL(k_update_xr_p2p, dim3(update_xr_grid(n)), dim3(256), 0, s, e0, e1, 0, n, rows, row0, p_new, apv, cpr, mv,
                          chan, epoch, x, rv, sc, parity, timeout_ticks, err, ap_src, split, stride);
    return hipGetLastError();
}

hipError_t launch_chunk_exchange_selftest(int n, int rows, int row0, const double *p_like, SegView apv, int cpr,
                                          const MailboxView &mv, int chan, unsigned long long epoch, long long timeout_ticks,
                                          int *err, const double *ap_src, int split, long stride, double *vals, double *sums,
                                          hipStream_t s)
{
    if (cpr != chunks_per_rank(apv.Sr) || (long)mv.nranks * cpr > kMaxChunkFlags || split < 1 || split > kMaxSplit)
        return hipErrorInvalidValue;
    if (mv.tagged) {
        hipLaunchKernelGGL((k_update_xr_p2p_tagged<true>), dim3(update_xr_grid(n)), dim3(256), 0, s, n, rows, row0, p_like, apv, cpr, mv,
                           chan, epoch, (double *)nullptr, SegView{}, (Scalars *)nullptr, 0, timeout_ticks, err, ap_src, split, stride,
                           vals, sums);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_chunk_exchange_selftest, dim3(update_xr_grid(n)), dim3(256), 0, s, n, rows, row0, p_like, apv, cpr, mv,
                       chan, epoch, timeout_ticks, err, ap_src, split, stride, vals, sums);
    return hipGetLastError();
}

hipError_t update_xr_p2p_resident_limit(int device, bool tagged, int *workgroups)
{
    int per_cu = 0, cus = 0;
    hipError_t e = tagged ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_update_xr_p2p_tagged<false>, 256, 0)
                          : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_update_xr_p2p, 256, 0);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (e != hipSuccess) return e;
    *workgroups = per_cu * cus;
    return hipSuccess;
}

hipError_t launch_prefold_ap(const double *parts, int split, long stride, int rows, int Sr, const double *p_loc, double *dst,
                             double *tail, const Scalars *sc, hipStream_t s)
{
    if (split < 1 || split > kMaxSplit) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_prefold_ap, dim3(chunks_per_rank(Sr)), dim3(256), 0, s, parts, split, stride, rows, Sr, p_loc, dst, tail, sc);
    return hipGetLastError();
}

hipError_t launch_close_iteration(Scalars *sc, SegView seg, int k, double tol, hipStream_t s)
{
    hipLaunchKernelGGL(k_close_iteration, dim3(1), dim3(256), 0, s, sc, seg, k, tol);
    return hipGetLastError();
}

hipError_t launch_reduce_partials(const double *partials, int n, double *out, hipStream_t s)
{
    hipLaunchKernelGGL((k_reduce_partials<1>), dim3(1), dim3(256), 0, s, partials, n, out);
    return hipGetLastError();
}

hipError_t launch_reduce_partials3(const double *partials, int n, double *out3, hipStream_t s)
{
    hipLaunchKernelGGL((k_reduce_partials<3>), dim3(1), dim3(256), 0, s, partials, n, out3);
    return hipGetLastError();
}

hipError_t launch_solve_begin_zero(int n, long lda, const double *b_full, double *x, SegView rv, double *p0, double *p1, double *apg,
                                   long apg_count, Scalars *sc, int *err, hipStream_t s)
{
    hipLaunchKernelGGL(k_solve_begin_zero, dim3(update_xr_grid(n)), dim3(256), 0, s, n, lda, b_full, x, rv, p0, p1, apg, apg_count, sc, err);
    return hipGetLastError();
}

hipError_t launch_solve_end(int n, const double *Ax, const double *b, const double *x, const Scalars *sc, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_solve_end, dim3(1), dim3(1024), 0, s, n, Ax, b, x, sc, out);
    return hipGetLastError();
}

hipError_t launch_init_residual(int n, const double *b_full, SegView apv, SegView rv, double *partials, hipStream_t s)
{
    hipLaunchKernelGGL(k_init_residual, dim3(update_xr_grid(n)), dim3(256), 0, s, n, b_full, apv, rv, partials);
    return hipGetLastError();
}

hipError_t launch_copy_doubles(double *dst, const double *src, long count, hipStream_t s)
{
    if (count <= 0) return hipSuccess;
    int grid = ceil_div(count, 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_copy_doubles, dim3(grid), dim3(256), 0, s, dst, src, count);
    return hipGetLastError();
}

hipError_t launch_unpack_segments(SegView seg, double *v_full, long lda, hipStream_t s)
{
    int grid = ceil_div(lda, 256);
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(k_unpack_segments, dim3(grid), dim3(256), 0, s, seg, v_full, lda);
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

hipError_t launch_fill_hash(double *A, long lda, int n, int row0, int rows, unsigned long long seed, int symmetric, double diag,
                            hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    const long total = (long)rows * (lda / 2);
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(k_fill_hash, dim3(grid), dim3(256), 0, s, A, lda, n, row0, rows, hash_mix64(seed), symmetric, diag);
    return hipGetLastError();
}

hipError_t launch_coo_assign(double *A, long lda, const DiaView *dv, double *dia_vals, int n, int row0, int rows,
                             const int *I, const int *J, const double *a, long nz, int sym, unsigned char *win, hipStream_t s)
{
    if (nz <= 0 || rows <= 0) return hipSuccess;
    const int grid = capped_grid(nz, 4096);
    if (dv) {
        hipLaunchKernelGGL((k_coo_assign<true, 0>), dim3(grid), dim3(256), 0, s, dia_vals, 0L, *dv, n, row0, rows, I, J, a, nz, sym, win);
        hipLaunchKernelGGL((k_coo_assign<true, 1>), dim3(grid), dim3(256), 0, s, dia_vals, 0L, *dv, n, row0, rows, I, J, a, nz, sym, win);
        hipLaunchKernelGGL((k_coo_assign<true, 2>), dim3(grid), dim3(256), 0, s, dia_vals, 0L, *dv, n, row0, rows, I, J, a, nz, sym, win);
    } else {
        hipLaunchKernelGGL((k_coo_assign<false, 0>), dim3(grid), dim3(256), 0, s, A, lda, DiaView{}, n, row0, rows, I, J, a, nz, sym, win);
        hipLaunchKernelGGL((k_coo_assign<false, 1>), dim3(grid), dim3(256), 0, s, A, lda, DiaView{}, n, row0, rows, I, J, a, nz, sym, win);
        hipLaunchKernelGGL((k_coo_assign<false, 2>), dim3(grid), dim3(256), 0, s, A, lda, DiaView{}, n, row0, rows, I, J, a, nz, sym, win);
    }
    return hipGetLastError();
}

hipError_t launch_mailbox_allgather(const MailboxView &mv, int chan, unsigned long long epoch, const double *src,
                                    int count, int tail_off, int tail_n, double *dst, long dst_stride, int sum_off,
                                    int copy_self, long long timeout_ticks, int *err, hipStream_t s)
{
    hipLaunchKernelGGL(k_mailbox_allgather, dim3(mv.nranks), dim3(256), 0, s, mv, chan, epoch, src, count, tail_off, tail_n,
                       dst, dst_stride, sum_off, copy_self, timeout_ticks, err);
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
