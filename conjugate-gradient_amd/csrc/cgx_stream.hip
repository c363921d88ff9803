// cgx_stream.hip -- the whole CG loop of code/MPI/cg.cc:95-137 as ONE persistent kernel for matrices that do NOT fit on the chip:
// 4096 < n <= 16384 on one GPU (BASELINE config 2, N = 10000; the reference's largest published size, N = 8192,
// results/strong_scaling.txt:22).
//
// There the per-launch path streams A at the rate the memory system gives (K1 at 0.89-0.91 of 8 TB/s, profiles/r05_midsize/),
// and what an iteration loses is everything around K1: the update kernel K3 (4.8-5.0 us of latency chain), two kernel
// boundaries and two launch ramps -- 6-7 % of an 80 us iteration at N = 8192.  This kernel keeps cgx_resident.hip's structure
// (r and p replicated in every workgroup, Ap exchanged as tagged words, no K3, no kernel boundary, no grid barrier) and
// streams the rows -- all but the few that the chip holds beside the vectors:
//
//   grid  = G <= 256 workgroups of 512 threads (8 waves, 2 per SIMD, up to 256 registers each), one per CU, all resident;
//           workgroup g owns the R consecutive rows g R ... g R + R - 1, R = ceil(n / 256) rounded up so that the streamed
//           rows are a whole number of batches; its first RL rows are copied into the LDS and the next RG into registers once
//           per launch (the thread keeps the pairs it multiplies), the other R - RL - RG are streamed every iteration:
//           S = 5 (n <= 5120): 2 + 7 of 20 rows; S = 6: 2 + 5 of 24; S = 7: 1 + 4 of 28; S = 8 (n <= 8192): 1 + 3 of 32;
//           S = 9: 1 + 2 of 36; S = 10 (n <= 10240): 0 + 1 of 40; S = 11: 0 + 1 of 44; above: none (the LDS holds the parked
//           Ap, 8 KB S, and nothing else of that size; a row costs 4 S registers);
//   state = r, p in registers, replicated in every workgroup: thread t owns the column pairs {1024 s + 2 t, + 1}, s < S =
//           ceil(n / 1024) -- the same columns whose entries of A it streams, so the GEMV needs no vector traffic at all;
//   A     = streamed through a RING of RB x S 16-byte registers per thread (RB rows of the thread's columns): a slot is
//           consumed (two FMAs) and at once re-issued for the row RB further on -- non-temporal BUFFER loads (descriptor of the
//           workgroup's row block in SGPRs, row and column step as a scalar offset, the thread's position ONE 32-bit register
//           for all of them; rows and columns outside the block read as 0 by the range check), issued through the compiler's
//           builtins, so every wait on streamed data is the compiler's own vmcnt bookkeeping (no hand-written waits: ADVICE r4).  The rows do not depend on p, so the ring simply
//           wraps around: while the workgroups exchange Ap and update r and p, the first RB rows of the NEXT iteration are
//           already in flight (64 KB per CU at N = 8192) and the memory pipe does not run dry across the iteration boundary;
//           every workgroup begins its sweep at a batch of its own (phi, below: 256 streams in step meet in the same memory
//           channels for some row pitches otherwise);
//   one iteration (cg.cc:96-137) =
//     Ap_sub = A_sub p            every wave sweeps its 128 columns of each 1024-column step of every row; per batch of RB rows
//                                 one wave reduction (v_permlane swaps + DPP), per row 8 wave partials in LDS    cg.cc:100-102
//     publish Ap_sub              R tagged doubles per workgroup (cgx_tagged.h)
//     gather Ap                   every thread polls one watched word, then the tagged words of its 2 S columns in chunks of
//                                 2 or 4 (agent-scope buffer loads: one 32-bit lane offset, parity and column step scalar; once
//                                 the watched word is there the others are L2 hits: small chunks cost no time, large ones
//                                 cost the registers of a row of A)
//     p.Ap, alpha, x, r, r.r, break test, beta, p: as cgx_resident.hip, every workgroup over the whole vectors, same order:
//                                 bit-identical everywhere                                                         cg.cc:105-132
//
// The barriers inside an iteration hand over LDS words only and are `s_waitcnt lgkmcnt(0); s_barrier` (lds_barrier):
// __syncthreads() would wait for vmcnt(0), i.e. for the prefetched rows of the next iteration, in front of every publish.
// State between launches, waits bounded by the wall clock, error word, test hooks: exactly cgx_resident.hip's (same
// ResidentArgs), so the host code (resident_steps, cgx_solve.cpp) is the same for both kernels.
// The arithmetic per element is the reference's; only the summation order of the dot products is this kernel's own (fixed,
// so a solve is bitwise reproducible).  No MFMA (0.25 flop/byte), no floating-point atomics.
#include "cgx_kernels.h"
#include "cgx_device.h"
#include "cgx_tagged.h"

namespace cgx {

namespace {

constexpr int kStrThreads = 512, kStrWaves = 8;

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#ifndef CGX_STREAM_AUX
#define CGX_STREAM_AUX 2
#endif
constexpr int kAuxNt = CGX_STREAM_AUX;           // 2 = nt: streamed once, do not keep (tools/exp_mall_policy.sh builds the others)
constexpr int kAuxSc1 = 16;                      // sc1: agent scope (loads past the L1, stores written through)
// (a poll is an sc1 load as well; what makes the compiler re-issue it every time round a loop is the `asm volatile("" ::: "memory")`
// in front of it -- the intrinsic's own volatile bit would turn it into a system-scope sc0 sc1 load)

__device__ __forceinline__ d2 as_d2(u4 w) { return __builtin_bit_cast(d2, w); }
__device__ __forceinline__ bool tag_ok(const u4 &w, unsigned tag) { return ((w.y ^ tag) | (w.w ^ tag)) == 0; }

// v + v(lane ^ 16) and so on down to groups of 1: the sum over each half wave, every lane of the half gets it; the 16-lane
// level by v_permlane16_swap (VALU) instead of a ds_bpermute round trip (same pairing as group_sum<32>, same bits)
__device__ __forceinline__ __attribute__((unused)) double half_wave_sum_swap(double v)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
    v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    return group_sum<16>(v);
}

// the row sums of a batch over the wave: afterwards v[0] of lane L is the total of row L / (64 / RB)
template <int RB>
__device__ __forceinline__ int batch_sum(double (&v)[RB], int lane)
{
    if constexpr (RB == 2) {
        v[0] = half_wave_sum_swap(swap_add<false>(v[0], v[1]));
        return lane >> 5;
    } else {
        return wave_sum_rows_swap<RB>(v, lane);
    }
}

// S = column steps of 1024 (n <= 1024 S); RB = rows per batch = depth of the ring in rows (a power of two: the batch's row
// sums are reduced together); CH = column steps per gather chunk (2 CH tagged words in flight per thread).
// Registers (hipcc 7.2, tools/kernel_resources.py): p and r 8 S, the ring 4 RB S, a register row 4 S, a gather chunk 8 CH; the gathered Ap is
// parked in LDS between p.Ap and the update of r (8 KB per column step), so it costs none.
// RL + RG of a workgroup's rows (its first ones) do not stream: RL are copied into the LDS and RG into registers once per launch
// (what the LDS holds beside the parked Ap, what the 256 registers of a thread hold beside r, p and the ring: stream_rl / stream_rg).
template <int S, int RB, int CH, int RL, int RG>
__global__ __launch_bounds__(kStrThreads, 2) void k_cg_stream(ResidentArgs a)
{
    constexpr int T = kStrThreads, W = kStrWaves;
    constexpr int RES = RL + RG;                                       // rows on the chip
    constexpr int RESP = RES <= 1 ? 1 : RES <= 2 ? 2 : RES <= 4 ? 4 : RES <= 8 ? 8 : 16;   // ... their sums are reduced together: a power of two
    static_assert(RES <= 16, "at most 16 rows on the chip");
    extern __shared__ double lds_all[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n, R = a.rows_per_wg, nb = (R - RES) / RB;
    double *lds_red = lds_all;                        // [W][R]: the waves' parts of the row sums
    double *lds_dot = lds_red + W * R;                // two sets of [p.Ap | r.r] x W wave partials
    double *lds_sum = lds_dot + 4 * W;                // W doubles for the set-up's block sum
    // one word: a wait of this workgroup expired (+ pad).  Plain LDS accesses, ordered by the barriers: a `volatile` access through a
    // cast pointer becomes a FLAT instruction, and with one of those pending the compiler turns every later wait into vmcnt(0)
    int *lds_fail = reinterpret_cast<int *>(lds_sum + W);
    unsigned *lds_rec = reinterpret_cast<unsigned *>(lds_fail) + 2;   // thread 0: [watch rounds | repeated gather rounds | first wait | longest later wait] (a.rec)
    d2 *lds_ap = reinterpret_cast<d2 *>(lds_sum + W + 4);   // [S][T] pairs: the gathered Ap of this thread's columns
    d2 *lds_A = lds_ap + S * T;                             // [RL][S][T] pairs: the rows kept in the LDS
    d2 *lds_own = lds_A + RL * S * T;                       // [3][R / 2 + 2] pairs: x, r, p of the workgroup's own rows
    const int row0 = blockIdx.x * R;
    const int my_rows = min(R, n - row0);             // >= 1 by construction of the grid

    if (tid == 0) *lds_fail = 0;
    if (tid < 4) lds_rec[tid] = 0;
    if (__syncthreads_or(__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (tid == 0) tail_report(a, 1, 0, a.k0, nullptr);
        return;
    }

    // A through a buffer descriptor that spans the matrix: row i of the workgroup, column step s = scalar offset (row0 + i) pitch
    // + 8 KB s, the thread's column pair = ONE 32-bit byte offset (a second one for the last column step, whose pair may lie
    // behind the pitch: it then reads the last pair inside the pitch instead -- the product is with p = 0 there, the thread's
    // column being >= n, i.e. exactly 0).  Rows behind the last one are read as row n - 1 (never published): nothing relies
    // on the descriptor's range check, and nothing outside the matrix is ever touched.
    const unsigned long long a_bytes = (unsigned long long)n * a.lda * 8;
    const rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(a.A), 0, (int)(unsigned)(a_bytes < 0xffffffffull ? a_bytes : 0xffffffffull), 0x00020000);
    const unsigned pitch_b = (unsigned)(a.lda * 8);
    const unsigned voff = 16u * (unsigned)tid;
    unsigned voff_last;
    {
        const long c = 2 * T * (S - 1) + 2 * tid;
        voff_last = (unsigned)(8 * ((c < a.lda - 2 ? c : a.lda - 2) - 2 * T * (S - 1)));
    }
    auto a_issue = [&](int i, int s) {
        return as_d2(__builtin_amdgcn_raw_buffer_load_b128(rs_a, s == S - 1 ? voff_last : voff, (int)((unsigned)min(row0 + i, n - 1) * pitch_b + 16u * T * s), kAuxNt));
    };
    // the same with the default cache policy: for the workgroup's first a.l2_rows streamed rows, which are meant to stay in the XCD's L2
    auto a_issue_l2 = [&](int i, int s) {
        return as_d2(__builtin_amdgcn_raw_buffer_load_b128(rs_a, s == S - 1 ? voff_last : voff, (int)((unsigned)min(row0 + i, n - 1) * pitch_b + 16u * T * s), 0));
    };
    // the exchange buffer: [2 parities][1024 S tagged doubles of 16 bytes]
    const rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(a.xbuf, 0, 2 * a.xslots * 16, 0x00020000);

    // the rows on the chip: the workgroup's first RL into the LDS, the next RG into registers (once per launch; the thread keeps
    // exactly the pairs it multiplies: nobody else reads them, no barrier)
    d2 areg[RG > 0 ? RG : 1][S];
#pragma unroll
    for (int i = 0; i < RL; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s) lds_A[(i * S + s) * T + tid] = a_issue(i, s);
#pragma unroll
    for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int s = 0; s < S; ++s) areg[i][s] = a_issue(RL + i, s);

    // The ring.  It is FILLED by the sweep loop itself (a batch -1 in front of the launch's first iteration: FMAs on zeros, its
    // row sums dropped), not by loads in front of the loop: the compiler's wait in front of a slot is exact (vmcnt(RB S - 1):
    // every other slot stays in flight) only when the loads it counts reach the loop's head in ONE order -- with a separate
    // fill the scheduler issued those loads in an order of its own and the loop began every batch with vmcnt(0) (seen in the ISA).
    d2 ring[RB][S];
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
        for (int s = 0; s < S; ++s) ring[j][s] = d2{0.0, 0.0};
    int b_first = -1, pcur = 0;
    const int l2_batches = a.l2_rows / RB;          // the first streamed batches are read with the default cache policy
    // where this workgroup begins its sweep: a batch of its own (a hash of the workgroup's number).  The workgroups sweep in step, and
    // with all of them at the same place in their rows the 256 streams lie multiples of R x pitch apart -- for some pitches
    // in the same memory channels (N = 8192, pitch + 512 B: 102 instead of 75 us; N = 8704, + 256 B: 109 instead of 87).  The
    // regular patterns tried each have such a pitch; the hash had none in 30 combinations (profiles/r05_stagger/README.md).
    const int phi = a.stagger ? (int)(((unsigned)blockIdx.x * 2654435761u >> 16) % (unsigned)nb) : 0;

    // ---- state: r, p for this thread's columns (replicated in every workgroup; exactly 0 in the pad columns n ... 1024 S,
    // where the gathered Ap is a published 0 as well: no masks inside the loop); x for the workgroup's own rows only
    const double *st_x = a.in, *st_r = a.in + state_off_r(a.lda), *st_p = a.in + state_off_p(a.lda);
    const Scalars *st_sc = reinterpret_cast<const Scalars *>(a.in + state_off_sc(a.lda));
    d2 r[S], p[S];
    // x of the workgroup's own rows: the thread whose column pair lies in them keeps x and a second copy of its r and p pair,
    // advanced by the same operations (same bits): the update needs no "is this my column step" test per step then.  The three
    // pairs live in the LDS (own[0 | 1 | 2][pair index - row0 / 2] = x | r | p; R / 2 + 1 pairs at most): 12 registers of every thread
    // for something a handful of threads touch twice per iteration would cost a row of A on the chip
    const int own_n = R / 2 + 2;
    int sx = -1;
    bool ox0 = false, ox1 = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int c = 2 * T * s + 2 * tid;
        const bool ok0 = c < n, ok1 = c + 1 < n;
        r[s].x = ok0 ? st_r[c] : 0.0;
        r[s].y = ok1 ? st_r[c + 1] : 0.0;
        if (a.k0 > 0) {
            p[s].x = ok0 ? st_p[c] : 0.0;
            p[s].y = ok1 ? st_p[c + 1] : 0.0;
        } else {
            p[s] = r[s];                              // p = r, cg.cc:85
        }
        // x: only the workgroup's own rows [row0, row0 + my_rows): fewer than 1024 columns, i.e. at most one s for a thread
        const bool own0 = c >= row0 && c < row0 + my_rows, own1 = c + 1 >= row0 && c + 1 < row0 + my_rows;
        if (own0 || own1) {
            sx = s;
            ox0 = own0;
            ox1 = own1;
            d2 xo = {0.0, 0.0};
            if (own0) xo.x = st_x[c];
            if (own1) xo.y = st_x[c + 1];
            const int q = T * s + tid - (row0 >> 1);
            lds_own[q] = xo;
            lds_own[own_n + q] = r[s];
            lds_own[2 * own_n + q] = p[s];
        }
    }
    const int ap_own = (sx < 0 ? 0 : sx) * T + tid;   // where the gathered Ap of the owned pair is parked
    const bool owner = sx >= 0;
    d2 *own = lds_own + (owner ? ap_own - (row0 >> 1) : 0);
    double rsold, rs_prev;
    if (a.k0 > 0) {
        rsold = st_sc->rs[a.k0 & 1];
        rs_prev = st_sc->rs[(a.k0 + 1) & 1];
    } else {
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) v += r[s].x * p[s].x + r[s].y * p[s].y;   // rsold = r.p, cg.cc:91-92
        rsold = block_sum<W>(v, lds_sum);
        rs_prev = rsold;
    }
    rsold = uniform(rsold);     // (the same bits in every lane: kept in scalar registers across the sweep)
    rs_prev = uniform(rs_prev);

    // Every load of the state has landed before the loop is entered (a real s_waitcnt, which the compiler's bookkeeping sees): with
    // p possibly still in flight from a branch in front of the loop, the loop's FIRST use of p was a vmcnt(0) in every batch,
    // i.e. the ring drained once per batch (seen in the ISA: no vmcnt(RB S - 1) anywhere).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)

    // what this thread publishes: the row sum of row row0 + tid (tid < my_rows), or -- dealt over the spare threads of all
    // workgroups -- a ZERO for one of the pad columns n ... 1024 S - 1, so that every word a gather reads validates itself
    int pub_col = -1;
    if (tid < my_rows) {
        pub_col = row0 + tid;
    } else {
        const int i = (tid - my_rows) * (int)gridDim.x + (int)blockIdx.x;
        if (i < 2 * T * S - n) pub_col = n + i;
    }
    const unsigned pub_off = 16u * (unsigned)xpos(pub_col < 0 ? 0 : pub_col);

    const unsigned goff = (unsigned)(16 * (128 * wave + lane));   // the thread's slot inside a 1024-column step of the exchange buffer
    int k = a.k0, stop = 0;
    const int k_end = a.k0 + a.iters;
    unsigned long long epoch = a.epoch0;
    for (; k < k_end; ++k) {
        ++epoch;
        const unsigned tag = p2p_tag(epoch);
        const int par_b = (int)(epoch & 1) * a.xslots * 16;       // this epoch's half of the exchange buffer
        double *dot = lds_dot + (k & 1) * 2 * W;

        // diagnostics (CGX_RESIDENT_PROFILE=1): in four iterations of the launch (20, 21, 40, 80) every workgroup notes when it began its sweep, when it
        // had its row sums, and when it had gathered all of Ap (100-MHz wall clock), and which XCD it runs on
        const int sj = (k - a.k0 == 20) ? 0 : (k - a.k0 == 21) ? 1 : (k - a.k0 == 40) ? 2 : (k - a.k0 == 80) ? 3 : -1;
        const bool stamp = a.prof != nullptr && tid == 0 && sj >= 0;
        long long *pst = a.prof + 8 + 4 * (256 * (sj < 0 ? 0 : sj) + blockIdx.x);
        if (stamp) pst[0] = wall_clock64();

        // Ap_sub = A_sub p (cblas_dgemv, cg.cc:100-102): batch b = rows b RB ... b RB + RB - 1, the thread's columns ascending;
        // every slot of the ring is re-issued for the batch after this one as soon as it has been consumed (the fence pins
        // that order: loads return in order, so the wait in front of slot q leaves the other RB S - 1 in flight)
        for (int b = b_first; b < nb; ++b) {
            // the batch behind this one (behind the last: the first one of the next iteration), counted from the workgroup's own
            // first batch phi: the workgroups sweep in step, and with every one of them at the same place in its rows the streams
            // meet in the same memory channels for some row pitches (profiles/r05_stagger/)
            int pn = ((b + 1 < nb) ? b + 1 : 0) + phi;
            if (pn >= nb) pn -= nb;
            const int nxt = RES + pn * RB;
            double v[RB];
            if (pn < l2_batches) {                       // (workgroup-uniform: the same number of loads on either side)
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const d2 av = ring[j][s];
                        s0 = fma(av.x, p[s].x, s0);
                        s1 = fma(av.y, p[s].y, s1);
                        ring[j][s] = a_issue_l2(nxt + j, s);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    v[j] = s0 + s1;
                }
            } else {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    double s0 = 0.0, s1 = 0.0;
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const d2 av = ring[j][s];
                        s0 = fma(av.x, p[s].x, s0);
                        s1 = fma(av.y, p[s].y, s1);
                        ring[j][s] = a_issue(nxt + j, s);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    v[j] = s0 + s1;
                }
            }
            const int myrow = batch_sum<RB>(v, lane);
            if (b >= 0 && (lane & (64 / RB - 1)) == 0) lds_red[wave * R + RES + pcur * RB + myrow] = v[0];
            pcur = pn;
        }
        b_first = 0;
        // the rows on the chip, with the first batch of the next iteration already on its way
        if constexpr (RES > 0) {
            double v[RESP];
#pragma unroll
            for (int j = 0; j < RESP; ++j) {
                double s0 = 0.0, s1 = 0.0;
                if (j < RES) {
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        d2 av;
                        if (j < RL) av = lds_A[(j * S + s) * T + tid];
                        else av = areg[j < RL ? 0 : j - RL][s];
                        s0 = fma(av.x, p[s].x, s0);
                        s1 = fma(av.y, p[s].y, s1);
                    }
                }
                v[j] = s0 + s1;
            }
            const int myrow = batch_sum<RESP>(v, lane);
            if ((lane & (64 / RESP - 1)) == 0 && myrow < RES) lds_red[wave * R + myrow] = v[0];
        }
        lds_barrier();
        if (stamp) pst[1] = wall_clock64();   // (behind the barrier: the workgroup's LAST wave has its sums)
        if (pub_col >= 0 && !(k == a.k0 && (int)blockIdx.x == a.mute_wg)) {   // (mute_wg: the test of the bounded waits)
            double ap = 0.0;
            if (tid < my_rows) {
                const double *q = lds_red + tid;
                ap = ((q[0] + q[R]) + (q[2 * R] + q[3 * R])) + ((q[4 * R] + q[5 * R]) + (q[6 * R] + q[7 * R]));
            }
            const unsigned long long bits = (unsigned long long)__double_as_longlong(ap);
            const u4 w = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};
            __builtin_amdgcn_raw_buffer_store_b128(w, rs_x, pub_off, par_b, kAuxSc1);
        }

        // gather Ap: one watched word first (column 2 tid), then the thread's columns in chunks of CH steps
        double pap = 0.0;
        {
            const long long t0 = wall_clock64();
            bool expired = false;
            int wr = 0;
            // (no pause in front of the first poll or between polls: measured with s_sleep 4 / 16 / 48 between and 16 in front, 4608 ... 10000:
            // nothing beyond +- 0.3 %, profiles/r05_watch/ -- the workgroups that are done do not take bandwidth from those still streaming)
            for (;;) {
                asm volatile("" ::: "memory");
                const u4 w = __builtin_amdgcn_raw_buffer_load_b128(rs_x, goff, par_b, kAuxSc1);
                ++wr;
                if (__all(tag_ok(w, tag))) break;      // the wave goes on together: every branch of the exchange is a scalar one
                if (wall_clock64() - t0 > a.timeout_ticks) { expired = true; break; }
            }
            int gr = 0;
#pragma unroll
            for (int c0 = 0; c0 < S; c0 += CH) {
                u4 w[2 * CH];
                bool ok;
                do {
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        const int s = c0 + i < S ? c0 + i : S - 1;
                        w[2 * i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, goff, par_b + 32 * T * s, kAuxSc1);
                        w[2 * i + 1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, goff + 1024, par_b + 32 * T * s, kAuxSc1);
                    }
                    ok = true;
#pragma unroll
                    for (int i = 0; i < 2 * CH; ++i) ok = ok && tag_ok(w[i], tag);
                    ok = __all(ok);
                    if (!ok) {
                        ++gr;
                        if (wall_clock64() - t0 > a.timeout_ticks) { expired = true; ok = true; }
                    }
                } while (!ok);
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int s = c0 + i;
                    if (s >= S) continue;
                    d2 g;
                    g.x = tagged_value(w[2 * i]);
                    g.y = tagged_value(w[2 * i + 1]);
                    pap += p[s].x * g.x + p[s].y * g.y;                   // cg.cc:105-106
                    lds_ap[s * T + tid] = g;
                }
            }
            // what the exchange of this iteration cost (ResidentTail): repeated polls are counted; the span from the publish to the
            // last gathered word is timed when it took several round trips, or in the launch's first iteration (a workgroup that
            // was placed late shows there)
            if (tid == 0) {
                // (thread 0 is the only one that touches these words: no atomics)
                if (wr > 1) lds_rec[0] += (unsigned)(wr - 1);
                if (gr > 0) lds_rec[1] += (unsigned)gr;
                if (wr + gr > 5 || k == a.k0) {
                    const unsigned dt = (unsigned)(wall_clock64() - t0);
                    unsigned *w = lds_rec + (k == a.k0 ? 2 : 3);
                    if (dt > *w) *w = dt;
                }
            }
            if (stamp) {
                pst[2] = wall_clock64();
                pst[3] = __builtin_amdgcn_s_getreg(GETREG_IMMED(4 - 1, 0, 20)) & 0xf;   // XCC_ID[3:0]
            }
            if (expired) {
                atomicExch(a.err, 1);
                *lds_fail = 1;
            }
        }

        pap = wave_sum_swap(pap);
        if (lane == 0) dot[wave] = pap;
        lds_barrier();
        if (*lds_fail) {               // uniform: written in front of the barrier
            if (tid == 0) tail_report(a, 1, 0, k, lds_rec);
            return;
        }
        const double conj = ((dot[0] + dot[1]) + (dot[2] + dot[3])) + ((dot[4] + dot[5]) + (dot[6] + dot[7]));
        const double alpha = safeguarded_alpha(rsold, conj);             // cg.cc:107
        double rr = 0.0;
        if (owner) {
            const d2 g = lds_ap[ap_own], px = own[2 * own_n];
            d2 xo = own[0], rx = own[own_n];
            xo.x = fma(alpha, px.x, xo.x);                               // cg.cc:110, the workgroup's own rows
            xo.y = fma(alpha, px.y, xo.y);
            rx.x = fma(-alpha, g.x, rx.x);
            rx.y = fma(-alpha, g.y, rx.y);
            own[0] = xo;
            own[own_n] = rx;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const d2 g = lds_ap[s * T + tid];
            r[s].x = fma(-alpha, g.x, r[s].x);                           // cg.cc:113
            r[s].y = fma(-alpha, g.y, r[s].y);
            rr += r[s].x * r[s].x + r[s].y * r[s].y;                     // cg.cc:116
        }
        rr = wave_sum_swap(rr);
        if (lane == 0) dot[W + wave] = rr;
        lds_barrier();
        const double *dr = dot + W;
        const double rsnew = uniform(((dr[0] + dr[1]) + (dr[2] + dr[3])) + ((dr[4] + dr[5]) + (dr[6] + dr[7])));   // cg.cc:116-117
        if (sqrt(rsnew) < a.tol) {                                       // cg.cc:120-121: break before the p update
            rs_prev = rsnew;
            stop = 1;
            break;
        }
        const double beta = rsnew / rsold;                               // cg.cc:124
#pragma unroll
        for (int s = 0; s < S; ++s) {
            p[s].x = fma(beta, p[s].x, r[s].x);                          // cg.cc:127-129
            p[s].y = fma(beta, p[s].y, r[s].y);
        }
        if (owner) {
            const d2 rx = own[own_n];
            d2 px = own[2 * own_n];
            px.x = fma(beta, px.x, rx.x);
            px.y = fma(beta, px.y, rx.y);
            own[2 * own_n] = px;
        }
        rs_prev = rsold;
        rsold = rsnew;                                                   // cg.cc:132
    }

    // ---- state back to memory: x by the workgroup that owns the rows, r / p / scalars by workgroup 0
    // (into the OUTPUT set: the state the launch started from stays intact, cgx_kernels.h)
    // (the column indices are formed again from an opaque copy of tid: otherwise the compiler keeps the set-up's 64-bit indices alive
    // across the whole loop for these few stores -- in scratch, where the registers are full)
    int te = tid;
    asm volatile("" : "+v"(te));
    if (owner) {
        const d2 *oe = own;
        asm volatile("" : "+v"(oe));
        const int c = 2 * ((int)(oe - lds_own) + (row0 >> 1));      // (= 2 T sx + 2 tid, from the one value that is alive anyway)
        const d2 xo = own[0];
        if (ox0) a.out[c] = xo.x;
        if (ox1) a.out[c + 1] = xo.y;
    }
    if (blockIdx.x == 0) {
        double *r_out = a.out + state_off_r(a.lda), *p_out = a.out + state_off_p(a.lda);
        Scalars *sc_out = reinterpret_cast<Scalars *>(a.out + state_off_sc(a.lda));
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int c = 2 * T * s + 2 * te;
            if (c < n) { r_out[c] = r[s].x; p_out[c] = p[s].x; }
            if (c + 1 < n) { r_out[c + 1] = r[s].y; p_out[c + 1] = p[s].y; }
        }
        if (tid == 0) {
            sc_out->rs[k & 1] = rsold;
            sc_out->rs[(k + 1) & 1] = rs_prev;
            sc_out->k_final = stop ? k : 0;
            sc_out->done = stop;
        }
    }
    if (tid == 0) tail_report(a, 0, stop, k, lds_rec);
    // the ring's last loads (the rows of an iteration that never comes) are simply dropped with the wave
}

template <int S, int RB, int CH, int RL, int RG>
hipError_t with_stream_kernel(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    auto kern = k_cg_stream<S, RB, CH, RL, RG>;
    if (!a) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, kStrThreads, pl.lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(kStrThreads), pl.lds_bytes, s, *a);
    return hipGetLastError();
}

// ring depth: RB rows of S column steps.  From S = 5 ONE row (5-16 slots of 16 bytes per thread, 40-128 KB per CU in flight): measured,
// a second row in flight gains nothing, and its 4 S registers hold a row of A instead (N = 8192: 73.7 -> 71.6 us per iteration,
// 5120: 20.3 -> 18.8, 9216: 96.5 -> 93.7)
constexpr int stream_rb(int S) { return S <= 2 ? 8 : S <= 4 ? 4 : 1; }
constexpr int stream_ch(int S) { return S <= 4 ? S : S == 8 || S == 10 ? 1 : 2; }
// rows on the chip (n > 4096 only: below, the resident kernel runs): in the LDS what fits beside the parked Ap (8 KB S each of
// 160 KB), in registers what the compiler places without a byte of scratch (tests/test_kernel_resources.py)
// streamed rows read with the default cache policy instead of nt: a workgroup always runs on the same XCD, and what 32 workgroups
// read of two 40-KB rows each (2.6 MB) is still in that XCD's 4-MB L2 an iteration later.  Measured (CGX_STREAM_L2_ROWS = 0 / 1 /
// 2 / 3 / 4): N = 4608: 15.4 / 14.3 / 14.0 / 14.6 / 15.7 us per iteration, 5120: 18.7 / 17.8 / 17.8 / 18.8 / 19.2, 6144: 32.3 / 31.6 /
// 32.8 / 33.2 / 33.0, 7168: 52.9 / 50.5 / 51.3 / 51.4 / 50.8, from 8192 no difference (a row of 64 KB x 32 is half the L2)
constexpr int stream_l2_rows(int S) { return S == 5 ? 2 : S == 6 || S == 7 ? 1 : 0; }
constexpr int stream_rl(int S) { return S < 5 || S > 9 ? 0 : S <= 6 ? 2 : 1; }
constexpr int stream_rg(int S) { return S == 5 ? 7 : S == 6 ? 5 : S == 7 ? 4 : S == 8 ? 3 : S == 9 ? 2 : S == 10 || S == 11 ? 1 : 0; }

template <int S>
hipError_t stream_dispatch_s(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    return with_stream_kernel<S, stream_rb(S), stream_ch(S), stream_rl(S), stream_rg(S)>(pl, a, s, per_cu);
}

}  // namespace

hipError_t stream_dispatch(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    switch (pl.S) {
    case 1: return stream_dispatch_s<1>(pl, a, s, per_cu);
    case 2: return stream_dispatch_s<2>(pl, a, s, per_cu);
    case 3: return stream_dispatch_s<3>(pl, a, s, per_cu);
    case 4: return stream_dispatch_s<4>(pl, a, s, per_cu);
    case 5: return stream_dispatch_s<5>(pl, a, s, per_cu);
    case 6: return stream_dispatch_s<6>(pl, a, s, per_cu);
    case 7: return stream_dispatch_s<7>(pl, a, s, per_cu);
    case 8: return stream_dispatch_s<8>(pl, a, s, per_cu);
    case 9: return stream_dispatch_s<9>(pl, a, s, per_cu);
    case 10: return stream_dispatch_s<10>(pl, a, s, per_cu);
    case 11: return stream_dispatch_s<11>(pl, a, s, per_cu);
    case 12: return stream_dispatch_s<12>(pl, a, s, per_cu);
    case 13: return stream_dispatch_s<13>(pl, a, s, per_cu);
    case 14: return stream_dispatch_s<14>(pl, a, s, per_cu);
    case 15: return stream_dispatch_s<15>(pl, a, s, per_cu);
    case 16: return stream_dispatch_s<16>(pl, a, s, per_cu);
    }
    return hipErrorInvalidValue;
}

bool plan_stream(int n, int cus, size_t lds_per_wg, ResidentPlan *out)
{
    ResidentPlan pl{};
    if (n < 1024 || n > 1024 * 16 || cus < 1) return false;   // (the watched word is column 2 tid < 1024: n >= 1024)
    const int G = cus < 256 ? cus : 256;
    pl.stream = 1;
    pl.S = (n + 1023) / 1024;
    pl.RB = stream_rb(pl.S);
    pl.xslots = 1024 * pl.S;
    pl.RL = stream_rl(pl.S);
    pl.RG = stream_rg(pl.S);
    pl.l2_rows = stream_l2_rows(pl.S);
    const int res = pl.RL + pl.RG, need = (n + G - 1) / G;
    if (need <= res) return false;                            // (never with n > 4096: 17 rows per workgroup and more)
    pl.R = res + (need - res + pl.RB - 1) / pl.RB * pl.RB;    // rows per workgroup: those on the chip + a whole number of batches
    pl.rows_per_wg = pl.R;
    pl.grid = (n + pl.R - 1) / pl.R;
    pl.lds_bytes = ((size_t)kStrWaves * pl.R + 4 * kStrWaves + kStrWaves + 4 + (size_t)2 * pl.S * kStrThreads * (1 + pl.RL) + (size_t)6 * (pl.R / 2 + 2)) * sizeof(double);
    if (pl.R >= kStrThreads || pl.lds_bytes > lds_per_wg) return false;
    *out = pl;
    return true;
}

}  // namespace cgx
