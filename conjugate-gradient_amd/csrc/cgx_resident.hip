// cgx_resident.hip -- the whole CG loop of code/MPI/cg.cc:95-137 as ONE persistent kernel for matrices that fit on the chip.
//
// The reference's own experiment sizes start at N = 1024, 1448, 2048 (code/MPI/cg.run:15-44, results/strong_scaling.txt,
// results/weak_scaling.txt).  There the per-launch path (K1 + K3 per iteration) is bound by two kernel boundaries and two
// launch ramps per iteration (8-12 us), not by memory.  An MI355X has 256 CUs x 160 KB of LDS = 40 MB: an n x n fp64 matrix
// with n <= 2048 (32 MB) fits, one row group per CU, and never has to be read again after the first iteration; up to n = 4096
// (128 MiB) the CUs' 512 KB register files hold most of the rest.
//
//   grid  = G workgroups of 256 threads, G = ceil(n / R) <= 256, ALL resident (one per CU: the LDS footprint allows no second);
//   A     = the workgroup's R rows (n <= 2048: R = the power of two with R x 256 >= n, 4 at n = 1024, 8 at 1448 and 2048, all of
//           them in LDS, row-major at a pitch of S x 512 doubles; 2048 < n <= 4096: R = 16, of which RL in LDS, RG in registers
//           and the remaining RS streamed from memory every iteration -- the table in front of the kernel);
//   state = r, p REPLICATED in every workgroup, held in registers: thread t owns the column pairs {512 s + 2 t, + 1}, s < S --
//           the same columns whose A entries it reads from LDS / holds in registers / streams, so the GEMV needs no vector
//           traffic at all; x only for the workgroup's own rows (nobody else needs it);
//   one iteration (cg.cc:96-137) =
//     Ap_sub = A_sub p            2 R S fma per thread, row sums by DPP + one LDS combine                         cg.cc:100-102
//     publish Ap_sub              <= 16 doubles per workgroup as tagged words ({32 bits of the value | 32-bit tag of the epoch}
//                                 twice, cgx_device.h), ONE 16-byte agent-scope write-through store per double
//     gather Ap                   every thread polls the tagged words of its own 2 S columns (sc1 loads): every word
//                                 validates itself, so there is no flag, no fence and no grid barrier
//     p.Ap, alpha                 every workgroup over the whole vectors, same order: bit-identical everywhere    cg.cc:105-107
//     x += alpha p (own rows), r -= alpha Ap, r.r, break test, beta, p = r + beta p                               cg.cc:110-132
//   so what travels is Ap -- exactly the design of the multi-GPU exchange (cgx_kernels.hip), with CUs in place of GPUs.
//
// Measured on one MI355X (tools/resident_check.py, profiles/r04_resident/): 3.3-4.1 us per iteration for n <= 2048, 4.5-8.7 us
// up to n = 4096, against 7-26 us of the per-launch path.  Where the time goes (CGX_RESIDENT_PROFILE=1, workgroup 0): ~1.0-1.5 us
// until the watched word of another workgroup has arrived (store -> memory -> load, plus the skew between workgroups),
// 0.3-1.7 us for the gather round (at n = 2048 every workgroup asks for 32 KB of tagged words; by FETCH_SIZE about one copy
// per XCD and iteration leaves the L2s, profiles/r04_resident/pmc/), 0.4-3.6 us for the GEMV and its row sums (the top end:
// n = 4096 with 32 MiB streamed), 0.7-1.9 us for the two block reductions and the scalar divisions.
//
// Two parities of the exchange buffer suffice: a workgroup publishes epoch e+2 only after it has read every workgroup's e+1,
// which those publish only after they have read all of e.  Every wait is bounded by the wall clock (cgx_config.p2p_timeout_ms,
// default 5 s) and raises the context's device error word; nothing can spin for ever.
// The arithmetic per element is the reference's; only the summation order of the dot products is this kernel's own (fixed,
// so a solve is bitwise reproducible), like every other K1 shape.  No MFMA (0.25 flop/byte), no floating-point atomics.
#include "cgx_kernels.h"
#include "cgx_device.h"
#include "cgx_tagged.h"

namespace cgx {

namespace {

#ifndef CGX_RES_STREAM_POLICY
// The streamed rows' cache policy: the DEFAULT one, not nt.  What is streamed is at most 4 rows x 256 workgroups x 32 KB = 32 MiB
// per iteration, 4 MiB per XCD -- the size of an XCD's L2 -- and the same bytes every iteration: with the default policy part
// of them is still in the L2 an iteration later (7.51 -> 5.99 us per iteration at n = 4096, 5.03 -> 4.84 at 3584; sc0 / sc1 the
// same; profiles/r05_mall/, tools/exp_mall_policy.sh builds the others).  cgx_stream.hip's rows pass once per iteration through
// caches they do not fit: nt there (default policy: 92 instead of 79 us at n = 8192).
#define CGX_RES_STREAM_POLICY ""
#endif
constexpr int kResThreads = 256;


// ------------------------------------------------------------------------------------------------------------------------
// Where a workgroup's rows live.  n <= 2048: all R <= 8 rows in LDS.  2048 < n <= 4096: the matrix (up to 128 MiB) no longer
// fits the LDS alone -- but a CU also has a 512 KB register file, and a workgroup of 256 threads at one per CU may use all of
// it: R = 16 rows per workgroup, RL rows in LDS, RG rows in REGISTERS (thread t holds its own 2 S columns of each: 4 S registers
// per row, loaded once per launch), and the remaining RS rows streamed from memory every iteration (16-byte loads, default cache policy,
// two rows per batch, the first batch of an iteration issued right behind the gather of the previous one).
// S = 5: 7 + 9 + 0 (all resident, n <= 2560); S = 6: 6 + 10 + 0 (all resident, n <= 3072); S = 7: 5 + 10 + 1;
// S = 8: 4 + 8 + 4 (n = 4096: 32 of 128 MiB re-read per iteration).
// ------------------------------------------------------------------------------------------------------------------------
// the pause between a workgroup's publish and its first look at the others' words (units of 64 clocks), and whether that first look is
// at ONE watched word or at all of them (the table at the gather)
// (a pause that is too short costs a second round of the gather, 0.25-0.35 us; one that is too long only its own length: the
// values sit one step of 2 behind the edge measured in profiles/r05_nowatch/pause_scan.txt -- 14 | 16 at n <= 1024, 18 | 20 up to
// 1536, 22 | 24 up to 2048; above, the optimum is flat: 26 up to n = 2560 (3.88-3.91 us at 24-26 against 4.03 with the watched word), 28 up to
// 3072 (4.29-4.32 at 26-28 against 4.40); from n = 3073 the watched word with a pause of 16 is what measures best: pause_scan2.txt)
constexpr int res_pause(int S) { return S <= 2 ? 18 : S == 3 ? 22 : S <= 5 ? 26 : S == 6 ? 28 : 16; }
constexpr bool res_watch(int S) { return S >= 7; }
constexpr int kHybR = 16;
constexpr int hyb_rl(int S) { return (150 * 1024) / (S * 512 * 8) < kHybR ? (150 * 1024) / (S * 512 * 8) : kHybR; }
// rows in registers: what the 512 registers of a thread hold beside r, p, the row sums, a batch of streamed rows and the
// gather's words without a byte of scratch (hipcc 7.2: 364 / 446 / 487 / 511 registers at S = 5 / 6 / 7 / 8)
constexpr int hyb_rg(int S) { return S == 5 ? 9 : S == 6 ? 10 : S == 7 ? 10 : 8; }
// streamed rows in flight at a time: two at S = 8 (4 streamed rows), one at S = 7 (1 streamed row: the second one's registers hold a row)
constexpr int hyb_sb(int S) { return S == 8 ? 2 : 1; }

// R = rows per workgroup, a power of two (the row sums are reduced together, wave_sum_rows; the last workgroup may own fewer);
// S = column steps of 512; RL of the rows in LDS, RG in registers, the remaining R - RL - RG streamed every iteration.
// n <= 2048: R = 1 ... 8, RL = R (k_cg_resident<R, S, R, 0>: everything in LDS); above: R = 16 with the table above.
template <int R, int S, int RL, int RG>
__global__ __launch_bounds__(kResThreads, 1) void k_cg_resident(ResidentArgs a)
{
    static_assert(RL >= 1 && RG >= 0 && RL + RG <= R, "rows in LDS + rows in registers <= rows per workgroup");
    constexpr int RS = R - RL - RG, SB = hyb_sb(S);
    extern __shared__ double lds_all[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = a.n;
    constexpr int pitch = S * 512;
    double *lds_A = lds_all;                          // RL x pitch
    constexpr int kScratch = 4 * R + 8;
    double *lds_red = lds_all + (size_t)RL * pitch;   // two sets of per-iteration scratch
    double *lds_sum = lds_red + 2 * kScratch;         // 4 doubles for block_sum (set-up only)
    // one word: a wait of this workgroup expired.  Plain LDS accesses, ordered by the barriers (a `volatile` access through a cast
    // pointer is a FLAT instruction: its wait is vmcnt(0) and lgkmcnt(0) together)
    int *lds_fail = reinterpret_cast<int *>(lds_sum + 4);
    const int row0 = blockIdx.x * R;
    const int my_rows = min(R, n - row0);             // >= 1 by construction of the grid

    if (tid == 0) *lds_fail = 0;
    if (__syncthreads_or(__hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
        if (tid == 0) tail_report(a, 1, 0, a.k0, nullptr);
        return;
    }

    // a row of the block as this thread sees it: its 2 S columns; rows behind the last one are read as row n-1 (never published),
    // columns behind the pitch as zero (columns n .. lda are zero in the block already)
    auto row_ptr = [&](int i) { return a.A + (size_t)min(row0 + i, n - 1) * a.lda; };
    auto a_load = [&](const double *row, int s) {
        const int c = 512 * s + 2 * tid;
        d2 v = {0.0, 0.0};
        if (c < a.lda) v = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(row + c));
        return v;
    };

    // The streamed rows are loaded with ONE 32-bit byte offset per column step (shared by all rows) on a workgroup-uniform row
    // base in SGPRs: left to the compiler, the RS x S 64-bit addresses are hoisted out of the iteration loop and held in
    // VGPRs (seen as 900 bytes of scratch per lane).  A column step that reaches behind the pitch reads the last pair of
    // columns inside the pitch instead: with the default pitch those are pad columns (zero by construction); with lda_pad = 0
    // they may be real entries -- either way the product is with p = 0 (the thread's own column is >= n), i.e. exactly 0.
    unsigned soff[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const long c = 512 * s + 2 * tid;
        soff[s] = (unsigned)(8 * (c < a.lda - 2 ? c : a.lda - 2));
    }
    auto stream_issue = [&](int i, int s) {
        const double *row = row_ptr(i);
        d2 v;
        asm volatile("global_load_dwordx4 %0, %1, %2" CGX_RES_STREAM_POLICY : "=v"(v) : "v"(soff[s]), "s"(row) : "memory");
        return v;
    };

    // ---- rows 0 .. RL-1 -> LDS, rows RL .. RL+RG-1 -> registers, once per launch
    for (int i = 0; i < RL; ++i) {
        const double *row = row_ptr(i);
#pragma unroll
        for (int s = 0; s < S; ++s) *reinterpret_cast<d2 *>(lds_A + (size_t)i * pitch + 512 * s + 2 * tid) = a_load(row, s);
    }
    d2 areg[RG > 0 ? RG : 1][S];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const double *row = row_ptr(RL + i);
#pragma unroll
        for (int s = 0; s < S; ++s) areg[i][s] = a_load(row, s);
    }

    // ---- state: r, p for this thread's columns (replicated in every workgroup); x for the workgroup's own rows only
    const double *st_x = a.in, *st_r = a.in + state_off_r(a.lda), *st_p = a.in + state_off_p(a.lda);
    const Scalars *st_sc = reinterpret_cast<const Scalars *>(a.in + state_off_sc(a.lda));
    d2 r[S], p[S];
    unsigned okmask = 0;                                              // bit 2 s / 2 s + 1: column 512 s + 2 tid / + 1 is below n
    d2 xo = {0.0, 0.0};
    int sx = -1;
    bool ox0 = false, ox1 = false;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int c = 512 * s + 2 * tid;
        const bool ok0 = c < n, ok1 = c + 1 < n;
        okmask |= (ok0 ? 1u : 0u) << (2 * s) | (ok1 ? 1u : 0u) << (2 * s + 1);
        r[s].x = ok0 ? st_r[c] : 0.0;
        r[s].y = ok1 ? st_r[c + 1] : 0.0;
        if (a.k0 > 0) {
            p[s].x = ok0 ? st_p[c] : 0.0;
            p[s].y = ok1 ? st_p[c + 1] : 0.0;
        } else {
            p[s] = r[s];                                              // p = r, cg.cc:85
        }
        // x: only the workgroup's own rows [row0, row0 + my_rows), at most 16 columns, i.e. at most one s for a thread
        const bool own0 = c >= row0 && c < row0 + my_rows, own1 = c + 1 >= row0 && c + 1 < row0 + my_rows;
        if (own0 || own1) {
            sx = s;
            ox0 = own0;
            ox1 = own1;
            if (own0) xo.x = st_x[c];
            if (own1) xo.y = st_x[c + 1];
        }
    }
    double rsold, rs_prev;
    if (a.k0 > 0) {
        rsold = st_sc->rs[a.k0 & 1];
        rs_prev = st_sc->rs[(a.k0 + 1) & 1];
    } else {
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) v += r[s].x * p[s].x + r[s].y * p[s].y;   // rsold = r.p, cg.cc:91-92
        rsold = block_sum<4>(v, lds_sum);
        rs_prev = rsold;
    }
    rsold = uniform(rsold);     // (the same bits in every lane: kept in scalar registers across the iteration)
    rs_prev = uniform(rs_prev);
    __syncthreads();   // the LDS rows are in place

    // The first batch of streamed rows of an iteration is issued as early as its buffer is free: the rows do not depend on p,
    // so the loads of iteration k+1 go out right behind the gather of iteration k and are in flight during the two block
    // reductions, the divisions and the p update.
    d2 sb[RS > 0 ? SB : 1][S];
    auto stream_first = [&]() {
        if constexpr (RS > 0) {
#pragma unroll
            for (int j = 0; j < SB; ++j)
#pragma unroll
                for (int s = 0; s < S; ++s) sb[j][s] = stream_issue(RL + RG + (j < RS ? j : RS - 1), s);
        }
    };
    stream_first();

    int k = a.k0, stop = 0;
    const int k_end = a.k0 + a.iters;
    unsigned long long epoch = a.epoch0;
    // what this workgroup's waits cost (a.rec): thread 0 keeps it in four LDS words (the kernel has no register to spare)
    unsigned *lds_rec = reinterpret_cast<unsigned *>(lds_fail) + 2;   // [watch rounds | repeated gather rounds | first wait | longest later wait]
    if (tid < 4) lds_rec[tid] = 0;
    for (; k < k_end; ++k) {
        ++epoch;
        const unsigned tag = p2p_tag(epoch);
        unsigned long long *slot = a.xbuf + (size_t)(epoch & 1) * (2 * a.xslots);
        double *red = lds_red + (k & 1) * kScratch;     // [4 waves][R] row sums | [4] p.Ap | [4] r.r
        const bool prof = a.prof != nullptr && blockIdx.x == 0 && tid == 0;
        long long tp[6] = {0, 0, 0, 0, 0, 0};
        int watch_rounds = 0, gather_rounds = 0;
        if (prof) tp[0] = clock64();

        // Ap_sub = A_sub p (cblas_dgemv, cg.cc:100-102): the thread's columns of every row, ascending
        double acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = 0.0;
        // the LDS rows, one column step at a time: the RL reads of a step are issued together, and the reads of the next step go
        // out before the FMAs of this one (sched_barrier: left to itself the compiler reuses one destination register quad and
        // the loop becomes read / wait / fma triples -- 35 exposed LDS latencies at S = 5, seen in the ISA and in the profile)
        {
            d2 av[2][RL];
#pragma unroll
            for (int i = 0; i < RL; ++i) av[0][i] = *reinterpret_cast<const d2 *>(lds_A + (size_t)i * pitch + 2 * tid);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (s + 1 < S) {
#pragma unroll
                    for (int i = 0; i < RL; ++i)
                        av[(s + 1) & 1][i] = *reinterpret_cast<const d2 *>(lds_A + (size_t)i * pitch + 512 * (s + 1) + 2 * tid);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RL; ++i) {
                    acc[i] = fma(av[s & 1][i].x, p[s].x, acc[i]);
                    acc[i] = fma(av[s & 1][i].y, p[s].y, acc[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s)                                   // the register rows
#pragma unroll
            for (int i = 0; i < RG; ++i) {
                acc[RL + i] = fma(areg[i][s].x, p[s].x, acc[RL + i]);
                acc[RL + i] = fma(areg[i][s].y, p[s].y, acc[RL + i]);
            }
        if constexpr (RS > 0) {
#pragma unroll
            for (int b = 0; b < RS; b += SB) {                        // the streamed rows, SB at a time
                stream_wait<SB * S>(&sb[0][0]);
#pragma unroll
                for (int j = 0; j < SB; ++j)
                    if (b + j < RS) {
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            acc[RL + RG + b + j] = fma(sb[j][s].x, p[s].x, acc[RL + RG + b + j]);
                            acc[RL + RG + b + j] = fma(sb[j][s].y, p[s].y, acc[RL + RG + b + j]);
                        }
                    }
                if (b + SB < RS) {
#pragma unroll
                    for (int j = 0; j < SB; ++j)
#pragma unroll
                        for (int s = 0; s < S; ++s) sb[j][s] = stream_issue(RL + RG + (b + SB + j < RS ? b + SB + j : RS - 1), s);
                }
            }
        }
        const int myrow = wave_sum_rows_swap<R>(acc, lane);               // acc[0] = this wave's part of row `myrow`
        if ((lane & (64 / R - 1)) == 0) red[wave * R + myrow] = acc[0];
        __syncthreads();
        if (tid < my_rows && !(k == a.k0 && (int)blockIdx.x == a.mute_wg)) {   // (mute_wg: the test of the bounded waits)
            const double ap = (red[tid] + red[R + tid]) + (red[2 * R + tid] + red[3 * R + tid]);
            tagged_put(slot + 2 * (size_t)xpos(row0 + tid), ap, tag);
        }

        // gather Ap: one watched word first, then the tagged words of all of the thread's columns in one round
        d2 ap[S];
        {
            unsigned need = okmask;
#pragma unroll
            for (int s = 0; s < S; ++s) ap[s].x = ap[s].y = 0.0;
            bool any = need != 0;
            const long long t0 = wall_clock64();
            bool expired = false;
            if (prof) tp[1] = clock64();
            // The first poll is NOT sent at once: a poll that reaches memory while the other workgroups' write-through stores of the
            // same lines are still on their way comes back late (and holds up, in order, whatever the wave asks for next); one
            // that leaves about 0.4 us later finds the words there.  Measured, the whole loop of `cgsolver n out` in us, delay in
            // units of s_sleep 1 (64 clocks): n = 512 / 1024 / 2048 / 4096: no delay 509 / 685 / 992 / 3134, 8: 382 / 530 / 926 / 3085,
            // 14: 333 / 497 / 923 / 3068, 16: 330 / 500 / 898 / 3063, 20: 353 / 522 / 910 / 3041, 28: 373 / 557 / 966 / 3070
            // (profiles/r05_window/poll_delay.txt): 2.8 us per iteration at n = 1024 instead of 3.3-3.9.
            // Up to n = 3072 there is no watched word either: behind a pause of the right length the gather's first round finds all
            // words there, and a poll in front of it is a round trip for nothing (what has not arrived is asked for again, as ever).
            // Measured, us per iteration at n = 256 / 512 / 1024 / 1448 / 2048 / 2896 / 4096 (profiles/r05_nowatch/):
            //   watched word, pause 16 (round 5 so far)   2.28 / 2.40 / 2.59 / 3.04 / 3.33 / 4.43 / 6.04
            //   no watched word, pause 16                 1.94 / 2.05 / 2.26 / 2.98 / 3.46 / 4.65 / 6.43
            //   no watched word, pause 20                 2.05 / 2.16 / 2.37 / 2.80 / 3.31 / 4.59 / 6.38
            //   no watched word, pause 24                 2.15 / 2.26 / 2.47 / 2.90 / 3.12 / 4.37 / 6.42
            //   built: pause 18 / 22 / 26 (res_pause)     2.00 / 2.07 / 2.26 / 2.81 / 3.11
            // (above n = 3072 -- rows streamed every iteration -- the workgroups' publishes lie further apart and the first round comes too
            // early more often)
            __builtin_amdgcn_s_sleep(res_pause(S));
            if (any && res_watch(S)) {
                const unsigned long long *watch = slot + 2 * (size_t)xpos(2 * tid);   // column 2 tid: valid whenever `any`
                for (;;) {
                    u4 w = tagged_issue(watch);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(w)::"memory");
                    ++watch_rounds;
                    if (w.y == tag && w.w == tag) break;
                    if (wall_clock64() - t0 > a.timeout_ticks) { expired = true; break; }
                }
            }
            if (prof) tp[2] = clock64();
            while (any && !expired) {
                ++gather_rounds;
                u4 w[2 * S];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const int c = ((need >> (2 * s)) & 3u) ? 512 * s + 2 * tid : 2 * tid;
                    w[2 * s] = tagged_issue(slot + 2 * (size_t)xpos(c));
                    w[2 * s + 1] = tagged_issue(slot + 2 * (size_t)xpos(c + 1));
                }
                tagged_wait<S>(w);
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    if (((need >> (2 * s)) & 1u) && w[2 * s].y == tag && w[2 * s].w == tag) {
                        ap[s].x = tagged_value(w[2 * s]);
                        need &= ~(1u << (2 * s));
                    }
                    if (((need >> (2 * s + 1)) & 1u) && w[2 * s + 1].y == tag && w[2 * s + 1].w == tag) {
                        ap[s].y = tagged_value(w[2 * s + 1]);
                        need &= ~(1u << (2 * s + 1));
                    }
                }
                any = need != 0;
                if (any && wall_clock64() - t0 > a.timeout_ticks) expired = true;
            }
            // What the exchange of this iteration cost (ResidentTail): the repeated polls are counted (two LDS adds that nobody waits
            // for); the span from the publish to the last gathered word is TIMED only when it took several round trips, or in the
            // launch's first iteration, where a workgroup that was placed late shows -- at n = 1024 the first poll of the watched
            // word usually comes too early (the exchange takes about two round trips), and a clock read in every iteration was
            // 0.6 us of 3.2.
            if (tid == 0) {
                // (thread 0 is the only one that touches these words: no atomics)
                if (watch_rounds > 1) lds_rec[0] += (unsigned)(watch_rounds - 1);
                if (gather_rounds > 1) lds_rec[1] += (unsigned)(gather_rounds - 1);
                if (watch_rounds + gather_rounds > 6 || k == a.k0) {
                    const unsigned dt = (unsigned)(wall_clock64() - t0);
                    unsigned *w = lds_rec + (k == a.k0 ? 2 : 3);
                    if (dt > *w) *w = dt;
                }
            }
            if (expired) {
                atomicExch(a.err, 1);
                *lds_fail = 1;
            }
        }

        stream_first();                                               // for iteration k + 1 (wasted behind the last one)
        if (prof) tp[3] = clock64();
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) v += p[s].x * ap[s].x + p[s].y * ap[s].y;   // cg.cc:105-106
        v = wave_sum_swap(v);
        if (lane == 0) red[4 * R + wave] = v;
        __syncthreads();
        if (*lds_fail) {                                             // uniform: written in front of the barrier
            if (tid == 0) tail_report(a, 1, 0, k, lds_rec);
            return;
        }
        const double conj = (red[4 * R] + red[4 * R + 1]) + (red[4 * R + 2] + red[4 * R + 3]);
        const double alpha = safeguarded_alpha(rsold, conj);         // cg.cc:107
        if (prof) tp[4] = clock64();
        double rr = 0.0;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (s == sx) {                                           // cg.cc:110, the workgroup's own rows
                xo.x = fma(alpha, p[s].x, xo.x);
                xo.y = fma(alpha, p[s].y, xo.y);
            }
            r[s].x = fma(-alpha, ap[s].x, r[s].x);                   // cg.cc:113
            r[s].y = fma(-alpha, ap[s].y, r[s].y);
            rr += r[s].x * r[s].x + r[s].y * r[s].y;                 // cg.cc:116
        }
        rr = wave_sum_swap(rr);
        if (lane == 0) red[4 * R + 4 + wave] = rr;
        __syncthreads();
        const double rsnew = uniform((red[4 * R + 4] + red[4 * R + 5]) + (red[4 * R + 6] + red[4 * R + 7]));   // cg.cc:116-117
        if (prof) {
            tp[5] = clock64();
            for (int i = 0; i < 5; ++i) a.prof[i] += tp[i + 1] - tp[i];
            a.prof[5] += watch_rounds;
            a.prof[6] += gather_rounds;
            a.prof[7] += 1;
        }
        if (sqrt(rsnew) < a.tol) {                                   // cg.cc:120-121: break before the p update
            rs_prev = rsnew;
            stop = 1;
            break;
        }
        const double beta = rsnew / rsold;                           // cg.cc:124
#pragma unroll
        for (int s = 0; s < S; ++s) {
            p[s].x = fma(beta, p[s].x, r[s].x);                      // cg.cc:127-129
            p[s].y = fma(beta, p[s].y, r[s].y);
        }
        rs_prev = rsold;
        rsold = rsnew;                                               // cg.cc:132
    }

    // ---- state back to memory: x by the workgroup that owns the rows, r / p / scalars by workgroup 0
    // (into the OUTPUT set: the state the launch started from stays intact, so that a launch whose waits expired can be redone
    // on the per-launch path, resident_steps in cgx_solve.cpp)
    // (the column indices are formed again from an opaque copy of tid: otherwise the compiler keeps the set-up's 64-bit indices alive
    // across the whole loop for these few stores)
    int te = tid;
    asm volatile("" : "+v"(te));
    if (sx >= 0) {
        const int c = 512 * sx + 2 * te;
        if (ox0) a.out[c] = xo.x;
        if (ox1) a.out[c + 1] = xo.y;
    }
    if (blockIdx.x == 0) {
        double *r_out = a.out + state_off_r(a.lda), *p_out = a.out + state_off_p(a.lda);
        Scalars *sc_out = reinterpret_cast<Scalars *>(a.out + state_off_sc(a.lda));
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int c = 512 * s + 2 * te;
            if ((okmask >> (2 * s)) & 1u) { r_out[c] = r[s].x; p_out[c] = p[s].x; }
            if ((okmask >> (2 * s + 1)) & 1u) { r_out[c + 1] = r[s].y; p_out[c + 1] = p[s].y; }
        }
        if (tid == 0) {
            sc_out->rs[k & 1] = rsold;
            sc_out->rs[(k + 1) & 1] = rs_prev;
            sc_out->k_final = stop ? k : 0;
            sc_out->done = stop;
        }
    }
    if (tid == 0) tail_report(a, 0, stop, k, lds_rec);
}

// a == nullptr: prepare (raise the kernel's dynamic-LDS limit, ask the runtime how many workgroups a CU keeps resident); else launch
template <int R, int S, int RL, int RG>
hipError_t with_kernel(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    auto kern = k_cg_resident<R, S, RL, RG>;
    if (!a) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes);
        if (e != hipSuccess) return e;
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, kern, kResThreads, pl.lds_bytes);
    }
    hipLaunchKernelGGL(kern, dim3(pl.grid), dim3(kResThreads), pl.lds_bytes, s, *a);
    return hipGetLastError();
}

template <int R>
hipError_t all_in_lds(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    switch (pl.S) {
    case 1: return with_kernel<R, 1, R, 0>(pl, a, s, per_cu);
    case 2: return with_kernel<R, 2, R, 0>(pl, a, s, per_cu);
    case 3: return with_kernel<R, 3, R, 0>(pl, a, s, per_cu);
    case 4: return with_kernel<R, 4, R, 0>(pl, a, s, per_cu);
    }
    return hipErrorInvalidValue;
}

hipError_t dispatch(const ResidentPlan &pl, const ResidentArgs *a, hipStream_t s, int *per_cu)
{
    if (pl.stream) return stream_dispatch(pl, a, s, per_cu);
    if (pl.hybrid) {
        switch (pl.S) {
        case 5: return with_kernel<kHybR, 5, hyb_rl(5), hyb_rg(5)>(pl, a, s, per_cu);
        case 6: return with_kernel<kHybR, 6, hyb_rl(6), hyb_rg(6)>(pl, a, s, per_cu);
        case 7: return with_kernel<kHybR, 7, hyb_rl(7), hyb_rg(7)>(pl, a, s, per_cu);
        case 8: return with_kernel<kHybR, 8, hyb_rl(8), hyb_rg(8)>(pl, a, s, per_cu);
        }
        return hipErrorInvalidValue;
    }
    switch (pl.R) {
    case 1: return all_in_lds<1>(pl, a, s, per_cu);
    case 2: return all_in_lds<2>(pl, a, s, per_cu);
    case 4: return all_in_lds<4>(pl, a, s, per_cu);
    case 8: return all_in_lds<8>(pl, a, s, per_cu);
    }
    return hipErrorInvalidValue;
}

}  // namespace

bool plan_resident(int n, int cus, size_t lds_per_wg, ResidentPlan *out)
{
    ResidentPlan pl{};
    if (n > 512 * 8) return plan_stream(n, cus, lds_per_wg, out);
    if (n < 1 || cus < 1) return false;
    const int G = cus < 256 ? cus : 256;
    pl.S = (n + 511) / 512;
    pl.xslots = 512 * pl.S;
    if (n > 512 * 4) {
        // 2048 < n <= 4096: 16 rows per workgroup, RL in LDS + RG in registers + RS streamed per iteration
        pl.hybrid = 1;
        pl.R = pl.rows_per_wg = kHybR;
        pl.RL = hyb_rl(pl.S);
        pl.RG = hyb_rg(pl.S);
        pl.grid = (n + kHybR - 1) / kHybR;
        if (pl.grid > G) return false;
        pl.lds_bytes = ((size_t)pl.RL * pl.S * 512 + 2 * (4 * kHybR + 8) + 8) * sizeof(double);
        if (pl.lds_bytes > lds_per_wg) return false;
        *out = pl;
        return true;
    }
    pl.R = 1;
    while ((long)pl.R * G < n) pl.R *= 2;
    if (pl.R > 8) return false;
    pl.rows_per_wg = pl.R;
    pl.RL = pl.R;
    pl.grid = (n + pl.rows_per_wg - 1) / pl.rows_per_wg;
    pl.lds_bytes = ((size_t)pl.rows_per_wg * pl.S * 512 + 2 * (4 * pl.R + 8) + 8) * sizeof(double);
    if (pl.lds_bytes > lds_per_wg) return false;
    *out = pl;
    return true;
}

hipError_t prepare_cg_resident(const ResidentPlan &pl, int *workgroups_per_cu)
{
    *workgroups_per_cu = 0;
    return dispatch(pl, nullptr, nullptr, workgroups_per_cu);
}

hipError_t launch_cg_resident(const ResidentPlan &pl, const ResidentArgs &a, hipStream_t s)
{
    if (a.rows_per_wg != pl.rows_per_wg || a.xslots != pl.xslots || a.iters < 0) return hipErrorInvalidValue;
    return dispatch(pl, &a, s, nullptr);
}

}  // namespace cgx
