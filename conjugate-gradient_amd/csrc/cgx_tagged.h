// cgx_tagged.h -- device helpers shared by the persistent kernels (cgx_resident.hip, cgx_stream.hip): the tagged-word
// exchange between workgroups (16-byte agent-scope loads and stores of {lo32, tag, hi32, tag}), the wave reductions on
// gfx950's v_permlane32_swap / v_permlane16_swap, and the layout of an exchange buffer.  Device code only.
#pragma once

#include "cgx_device.h"
#include "cgx_kernels.h"

namespace cgx {

// 16-byte agent-scope load (sc1: past the L1, which no other CU's store refreshes) of one tagged double
// {lo32, tag, hi32, tag}.  The caller waits with tagged_wait().
__device__ __forceinline__ u4 tagged_issue(const unsigned long long *src)
{
    u4 w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w) : "v"(src) : "memory");
    return w;
}

// One double as two tagged words, ONE 16-byte agent-scope write-through store (between GPUs the same words travel at system
// scope: tagged_store in cgx_device.h).
__device__ __forceinline__ void tagged_put(unsigned long long *dst, double v, unsigned tag)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const u4 w = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(w) : "memory");
}

template <int S>
__device__ __forceinline__ void tagged_wait(u4 (&w)[2 * S])
{
    // the loaded registers are operands of the wait, so that no use of them can be scheduled in front of it
    if constexpr (S == 1) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1])::"memory");
    if constexpr (S == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3])::"memory");
    if constexpr (S == 3)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5])::"memory");
    if constexpr (S == 4)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])::"memory");
    if constexpr (S > 4) {   // (an asm statement takes at most 30 operands)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])::"memory");
        if constexpr (S == 5) asm volatile("" : "+v"(w[8]), "+v"(w[9])::"memory");
        if constexpr (S == 6) asm volatile("" : "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11])::"memory");
        if constexpr (S == 7) asm volatile("" : "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13])::"memory");
        if constexpr (S == 8)
            asm volatile("" : "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15])::"memory");
    }
}

// a value every lane holds alike, moved into scalar registers (two of them instead of two vector registers per lane)
__device__ __forceinline__ double uniform(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// wait for the N 16-byte loads of a batch of streamed rows (the registers are operands, as in tagged_wait)
template <int N>
__device__ __forceinline__ void stream_wait(d2 *v)
{
    static_assert(N >= 1 && N <= 16, "a batch of streamed rows is at most 16 loads");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(v[i])::"memory");
}

// Sum over the 64 lanes, every lane gets it: the same pairing and order as wave_sum (lane ^ 32, lane ^ 16, then the four DPP
// levels), so the same bits -- but the two upper levels by gfx950's v_permlane32_swap / v_permlane16_swap (VALU, a few cycles)
// instead of ds_bpermute round trips through the LDS crossbar.  With both operands the same register x, permlane32_swap leaves
// [x.lower | x.lower] in one result and [x.upper | x.upper] in the other: their sum is x + x(lane ^ 32) in every lane.
__device__ __forceinline__ double wave_sum_swap(double v)
{
    {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
        v = __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
    return group_sum<16>(v);
}

// a + (b of the partner half) for the first two levels of wave_sum_rows (cgx_device.h), by the same instructions: with
// operands a = v[i], b = v[i + N/2], permlane32_swap leaves [a.lower | b.lower] and [a.upper | b.upper]; their sum is, in the
// lower 32 lanes, own v[i] + the partner's v[i], and in the upper 32, own v[i + N/2] + the partner's v[i + N/2]: exactly what
// the exchange "keep one half of the rows, hand the other half over" computes, without a select.  Same pairing, same bits.
template <bool ROW16>
__device__ __forceinline__ double swap_add(double a, double b)
{
    if constexpr (ROW16) {
        const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
        return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
        return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
    }
}

template <int R>
__device__ __forceinline__ int wave_sum_rows_swap(double (&v)[R], int lane)
{
    static_assert(R >= 1 && R <= 64 && (R & (R - 1)) == 0, "rows per workgroup must be a power of two");
    if constexpr (R == 1) {
        v[0] = wave_sum_swap(v[0]);
    } else {
#pragma unroll
        for (int i = 0; i < R / 2; ++i) v[i] = swap_add<false>(v[i], v[i + R / 2]);
        if constexpr (R == 2) {
            wave_sum_rows_step<R, 1, 16>(v, lane);
        } else {
#pragma unroll
            for (int i = 0; i < R / 4; ++i) v[i] = swap_add<true>(v[i], v[i + R / 4]);
            wave_sum_rows_step<R, R / 4, 8>(v, lane);
        }
    }
    return lane / (64 / R);
}

// Where the tagged double of column c sits in a parity of the exchange buffer: inside each block of 128 columns the even ones
// first, then the odd ones.  A thread owns the column pair (c, c + 1) (its LDS reads are 16-byte pairs of adjacent columns), so
// with this layout the 64 lanes of a wave fetch their even columns with ONE fully coalesced 1-KiB load and their odd columns
// with another, instead of two loads that each touch half of 16 lines.
__device__ __forceinline__ int xpos(int c)
{
    return (c & ~127) | ((c & 1) << 6) | ((c & 127) >> 1);
}

__device__ __forceinline__ double tagged_value(const u4 &w)
{
    return __longlong_as_double((long long)((unsigned long long)w.x | ((unsigned long long)w.z << 32)));
}

// A workgroup's report at the end of a persistent launch (ResidentTail, cgx_kernels.h), thread 0 only.  `rec` = the four LDS
// words it kept during the launch: polls of the watched word that had to be repeated, gather rounds that had to be repeated, ticks
// from its publish of the launch's first iteration until it had all of Ap, the longest such span of a later iteration in which
// a poll had to be repeated.  An iteration whose polls all succeed at once is not timed: one round trip is the price of the
// exchange, not a wait.
__device__ __forceinline__ void tail_report(const ResidentArgs &a, int err, int stop, int k, const unsigned *rec)
{
    ResidentTail *t = a.tail;
    t->waits[blockIdx.x][0] = rec ? rec[2] : 0u;
    t->waits[blockIdx.x][1] = rec ? rec[3] : 0u;
    if (blockIdx.x == 0) {
        t->done = stop;
        t->k_final = stop ? k : 0;
        t->err = err;
        t->iterations = k - a.k0 + stop;
        t->watch_repeats = rec ? rec[0] : 0u;
        t->gather_repeats = rec ? rec[1] : 0u;
        t->stamp = a.stamp;
    }
}

// LDS hand-off between the waves of a workgroup WITHOUT draining the wave's global loads: __syncthreads() is a workgroup-scope
// release + acquire, which on gfx9 waits for vmcnt(0) as well -- i.e. for every row of A a persistent kernel has prefetched
// for the next iteration.  What the kernels hand over at their barriers lives in LDS only, so the LDS counter is enough.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace cgx
