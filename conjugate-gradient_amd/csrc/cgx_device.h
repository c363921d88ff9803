// cgx_device.h -- device-side helpers shared by the kernel translation units (cgx_kernels.hip, cgx_resident.hip):
// the fixed-order reductions, the safeguard of alpha, and the tagged-word store.  Device code only.
#pragma once

#include <hip/hip_runtime.h>

namespace cgx {

typedef double d2 __attribute__((ext_vector_type(2)));

static constexpr double kNearZero = 1.0e-14;   // NEARZERO, code/MPI/cg.cc:8

// alpha = rsold / std::max(conj, rsold * NEARZERO), cg.cc:107.  std::max(a, b) is (a < b) ? b : a: a NaN p.Ap stays a
// NaN (the comparison is false), a NaN bound is ignored.  fmax would return the other operand in both cases.
__device__ __forceinline__ double safeguarded_alpha(double rsold, double conj)
{
    const double bound = rsold * kNearZero;
    return rsold / ((conj < bound) ? bound : conj);
}


// ------------------------------------------------------------------------------------------------
// reductions: fixed order => bitwise reproducible for a given launch shape
// ------------------------------------------------------------------------------------------------

// Sums of R independent per-lane values over the 64 lanes at once (R a power of two).  Instead of R butterflies of six
// exchanges each, the lanes first split the rows among themselves: at every halving step a lane keeps half of its rows
// and hands the other half to its partner, so the exchanges go R/2 + R/4 + ... + 1, and the rest of the butterfly runs
// on ONE value.  R = 8: 10 exchanges instead of 48.  Afterwards v[0] of lane L is the total of row L >> (6 - log2 R)
// (every lane of that group holds it).  Fixed order => bitwise reproducible for a given shape.
// v + (v of the partner lane), the partner given by a DPP control word: no LDS crossbar, a few cycles instead of a
// ds_bpermute round trip.  0xB1 / 0x4E: quad_perm = lane ^ 1 / lane ^ 2.  0x141 / 0x140: row_half_mirror / row_mirror
// pair lane i with 7 - i of its 8 / 15 - i of its 16 lanes -- as good as lane ^ 4 / lane ^ 8 for a sum once the lower
// levels have made the lanes of each quad / each 8 hold the same value (which is the order they are used in below).
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return v + __hiloint2double(hi, lo);
}

// Total of one value over groups of SPAN consecutive lanes (SPAN a power of two <= 64), every lane of the group gets it:
// levels 1, 2, 4, 8 by DPP, 16 and 32 by ds_bpermute.
template <int SPAN>
__device__ __forceinline__ double group_sum(double v)
{
    if constexpr (SPAN > 32) v += __shfl_xor(v, 32, 64);
    if constexpr (SPAN > 16) v += __shfl_xor(v, 16, 64);
    if constexpr (SPAN > 1) v = dpp_add<0xB1>(v);
    if constexpr (SPAN > 2) v = dpp_add<0x4E>(v);
    if constexpr (SPAN > 4) v = dpp_add<0x141>(v);
    if constexpr (SPAN > 8) v = dpp_add<0x140>(v);
    return v;
}

__device__ __forceinline__ double wave_sum(double v)
{
    return group_sum<64>(v);   // every lane holds the total
}

template <int R, int N, int WIDTH>
__device__ __forceinline__ void wave_sum_rows_step(double (&v)[R], int lane)
{
    if constexpr (N > 1) {
        const bool upper = (lane & WIDTH) != 0;
#pragma unroll
        for (int i = 0; i < N / 2; ++i) {
            const double send = upper ? v[i] : v[i + N / 2];
            const double keep = upper ? v[i + N / 2] : v[i];
            v[i] = keep + __shfl_xor(send, WIDTH, 64);
        }
        wave_sum_rows_step<R, N / 2, WIDTH / 2>(v, lane);
    } else {
        v[0] = group_sum<2 * WIDTH>(v[0]);   // the lanes that still differ: groups of 2*WIDTH = 64/R
    }
}

template <int R>
__device__ __forceinline__ int wave_sum_rows(double (&v)[R], int lane)
{
    static_assert(R >= 1 && R <= 64 && (R & (R - 1)) == 0, "rows per workgroup must be a power of two");
    wave_sum_rows_step<R, R, 32>(v, lane);
    return lane / (64 / R);   // the row this lane holds: its top log2(R) lane bits
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

// The tag of an epoch: 1 + (epoch mod (2^32 - 1)), i.e. 1 ... 2^32 - 1, never 0; consecutive epochs of one parity (e-2, e)
// always differ, and two epochs share a tag only 2^32 - 1 apart.
__device__ __forceinline__ unsigned p2p_tag(unsigned long long epoch)
{
    return (unsigned)(epoch % 0xFFFFFFFFull) + 1u;
}
// Both words of one double leave as ONE 16-byte store with the system-scope write-through bits (what the two relaxed 8-byte
// atomic stores of the definition compile to, `global_store_dwordx2 ... sc0 sc1`, as one instruction and one request: over
// xGMI a request is a packet, and 8-byte packets cost 2.7x the time per byte of 16-byte ones, measured for sc1 stores in the
// guide).  Nothing depends on the two words arriving together: each validates itself.
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void tagged_store(unsigned long long *dst, double v, unsigned tag)
{
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const u4 w = {(unsigned)bits, tag, (unsigned)(bits >> 32), tag};   // little endian: {lo32 | tag<<32}, {hi32 | tag<<32}
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(w) : "memory");
}

}  // namespace cgx
