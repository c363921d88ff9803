// Dev tool: can part of a row block that is re-read every iteration be kept in the 256 MiB Infinity Cache?
// A block of `rows` x 32768 doubles (1 GiB at 4096 rows) is swept repeatedly, as a rank of an 8-GPU run sweeps its shard.
// The first `keep` rows are read with default-policy loads (they may stay resident), the rest with non-temporal loads (they
// should not displace them).  Two launches per sweep (one per policy: inside ONE kernel the compiler merges the two load
// blocks and drops the nt).  hipcc --offload-arch=gfx950 -O3 tools/hbm_mall_keep.hip -o /tmp/hbm_mall_keep && /tmp/hbm_mall_keep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int R, int U, bool NT>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ A, long pitch, int ncols, const double* __restrict__ v,
                                              double* out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = reinterpret_cast<const char*>(A + ((long)blockIdx.x * R + i) * pitch);
    double s0 = 0, s1 = 0;
    constexpr int kStep = 512;
    for (int c = w * 128 + lane * 2; c + (U - 1) * kStep < ncols; c += U * kStep) {
        d2 av[U][R], pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(v) + (unsigned)(c + u * kStep) * 8u);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const d2* p = reinterpret_cast<const d2*>(a[i] + (unsigned)(c + u * kStep) * 8u);
                av[u][i] = NT ? __builtin_nontemporal_load(p) : *p;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < R; ++i) { s0 = fma(av[u][i].x, pv[u].x, s0); s1 = fma(av[u][i].y, pv[u].y, s1); }
    }
    if (s0 + s1 == 12345.678) out[blockIdx.x] = s0;
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 4096, ncols = 32768;
    const long pitch = ncols + 16;
    double *A, *v, *out;
    hipMalloc(&A, (size_t)rows * pitch * 8); hipMalloc(&v, (ncols + 64) * 8); hipMalloc(&out, 1 << 20);
    hipMemset(A, 0x11, (size_t)rows * pitch * 8); hipMemset(v, 0, (ncols + 64) * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    constexpr int R = 8;
    const int keeps[] = {0, 256, 512, 768, 896, 1024, 2048};
    hipEvent_t ea, eb;
    hipEventCreate(&ea); hipEventCreate(&eb);
    for (int keep : keeps) {
        if (keep > rows) continue;
        // the resident part with 2 rows per workgroup (4x as many workgroups: a few hundred rows must still fill the chip)
        auto part_a = [&]() { if (keep > 0) hipLaunchKernelGGL((k_rows<2, 8, false>), dim3(keep / 2), dim3(256), 0, 0, A, pitch, ncols, v, out); };
        auto part_b = [&]() {
            if (keep < rows)
                hipLaunchKernelGGL((k_rows<R, 2, true>), dim3((rows - keep) / R), dim3(256), 0, 0, A + (size_t)keep * pitch, pitch, ncols, v, out);
        };
        for (int i = 0; i < 10; ++i) { part_a(); part_b(); }
        const int reps = 40;
        float ta = 0, tb = 0, ms;
        for (int i = 0; i < reps; ++i) {
            hipEventRecord(e0, 0); part_a(); hipEventRecord(ea, 0); part_b(); hipEventRecord(eb, 0);
            hipEventSynchronize(eb);
            hipEventElapsedTime(&ms, e0, ea); ta += ms;
            hipEventElapsedTime(&ms, ea, eb); tb += ms;
        }
        // the resident part alone, back to back: the Infinity Cache's own rate for this access
        float alone = 0;
        if (keep > 0) {
            for (int i = 0; i < 10; ++i) part_a();
            hipEventRecord(e0, 0);
            for (int i = 0; i < reps; ++i) part_a();
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            hipEventElapsedTime(&alone, e0, e1);
        }
        const double mib = keep * (double)ncols * 8 / 1048576.0;
        printf("rows=%d keep=%4d (%4.0f MiB): resident part %.1f us (%.0f GB/s; alone, back to back: %.1f us = %.0f GB/s), streamed part %.1f us "
               "(%.0f GB/s), sweep %.1f us\n", rows, keep, mib, ta / reps * 1e3, keep ? mib * 1048576 / (ta / reps * 1e-3) / 1e9 : 0.0,
               alone / reps * 1e3, keep ? mib * 1048576 / (alone / reps * 1e-3) / 1e9 : 0.0, tb / reps * 1e3,
               8.0 * (rows - keep) * ncols / (tb / reps * 1e-3) / 1e9, (ta + tb) / reps * 1e3);
    }
    return 0;
}
