// Dev tool: what does the memory system deliver for K1's ACCESS PATTERN alone (no vector operands, no FMAs that matter)?
// A row block of `rows` x `ncols` doubles at pitch `pitch`; one workgroup of 4 waves owns R consecutive rows and sweeps
// them in lock step: per step every wave loads 1 KiB (lane = 16 B) of each of the R rows, U steps per trip -- exactly the
// loads of k_gemv_colsplit<R,U,4>.  VEC=1 adds the two L2-resident vector loads per step (p_old and r).
// hipcc --offload-arch=gfx950 -O3 tools/hbm_rows_bw.hip -o /tmp/hbm_rows_bw && /tmp/hbm_rows_bw [rows] [ncols]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int R, int U, int VEC>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ A, long pitch, int ncols, const double* __restrict__ v,
                                              const double* __restrict__ r, double* out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const char* a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = reinterpret_cast<const char*>(A + ((long)blockIdx.x * R + i) * pitch);
    double s0 = 0, s1 = 0;
    constexpr int kStep = 512;
    for (int c = w * 128 + lane * 2; c + (U - 1) * kStep < ncols; c += U * kStep) {
        d2 av[U][R], pv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = (unsigned)(c + u * kStep) * 8u;
            if (VEC) {
                pv[u] = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(v) + off);
                rv[u] = *reinterpret_cast<const d2*>(reinterpret_cast<const char*>(r) + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < R; ++i)
                av[u][i] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(a[i] + (unsigned)(c + u * kStep) * 8u));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            double px = 1.0, py = 1.0;
            if (VEC) { px = pv[u].x + rv[u].x; py = pv[u].y + rv[u].y; }
#pragma unroll
            for (int i = 0; i < R; ++i) { s0 = fma(av[u][i].x, px, s0); s1 = fma(av[u][i].y, py, s1); }
        }
    }
    if (s0 + s1 == 12345.678) out[blockIdx.x] = s0;
}

template <int R, int U, int VEC>
float run(const double* A, long pitch, int rows, int ncols, const double* v, const double* r, double* out, hipEvent_t e0, hipEvent_t e1)
{
    const int grid = rows / R;
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_rows<R, U, VEC>), dim3(grid), dim3(256), 0, 0, A, pitch, ncols, v, r, out);
    const int reps = 40;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rows<R, U, VEC>), dim3(grid), dim3(256), 0, 0, A, pitch, ncols, v, r, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 4096, ncols = argc > 2 ? atoi(argv[2]) : 32768;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    double *v, *r, *out;
    hipMalloc(&v, (ncols + 64) * 8); hipMalloc(&r, (ncols + 64) * 8); hipMalloc(&out, 1 << 20);
    hipMemset(v, 0, (ncols + 64) * 8); hipMemset(r, 0, (ncols + 64) * 8);
    const int pads[] = {0, 16, 32};
    for (int pad : pads) {
        const long pitch = ncols + pad;
        double* A;
        hipMalloc(&A, (size_t)rows * pitch * 8);
        hipMemset(A, 0x11, (size_t)rows * pitch * 8);
        const double bytes = 8.0 * rows * ncols;
        struct { const char* name; float ms; } res[] = {
            {"R=8 U=2      ", run<8, 2, 0>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=8 U=2 +vec ", run<8, 2, 1>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=4 U=4      ", run<4, 4, 0>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=4 U=4 +vec ", run<4, 4, 1>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=2 U=8      ", run<2, 8, 0>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=1 U=16     ", run<1, 16, 0>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=16 U=1     ", run<16, 1, 0>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=16 U=1 +vec", run<16, 1, 1>(A, pitch, rows, ncols, v, r, out, e0, e1)},
        };
        for (auto& x : res)
            printf("rows=%d ncols=%d pad=%2d  %s : %.4f ms per launch incl. boundary  %.1f GB/s\n", rows, ncols, pad, x.name, x.ms,
                   bytes / (x.ms * 1e-3) / 1e9);
        hipFree(A);
    }
    return 0;
}
