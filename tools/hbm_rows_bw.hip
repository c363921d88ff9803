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

// SPLIT > 1: the columns of a row group are cut into SPLIT pieces, one workgroup each (SPLIT times as many, smaller
// workgroups: finer units for the dispatcher to balance).
template <int R, int U, int VEC, int SPLIT = 1>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ A, long pitch, int ncols, const double* __restrict__ v,
                                              const double* __restrict__ r, double* out)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int g = blockIdx.x / SPLIT, piece = blockIdx.x % SPLIT;
    const char* a[R];
#pragma unroll
    for (int i = 0; i < R; ++i) a[i] = reinterpret_cast<const char*>(A + ((long)g * R + i) * pitch);
    double s0 = 0, s1 = 0;
    constexpr int kStep = 512;
    const int c_lo = piece * (ncols / SPLIT), c_hi = (piece + 1) * (ncols / SPLIT);
    for (int c = c_lo + w * 128 + lane * 2; c + (U - 1) * kStep < c_hi; c += U * kStep) {
        d2 av[U][R], pv[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = (unsigned)(c + u * kStep) * 8u;
            if (VEC != 0) {
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
            if (VEC != 0) { px = pv[u].x + rv[u].x; py = pv[u].y + rv[u].y; }
#pragma unroll
            for (int i = 0; i < R; ++i) { s0 = fma(av[u][i].x, px, s0); s1 = fma(av[u][i].y, py, s1); }
        }
    }
    if (VEC == 2) {
        // K1's epilogue, schematically: R butterflies' worth of exchanges shared, LDS combine over the 4 waves, one store
        __shared__ double red[4][R];
        double acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = s0 + i * s1;
#pragma unroll
        for (int i = 0; i < R; ++i)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < R; ++i) red[w][i] = acc[i];
        __syncthreads();
        if (w == 0 && lane < R) {
            double t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
            t *= v[blockIdx.x * R + lane];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
            if (lane == 0) out[blockIdx.x] = t;
        }
        return;
    }
    if (s0 + s1 == 12345.678) out[blockIdx.x] = s0;
}

static int g_blocks = 1;          // > 1: successive launches read different row blocks (what P logical shards on one GPU do;
static size_t g_block_stride = 0;   // a real rank re-reads its ONE block every iteration)
template <int R, int U, int VEC, int SPLIT = 1>
float run(const double* A0, long pitch, int rows, int ncols, const double* v, const double* r, double* out, hipEvent_t e0, hipEvent_t e1)
{
    const int grid = rows / R * SPLIT;
    int launch_no = 0;
#define A (A0 + (size_t)(launch_no++ % g_blocks) * g_block_stride)
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_rows<R, U, VEC, SPLIT>), dim3(grid), dim3(256), 0, 0, A, pitch, ncols, v, r, out);
    const int reps = 40;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_rows<R, U, VEC, SPLIT>), dim3(grid), dim3(256), 0, 0, A, pitch, ncols, v, r, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
#undef A
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
    const int pads[] = {16};
    g_blocks = argc > 3 ? atoi(argv[3]) : 1;
    for (int pad : pads) {
        const long pitch = ncols + pad;
        double* A;
        g_block_stride = (size_t)rows * pitch;
        hipMalloc(&A, g_block_stride * 8 * g_blocks);
        hipMemset(A, 0x11, g_block_stride * 8 * g_blocks);
        printf("-- %d block(s) of %d x %d, launches rotate over them\n", g_blocks, rows, ncols);
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
            {"R=8 U=2 +vec +old epilogue (R butterflies)", run<8, 2, 2>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=4 U=4 +vec +old epilogue (R butterflies)", run<4, 4, 2>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=8 U=2 +vec, columns split 2", run<8, 2, 1, 2>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=8 U=2 +vec, columns split 4", run<8, 2, 1, 4>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=8 U=2 +vec, columns split 8", run<8, 2, 1, 8>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=16 U=1 +vec, columns split 4", run<16, 1, 1, 4>(A, pitch, rows, ncols, v, r, out, e0, e1)},
            {"R=4 U=4 +vec, columns split 2", run<4, 4, 1, 2>(A, pitch, rows, ncols, v, r, out, e0, e1)},
        };
        for (auto& x : res)
            printf("rows=%d ncols=%d pad=%2d  %s : %.4f ms per launch incl. boundary  %.1f GB/s\n", rows, ncols, pad, x.name, x.ms,
                   bytes / (x.ms * 1e-3) / 1e9);
        hipFree(A);
    }
    return 0;
}
