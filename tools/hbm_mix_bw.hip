// Dev tool: what does the memory system deliver for K1b's MIX of streams -- NIN arrays read once (16 B per lane, the
// first 5 non-temporal like the diagonals), NOUT arrays written once -- with nothing else in the kernel?
// hipcc --offload-arch=gfx950 -O3 tools/hbm_mix_bw.hip -o /tmp/hbm_mix_bw && /tmp/hbm_mix_bw [log2 elements per array]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int NIN, int NOUT>
__global__ __launch_bounds__(256) void k_mix(const double* __restrict__ in, double* __restrict__ out, size_t n)
{
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2; i < n; i += (size_t)gridDim.x * 512) {
        d2 v[NIN];
#pragma unroll
        for (int s = 0; s < NIN; ++s) {
            const d2* p = reinterpret_cast<const d2*>(in + (size_t)s * n + i);
            v[s] = s < 5 ? __builtin_nontemporal_load(p) : *p;
        }
        __builtin_amdgcn_sched_barrier(0);
        d2 acc = v[0];
#pragma unroll
        for (int s = 1; s < NIN; ++s) { acc.x = fma(acc.x, 0.5, v[s].x); acc.y = fma(acc.y, 0.5, v[s].y); }
#pragma unroll
        for (int s = 0; s < NOUT; ++s) {
            d2 o = acc;
            o.x += s;
            *reinterpret_cast<d2*>(out + (size_t)s * n + i) = o;
        }
    }
}
template <int NIN, int NOUT>
void run(const double* in, double* out, size_t n, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_mix<NIN, NOUT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    const int reps = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_mix<NIN, NOUT>), dim3(grid), dim3(256), 0, 0, in, out, n);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 8.0 * n * (NIN + NOUT);
    printf("n=%zu  %d in + %d out, grid %5d : %.1f us per launch  %.0f GB/s  (%.1f %% of 8 TB/s)\n", n, NIN, NOUT, grid, ms / reps * 1e3,
           bytes / (ms / reps * 1e-3) / 1e9, bytes / (ms / reps * 1e-3) / 8e12 * 100);
}
int main(int argc, char** argv)
{
    const size_t n = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 24);
    double *in, *out;
    hipMalloc(&in, n * 8 * 7); hipMalloc(&out, n * 8 * 2);
    hipMemset(in, 0, n * 8 * 7);
    for (int grid : {2048, 8192, 32768}) {
        run<7, 2>(in, out, n, grid);
        run<7, 0>(in, out, n, grid);
        run<5, 2>(in, out, n, grid);
        run<1, 1>(in, out, n, grid);
    }
    return 0;
}
