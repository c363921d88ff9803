// Dev tool: ceiling of a pure streaming READ on this MI355X (what K1 can reach at most).
// hipcc --offload-arch=gfx950 -O3 tools/hbm_read_bw.hip -o /tmp/hbm_read_bw && /tmp/hbm_read_bw [MiB] [const|hash]
// const: every byte 0x11 (rounds 1-3); hash: every double a different number in [-1, 1) with a random mantissa (the
// splitmix64 finaliser of the element index: the data of cgx_probe_fill_matrix_hash) -- is the ceiling data dependent?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_read(const double* __restrict__ a, size_t n_per_wg, double* out)
{
    const double* base = a + (size_t)blockIdx.x * n_per_wg;
    double s0 = 0, s1 = 0;
    for (size_t c = (size_t)threadIdx.x * 2; c + (U - 1) * 512 < n_per_wg; c += U * 512) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const d2* p = reinterpret_cast<const d2*>(base + c + u * 512);
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) { s0 += v[u].x; s1 += v[u].y; }
    }
    if (s0 + s1 == 12345.678) out[blockIdx.x] = s0;   // keep the loads alive
}
__global__ __launch_bounds__(256) void k_fill_hash(double* a, size_t n)
{
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        unsigned long long z = t + 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        a[t] = (double)(z >> 11) * 0x1.0p-52 - 1.0;
    }
}
int main(int argc, char** argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 8192ull) << 20;   // MiB
    double *a, *out;
    hipMalloc(&a, bytes); hipMalloc(&out, 1 << 20);
    const bool hash = argc > 2 && argv[2][0] == 'h';
    if (hash) hipLaunchKernelGGL(k_fill_hash, dim3(8192), dim3(256), 0, 0, a, bytes / 8);
    else hipMemset(a, 0x11, bytes);
    hipDeviceSynchronize();
    printf("fill: %s\n", hash ? "hash (incompressible doubles in [-1,1))" : "const 0x11");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { int wgs; int u; bool nt; };
    std::vector<Cfg> cfgs = {{4096, 8, true}, {4096, 16, true}, {2048, 16, true}, {8192, 8, true}, {1024, 16, true}, {512, 16, true}, {4096, 16, false}, {16384, 8, true}};
    for (int rep = 0; rep < 2; ++rep)
    for (auto c : cfgs) {
        size_t n_per_wg = bytes / 8 / c.wgs;
        auto launch = [&]() {
            if (c.u == 8 && c.nt) hipLaunchKernelGGL((k_read<8, true>), dim3(c.wgs), dim3(256), 0, 0, a, n_per_wg, out);
            else if (c.u == 16 && c.nt) hipLaunchKernelGGL((k_read<16, true>), dim3(c.wgs), dim3(256), 0, 0, a, n_per_wg, out);
            else hipLaunchKernelGGL((k_read<16, false>), dim3(c.wgs), dim3(256), 0, 0, a, n_per_wg, out);
        };
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e0, 0);
        const int reps = 50;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("read %zu MiB  wgs=%5d U=%2d nt=%d : %.4f ms  %.1f GB/s  (%.1f %% of 8 TB/s)\n", bytes >> 20, c.wgs, c.u, (int)c.nt, ms / reps,
               bytes / (ms / reps * 1e-3) / 1e9, bytes / (ms / reps * 1e-3) / 8e12 * 100);
    }
    return 0;
}
