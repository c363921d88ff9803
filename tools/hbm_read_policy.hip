// Dev tool (round 4): does any cache-policy flavour of the 16-B streaming load read faster than `nt` (what K1 uses)?
// Same shape as tools/hbm_read_bw.hip (4096 workgroups x 256 threads, U = 16 loads of 16 B in flight per lane), the load
// written as inline asm with every combination of sc0 / sc1 / nt, plus the LDS-DMA form (global_load_lds_dwordx4, no VGPR
// destination).  hipcc --offload-arch=gfx950 -O3 tools/hbm_read_policy.hip -o /tmp/hbm_read_policy && /tmp/hbm_read_policy [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

#define LOADER(NAME, MODS)                                                                                    \
    __device__ __forceinline__ f4 NAME(const void *p)                                                         \
    {                                                                                                         \
        f4 v;                                                                                                 \
        asm volatile("global_load_dwordx4 %0, %1, off " MODS : "=v"(v) : "v"(p) : "memory");                  \
        return v;                                                                                             \
    }
LOADER(ld_plain, "")
LOADER(ld_nt, "nt")
LOADER(ld_sc0, "sc0")
LOADER(ld_sc1, "sc1")
LOADER(ld_sc0sc1, "sc0 sc1")
LOADER(ld_sc0nt, "sc0 nt")
LOADER(ld_sc1nt, "sc1 nt")
LOADER(ld_sc0sc1nt, "sc0 sc1 nt")

template <int WHICH, int U>
__global__ __launch_bounds__(256) void k_read(const char *__restrict__ a, size_t bytes_per_wg, float *out)
{
    const char *base = a + (size_t)blockIdx.x * bytes_per_wg + (size_t)threadIdx.x * 16;
    float s = 0.f;
    for (size_t c = 0; c + (size_t)(U - 1) * 4096 < bytes_per_wg; c += (size_t)U * 4096) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const void *p = base + c + (size_t)u * 4096;
            if (WHICH == 0) v[u] = ld_plain(p);
            else if (WHICH == 1) v[u] = ld_nt(p);
            else if (WHICH == 2) v[u] = ld_sc0(p);
            else if (WHICH == 3) v[u] = ld_sc1(p);
            else if (WHICH == 4) v[u] = ld_sc0sc1(p);
            else if (WHICH == 5) v[u] = ld_sc0nt(p);
            else if (WHICH == 6) v[u] = ld_sc1nt(p);
            else v[u] = ld_sc0sc1nt(p);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) s += v[u].x + v[u].w;
    }
    if (s == 12345.678f) out[blockIdx.x] = s;
}

// LDS-DMA: the data goes straight to LDS (one 16-B slot per lane and load in flight), nothing is summed
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read_lds(const char *__restrict__ a, size_t bytes_per_wg, float *out)
{
    __shared__ f4 buf[U * 256];
    const char *base = a + (size_t)blockIdx.x * bytes_per_wg + (size_t)threadIdx.x * 16;
    for (size_t c = 0; c + (size_t)(U - 1) * 4096 < bytes_per_wg; c += (size_t)U * 4096) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const void *p = base + c + (size_t)u * 4096;
            // M0 holds the LDS base of the wave's destination; each lane writes at base + lane * 16
            const unsigned lds_off = (unsigned)(size_t)(&buf[u * 256 + (threadIdx.x & ~63)]);
            asm volatile("s_mov_b32 m0, %0" : : "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
            if (NT) asm volatile("global_load_lds_dwordx4 %0, off nt" : : "v"(p) : "memory");
            else asm volatile("global_load_lds_dwordx4 %0, off" : : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (buf[threadIdx.x].x == 12345.678f) out[blockIdx.x] = buf[threadIdx.x].y;
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 8192ull) << 20;
    char *a; float *out;
    // argv[2]: how the buffer is allocated: 0 = hipMalloc (default), 1 = fine-grained, 2 = uncached
    const int kind = argc > 2 ? atoi(argv[2]) : 0;
    hipError_t ae = kind == 0 ? hipMalloc(&a, bytes)
                              : hipExtMallocWithFlags(reinterpret_cast<void **>(&a), bytes, kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
    printf("allocation kind %d (%s): %s\n", kind, kind == 0 ? "hipMalloc" : (kind == 1 ? "fine-grained" : "uncached"), hipGetErrorString(ae));
    if (ae != hipSuccess) return 1;
    (void)hipMalloc(&out, 1 << 20);
    (void)hipMemset(a, 0x11, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int wgs = 4096;
    const size_t per = bytes / wgs;
    const char *names[10] = {"plain", "nt", "sc0", "sc1", "sc0 sc1", "sc0 nt", "sc1 nt", "sc0 sc1 nt", "lds-dma", "lds-dma nt"};
    for (int rep = 0; rep < 2; ++rep)
        for (int w = 0; w < 10; ++w) {
            auto launch = [&]() {
                switch (w) {
                case 0: hipLaunchKernelGGL((k_read<0, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 1: hipLaunchKernelGGL((k_read<1, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 2: hipLaunchKernelGGL((k_read<2, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 3: hipLaunchKernelGGL((k_read<3, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 4: hipLaunchKernelGGL((k_read<4, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 5: hipLaunchKernelGGL((k_read<5, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 6: hipLaunchKernelGGL((k_read<6, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 7: hipLaunchKernelGGL((k_read<7, 16>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                case 8: hipLaunchKernelGGL((k_read_lds<false, 8>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                default: hipLaunchKernelGGL((k_read_lds<true, 8>), dim3(wgs), dim3(256), 0, 0, a, per, out); break;
                }
            };
            for (int i = 0; i < 10; ++i) launch();
            (void)hipEventRecord(e0, 0);
            const int reps = 30;
            for (int i = 0; i < reps; ++i) launch();
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("read %zu MiB  %-12s : %.4f ms  %.1f GB/s  (%.1f %% of 8 TB/s)  %s\n", bytes >> 20, names[w], ms / reps,
                   bytes / (ms / reps * 1e-3) / 1e9, bytes / (ms / reps * 1e-3) / 8e12 * 100, hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
