// probe.hip -- does the 256 MiB Infinity Cache (MALL) serve a weight stream faster than HBM, and can a concurrent prefetch kernel
// put the NEXT kernel's weights there while the current one streams?   hipcc --offload-arch=gfx950 -O3 probe.hip -o probe
//   cold      : stream X after 1.5 GB of other data has gone through the caches
//   warm      : stream X right after a prefetch kernel has read X once
//   overlap   : stream X0 on stream A while a 256-wave prefetcher reads X1 on stream B; then stream X1
// Loads are 1 KiB per wave instruction, non-temporal (as the decode GEMV issues them); every wave keeps 16 KiB in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("hip error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ p, size_t n_vec, unsigned* __restrict__ sink) {
    const size_t wave_g = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (size_t)gridDim.x * 4;
    const unsigned lane = threadIdx.x & 63;
    u32x4 acc = {0u, 0u, 0u, 0u};
    // a wave owns a contiguous slice; 16 loads (16 KiB) in flight
    const size_t per = (n_vec / 64 + n_waves - 1) / n_waves;     // KiB-tiles per wave
    const size_t t0 = wave_g * per, t1 = min(n_vec / 64, t0 + per);
    for (size_t t = t0; t < t1; t += 16) {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const size_t tt = min(t + u, t1 - 1);
            if (NT) v[u] = __builtin_nontemporal_load(p + tt * 64 + lane); else v[u] = p[tt * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;     // never true for random data: keeps the loads alive
}

static float run(hipStream_t s, int grid, int nt, const void* p, size_t bytes, unsigned* sink) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    if (nt) hipLaunchKernelGGL(stream_read<1>, dim3(grid), dim3(256), 0, s, (const u32x4*)p, bytes / 16, sink);
    else hipLaunchKernelGGL(stream_read<0>, dim3(grid), dim3(256), 0, s, (const u32x4*)p, bytes / 16, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

int main(int argc, char** argv) {
    const size_t MB = 1 << 20;
    const size_t total = 4096 * MB;
    char* buf; CK(hipMalloc(&buf, total));
    unsigned* sink; CK(hipMalloc(&sink, 256));
    CK(hipMemset(buf, 0x5a, total));
    hipStream_t sa, sb; CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    const int grid = 768;                        // 3 workgroups of 4 waves per CU, like the decode GEMV
    for (size_t xmb : {32, 64, 128, 192}) {
        const size_t X = xmb * MB;
        char* x0 = buf; char* x1 = buf + 512 * MB; char* flush = buf + 1024 * MB;
        float cold = 0, warm = 0, warm_c = 0;
        for (int rep = 0; rep < 3; ++rep) {
            run(sa, grid, 1, flush, 1536 * MB, sink);                     // push everything else through the caches
            cold = run(sa, grid, 1, x0, X, sink);
            run(sa, grid, 1, flush, 1536 * MB, sink);
            run(sa, 256, 1, x0, X, sink);                                  // "prefetch": one wave-quad per CU reads X with nt loads
            warm = run(sa, grid, 1, x0, X, sink);
            run(sa, grid, 1, flush, 1536 * MB, sink);
            run(sa, 256, 0, x0, X, sink);                                  // prefetch with cacheable loads
            warm_c = run(sa, grid, 1, x0, X, sink);
        }
        // overlap: A streams x0 (cold) while B prefetches x1; then A streams x1
        run(sa, grid, 1, flush, 1536 * MB, sink);
        hipEvent_t e0, e1, e2, eb; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&eb));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa));
        hipLaunchKernelGGL(stream_read<1>, dim3(64), dim3(256), 0, sb, (const u32x4*)x1, X / 16, sink + 8);   // 1 wave per CU prefetcher
        CK(hipEventRecord(eb, sb));
        hipLaunchKernelGGL(stream_read<1>, dim3(grid), dim3(256), 0, sa, (const u32x4*)x0, X / 16, sink);
        CK(hipEventRecord(e1, sa));
        CK(hipStreamWaitEvent(sa, eb, 0));
        hipLaunchKernelGGL(stream_read<1>, dim3(grid), dim3(256), 0, sa, (const u32x4*)x1, X / 16, sink);
        CK(hipEventRecord(e2, sa));
        CK(hipDeviceSynchronize());
        float t_a0 = 0, t_a1 = 0; CK(hipEventElapsedTime(&t_a0, e0, e1)); CK(hipEventElapsedTime(&t_a1, e1, e2));
        printf("{\"mb\": %zu, \"cold_us\": %.1f, \"cold_TBps\": %.2f, \"warm_nt_prefetch_us\": %.1f, \"warm_TBps\": %.2f, \"warm_cacheable_prefetch_us\": %.1f, \"warm_c_TBps\": %.2f, "
               "\"overlap_first_us\": %.1f, \"overlap_second_us\": %.1f, \"overlap_second_TBps\": %.2f}\n",
               xmb, cold * 1e3, X / 1e12 / (cold * 1e-3), warm * 1e3, X / 1e12 / (warm * 1e-3), warm_c * 1e3, X / 1e12 / (warm_c * 1e-3),
               t_a0 * 1e3, t_a1 * 1e3, X / 1e12 / (t_a1 * 1e-3));
    }
    return 0;
}
