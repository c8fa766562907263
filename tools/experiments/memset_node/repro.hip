// Stand-alone question: does a MEMSET NODE of a replayed hipGraph keep writing the value it was captured with while other graph
// executables are created, launched and destroyed in the same process?  (profiles/r06_memset_node_ab.txt: inside the
// engine it did not.)   hipcc --offload-arch=gfx950 -O2 repro.hip -o repro && ./repro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); return 2; } } while (0)
__global__ void touch(unsigned* p, unsigned v) { if (threadIdx.x == 0 && blockIdx.x == 0) atomicMax(p, v); }
__global__ void work(float* x, int n, float a) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * a + 1.0f; }

static int check(const char* when, hipGraphExec_t A, hipStream_t st, unsigned* buf, int words, int& bad_total) {
    std::vector<unsigned> h(words);
    CK(hipGraphLaunch(A, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), buf, words * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 1; i < words; ++i) if (h[i]) bad++;
    if (bad) { printf("  %-46s memset node left %d of %d words non-zero: %08x %08x %08x %08x\n", when, bad, words - 1, h[4], h[5], h[6], h[7]); bad_total++; }
    return 0;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int words = 4096;
    unsigned* buf; CK(hipMalloc(&buf, words * 4)); CK(hipMemset(buf, 0xAB, words * 4));
    float* x; const int n = 1 << 20; CK(hipMalloc(&x, n * 4)); CK(hipMemset(x, 0, n * 4));
    // graph A: memset node + a few hundred kernel nodes (the shape of the engine's sampler sequence)
    hipGraph_t g; hipGraphExec_t A;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    CK(hipMemsetAsync(buf, 0, words * 4, st));
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, buf, 7u);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, x, n, 0.5f);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&A, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    int bad = 0;
    for (int i = 0; i < 100; ++i) if (check("100 plain replays", A, st, buf, words, bad)) return 2;
    printf("after 100 plain replays: %d bad\n", bad);
    // other executables come and go: kernel-only graphs, graphs with their own memset nodes, replayed between launches of A
    for (int round = 0; round < 6; ++round) {
        hipGraph_t g2; hipGraphExec_t B;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        if (round & 1) CK(hipMemsetAsync(x, 0, 4096, st));
        for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, x, n, 0.25f);
        CK(hipStreamEndCapture(st, &g2)); CK(hipGraphInstantiate(&B, g2, nullptr, nullptr, 0)); CK(hipGraphDestroy(g2));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipGraphLaunch(B, st)); CK(hipEventRecord(e0, st));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(B, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
        CK(hipGraphExecDestroy(B));
        char w[64]; snprintf(w, 64, "after executable %d came and went", round);
        for (int i = 0; i < 20; ++i) if (check(w, A, st, buf, words, bad)) return 2;
        // device memory churn + a new long-lived executable
        void* t; CK(hipMalloc(&t, 64 << 20)); CK(hipFree(t));
        hipGraph_t g3; hipGraphExec_t C;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, st, x, n, 0.75f);
        CK(hipStreamEndCapture(st, &g3)); CK(hipGraphInstantiate(&C, g3, nullptr, nullptr, 0)); CK(hipGraphDestroy(g3));
        CK(hipGraphLaunch(C, st));
        snprintf(w, 64, "after a new executable was kept (%d)", round);
        for (int i = 0; i < 20; ++i) if (check(w, A, st, buf, words, bad)) return 2;
    }
    // eager fills / copies of other buffers with other values between replays of A
    unsigned* other; CK(hipMalloc(&other, 1 << 20));
    std::vector<unsigned> hostbuf(1 << 18, 0x12345678u);
    for (int round = 0; round < 8; ++round) {
        const int n_eager = 1 << (2 * round);                     // 1, 4, 16, ... 16384 eager operations
        for (int i = 0; i < n_eager; ++i) {
            if (round & 1) CK(hipMemsetAsync(other, 0x5A, 4096 + 64 * (i & 63), st));
            else CK(hipMemsetD32Async((hipDeviceptr_t)other, 0xFFC01234u + i, 1024, st));
        }
        CK(hipMemcpyAsync(other, hostbuf.data(), 1 << 20, hipMemcpyHostToDevice, st));      // pageable host source: staged by the runtime
        CK(hipMemcpyAsync(other + 1024, other, 4096, hipMemcpyDeviceToDevice, st));
        CK(hipStreamSynchronize(st));
        char w[64]; snprintf(w, 64, "after %d eager fills of another buffer", n_eager);
        for (int i = 0; i < 10; ++i) if (check(w, A, st, buf, words, bad)) return 2;
    }
    // fills on the NULL stream and synchronous ones (what an allocator that zeroes its buffers does)
    for (int i = 0; i < 2000; ++i) CK(hipMemset(other, 0x77, 256 + 16 * (i & 255)));
    for (int i = 0; i < 10; ++i) if (check("after 2000 synchronous hipMemset calls", A, st, buf, words, bad)) return 2;
    printf("total replays of A that found a non-zero fill: %d\n", bad);
    return bad ? 1 : 0;
}
