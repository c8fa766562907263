// Measures the fixed cost of dependent kernel launches on this box: eager vs hipGraph replay,
// trivial kernels of 1 / 256 / 2048 workgroups.  hipcc --offload-arch=gfx950 -O3 tools/launch_floor.hip -o /tmp/lf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
__global__ void k_empty() {}
int main() {
    float* d; hipMalloc(&d, 1 << 22);
    hipMemset(d, 0, 1 << 22);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 600;
    for (int blocks : {1, 256, 2048}) {
        for (int mode = 0; mode < 2; ++mode) {           // 0 empty, 1 touch
            // eager
            for (int w = 0; w < 2; ++w) {
                auto t0 = std::chrono::high_resolution_clock::now();
                for (int i = 0; i < N; ++i) { if (mode) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, d, blocks * 256); else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s); }
                hipStreamSynchronize(s);
                auto t1 = std::chrono::high_resolution_clock::now();
                if (w) printf("blocks %4d %s eager : %.2f us/kernel\n", blocks, mode ? "touch" : "empty", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
            }
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
            for (int i = 0; i < N; ++i) { if (mode) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, d, blocks * 256); else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s); }
            hipStreamEndCapture(s, &g);
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int w = 0; w < 3; ++w) {
                auto t0 = std::chrono::high_resolution_clock::now();
                hipGraphLaunch(ge, s); hipStreamSynchronize(s);
                auto t1 = std::chrono::high_resolution_clock::now();
                if (w == 2) printf("blocks %4d %s graph : %.2f us/kernel\n", blocks, mode ? "touch" : "empty", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
            }
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
