// Standalone A/B of the prefill GEMM schedules (prefill.hip): vv_gemm4 (variant 0) against the vv_gemm5 forms (variants 1..4) on
// the four GEMMs of a 7B layer at the benchmark's prompt length, with and without the K-split partial round.  Links
// libvvhip.so; operands are random bf16 bit patterns written straight into the packed layouts (both kernels read the same
// buffers, so equal summation order means bit-identical outputs).
//   hipcc --offload-arch=gfx950 -O2 bench_gemm.hip -o bench_gemm -L../../../vibevoice_amd -lvvhip -Wl,-rpath,'$ORIGIN/../../../vibevoice_amd'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

struct VVGemmWs { float* partials; unsigned* flags; unsigned* err; };
extern "C" int vv_gemm3_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, int T, int N, int K,
                               int ldy, int epi, const VVGemmWs* ws, hipStream_t s);
extern "C" void vv_gemm_variant_set(int v);

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // sign + exponent in [2^-4, 2^-1) + 7 mantissa bits: finite, O(0.1) magnitudes
        const unsigned e = 123u + (h & 3u) % 3u;
        p[i] = (unsigned short)(((h >> 31) << 15) | (e << 7) | ((h >> 8) & 127u));
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((int)(h & 0xffff) - 32768) / 65536.0f;
    }
}
static size_t packed_elems(int rows, int K) { return (size_t)((rows + 15) / 16) * ((K + 31) / 32) * 512; }

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 10922;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    struct Shape { const char* name; int N, K, epi; };
    const Shape shapes[] = {{"gate_up", 18944, 3584, 3}, {"down", 3584, 18944, 4}, {"qkv", 4608, 3584, 1}, {"o", 3584, 3584, 4}};
    hipStream_t st; CK(hipStreamCreate(&st));
    VVGemmWs ws;
    CK(hipMalloc(&ws.partials, (size_t)256 * 32 * 512 * 16));
    CK(hipMalloc(&ws.flags, 1024)); CK(hipMemset(ws.flags, 0, 1024));
    CK(hipHostMalloc((void**)&ws.err, 4, hipHostMallocMapped)); *ws.err = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"T\": %d, \"results\": [\n", T);
    bool first = true;
    for (const Shape& sh : shapes) {
        const size_t nw = packed_elems(sh.N, sh.K), nx = packed_elems(T, sh.K), ny = (size_t)T * sh.N, nyp = packed_elems(T, sh.N);
        unsigned short *W, *W2 = nullptr, *Xp, *Yp = nullptr, *Yp_ref = nullptr;
        float *Y = nullptr, *Y0 = nullptr, *Y_ref = nullptr, *bias;
        CK(hipMalloc(&W, nw * 2)); CK(hipMalloc(&Xp, nx * 2)); CK(hipMalloc(&bias, sh.N * 4));
        fill_bf16<<<2048, 256, 0, st>>>(W, nw, 11u); fill_bf16<<<2048, 256, 0, st>>>(Xp, nx, 23u); fill_f32<<<64, 256, 0, st>>>(bias, sh.N, 5u);
        if (sh.epi == 3) { CK(hipMalloc(&W2, nw * 2)); fill_bf16<<<2048, 256, 0, st>>>(W2, nw, 37u); CK(hipMalloc(&Yp, nyp * 2)); CK(hipMalloc(&Yp_ref, nyp * 2)); }
        else { CK(hipMalloc(&Y, ny * 4)); CK(hipMalloc(&Y0, ny * 4)); CK(hipMalloc(&Y_ref, ny * 4)); fill_f32<<<2048, 256, 0, st>>>(Y0, ny, 77u); }
        CK(hipStreamSynchronize(st));
        for (int use_ws = 0; use_ws < 2; ++use_ws) {
            for (int v : {400, 401, 410, 411, 810, 811, 210}) {   // sfw * 100 + schedule
                vv_gemm_variant_set(v);
                const VVGemmWs* w = use_ws ? &ws : nullptr;
                auto run = [&]() {
                    if (sh.epi == 3) return vv_gemm3_launch(W, W2, Xp, nullptr, Yp, nullptr, T, sh.N, sh.K, 0, 3, w, st);
                    return vv_gemm3_launch(W, nullptr, Xp, Y, nullptr, sh.epi == 1 ? bias : nullptr, T, sh.N, sh.K, sh.N, sh.epi, w, st);
                };
                // correctness run (residual epilogue accumulates: start from Y0 every time)
                if (sh.epi == 3) CK(hipMemsetAsync(Yp, 0, nyp * 2, st)); else CK(hipMemcpyAsync(Y, Y0, ny * 4, hipMemcpyDeviceToDevice, st));
                int rc = run();
                CK(hipStreamSynchronize(st));
                if (rc) { fprintf(stderr, "launch rc %d\n", rc); return 3; }
                long long ndiff = -1;
                if (v == 400 && use_ws == 0) {
                    if (sh.epi == 3) CK(hipMemcpy(Yp_ref, Yp, nyp * 2, hipMemcpyDeviceToDevice)); else CK(hipMemcpy(Y_ref, Y, ny * 4, hipMemcpyDeviceToDevice));
                    ndiff = 0;
                } else {
                    const size_t nb = sh.epi == 3 ? nyp * 2 : ny * 4;
                    std::vector<unsigned char> a(nb), b(nb);
                    CK(hipMemcpy(a.data(), sh.epi == 3 ? (void*)Yp : (void*)Y, nb, hipMemcpyDeviceToHost));
                    CK(hipMemcpy(b.data(), sh.epi == 3 ? (void*)Yp_ref : (void*)Y_ref, nb, hipMemcpyDeviceToHost));
                    if (use_ws == 0 || sh.epi == 3) {          // same summation order: bytes must match (split: packed bf16 may differ in the last bit)
                        ndiff = 0;
                        if (use_ws == 0) { ndiff = memcmp(a.data(), b.data(), nb) ? 1 : 0; if (ndiff) { ndiff = 0; for (size_t i = 0; i < nb; ++i) ndiff += a[i] != b[i]; } }
                        else { const unsigned short* pa = (const unsigned short*)a.data(); const unsigned short* pb = (const unsigned short*)b.data();
                               for (size_t i = 0; i < nb / 2; ++i) { int d = (int)pa[i] - (int)pb[i]; if (d < -1 || d > 1) ++ndiff; } }
                    } else {                                     // split partial round: fp32 sums in a different order
                        const float* pa = (const float*)a.data(); const float* pb = (const float*)b.data();
                        ndiff = 0; double worst = 0;
                        for (size_t i = 0; i < ny; ++i) { double d = pa[i] - pb[i]; if (d < 0) d = -d; double tol = 1e-4 * (1.0 + (pb[i] < 0 ? -pb[i] : pb[i])); if (d > tol) ++ndiff; if (d > worst) worst = d; }
                    }
                }
                // timing
                for (int i = 0; i < 2; ++i) run();
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < reps; ++i) run();
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1000.0 / reps;
                const double tf = 2.0 * T * (double)sh.N * sh.K * (sh.epi == 3 ? 2.0 : 1.0) / (us * 1e-6) / 1e12;
                printf("%s {\"gemm\": \"%s\", \"ksplit\": %d, \"variant\": %d, \"us\": %.1f, \"tflops\": %.1f, \"mismatch\": %lld, \"err\": %u}",
                       first ? " " : ",\n ", sh.name, use_ws, v, us, tf, ndiff, *ws.err);
                first = false;
                fflush(stdout);
            }
        }
        hipFree(W); hipFree(Xp); hipFree(bias); if (W2) hipFree(W2); if (Yp) hipFree(Yp); if (Yp_ref) hipFree(Yp_ref);
        if (Y) hipFree(Y); if (Y0) hipFree(Y0); if (Y_ref) hipFree(Y_ref);
    }
    printf("\n]}\n");
    return 0;
}
