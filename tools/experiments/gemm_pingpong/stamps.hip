// Phase stamps of the ping-pong GEMM (timing build, variant 9): workgroup 0's waves 0 and 4 stamp the core clock around every
// barrier / segment of the gate-up GEMM at T = 10,922; prints per-segment averages over the recorded stages.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct VVGemmWs { float* partials; unsigned* flags; unsigned* err; };
extern "C" int vv_gemm3_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, int T, int N, int K,
                               int ldy, int epi, const VVGemmWs* ws, hipStream_t s);
extern "C" void vv_gemm_variant_set(int v);
extern "C" void vv_gemm_dbg_set(unsigned long long* p);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
static size_t packed_elems(int rows, int K) { return (size_t)((rows + 15) / 16) * ((K + 31) / 32) * 512; }
int main() {
    const int T = 10922, N = 18944, K = 3584;
    void *W, *W2, *Xp, *Yp; unsigned long long* dbg;
    CK(hipMalloc(&W, packed_elems(N, K) * 2)); CK(hipMalloc(&W2, packed_elems(N, K) * 2)); CK(hipMalloc(&Xp, packed_elems(T, K) * 2));
    CK(hipMalloc(&Yp, packed_elems(T, N) * 2)); CK(hipMalloc(&dbg, 512 * 8));
    CK(hipMemset(W, 0, packed_elems(N, K) * 2)); CK(hipMemset(W2, 0, packed_elems(N, K) * 2)); CK(hipMemset(Xp, 0, packed_elems(T, K) * 2));
    CK(hipMemset(dbg, 0, 512 * 8));
    vv_gemm_dbg_set(dbg); vv_gemm_variant_set(9);
    for (int it = 0; it < 2; ++it) { int rc = vv_gemm3_launch(W, W2, Xp, nullptr, Yp, nullptr, T, N, K, 0, 3, nullptr, 0); if (rc) { printf("rc %d\n", rc); return 1; } CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(512);
    CK(hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost));
    // 6 stamps per stage per half.  half 0: [pre-compute, post-compute+vmcnt, post-barrier, post-read, post-issue, post-lgkm]
    //                               half 1: [pre-read, post-read, post-issue, post-waits, post-barrier, post-compute]
    const char* n0[6] = {"compute+vmcnt", "barrierA wait", "read issue", "dma issue", "lgkm wait", "barrierB wait -> next"};
    const char* n1[6] = {"read issue", "dma issue", "lgkm+vmcnt wait", "barrierA wait", "compute", "barrierB wait -> next"};
    for (int hf = 0; hf < 2; ++hf) {
        double sum[6] = {0}; int cnt = 0;
        for (int s = 4; s < 40; ++s) {            // skip the first stages
            for (int k = 0; k < 6; ++k) {
                const unsigned long long a = h[hf * 256 + s * 6 + k], b = h[hf * 256 + s * 6 + k + 1];
                sum[k] += (double)(b - a);
            }
            ++cnt;
        }
        printf("half %d (avg cycles over %d stages):", hf, cnt);
        double tot = 0;
        for (int k = 0; k < 6; ++k) { printf("  %s %.0f", hf ? n1[k] : n0[k], sum[k] / cnt); tot += sum[k] / cnt; }
        printf("  | stage total %.0f\n", tot);
    }
    printf("raw half0:"); for (int i = 24; i < 48; ++i) printf(" %llu", h[i] - h[24]); printf("\n");
    printf("raw half1:"); for (int i = 24; i < 48; ++i) printf(" %llu", h[256 + i] - h[24]); printf("\n");
    return 0;
}
