// Standalone A/B of the prefill attention schedules (prefill.hip) at the benchmark's shape: R = 10,922 consecutive positions of
// one cache, 28 query / 4 kv heads x 128.  Links libvvhip.so; K / V / q are random (finite) values written straight into the
// tile layouts; variant 0 = vv_attn_prefill3_kernel, the reference for the bitwise comparison (same arithmetic order).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
struct VVRow { int cache; int pos; };
extern "C" int vv_attn_prefill3_launch(int D, const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq, int Hkv,
                                       int64_t cache_stride, int64_t head_stride, float* out, hipStream_t s);
extern "C" void vv_attn_variant_set(int v);
extern "C" void vv_attn_dbg_set(unsigned long long* p);
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned e = 124u + (h & 3u) % 3u;
        p[i] = (unsigned short)(((h >> 31) << 15) | (e << 7) | ((h >> 8) & 127u));
    }
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = scale * ((int)(h & 0xffff) - 32768) / 32768.0f;
    }
}
int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 10922, Hq = 28, Hkv = 4, D = 128, reps = 5;
    const int Lpad = (R + 63) / 64 * 64 + 64;
    const int64_t head_stride = (int64_t)Lpad * D, cache_stride = head_stride * Hkv;
    unsigned short *kc, *vc; float *q, *out, *ref; VVRow* rows; unsigned long long* dbg;
    CK(hipMalloc(&kc, cache_stride * 2)); CK(hipMalloc(&vc, cache_stride * 2));
    CK(hipMalloc(&q, (size_t)R * Hq * D * 4)); CK(hipMalloc(&out, (size_t)R * Hq * D * 4)); CK(hipMalloc(&ref, (size_t)R * Hq * D * 4));
    CK(hipMalloc(&rows, sizeof(VVRow) * R)); CK(hipMalloc(&dbg, 512 * 8)); CK(hipMemset(dbg, 0, 512 * 8));
    fill_bf16<<<2048, 256>>>(kc, cache_stride, 3u); fill_bf16<<<2048, 256>>>(vc, cache_stride, 5u);
    fill_f32<<<2048, 256>>>(q, (size_t)R * Hq * D, 7u, 0.35f);
    std::vector<VVRow> hr(R); for (int i = 0; i < R; ++i) { hr[i].cache = 0; hr[i].pos = i; }
    CK(hipMemcpy(rows, hr.data(), sizeof(VVRow) * R, hipMemcpyHostToDevice));
    vv_attn_dbg_set(dbg);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flops = 4.0 * Hq * D * (double)R * R / 2.0;
    const size_t nb = (size_t)R * Hq * D * 4;
    std::vector<float> a(nb / 4), b(nb / 4);
    const int variants[] = {0, 4, 5, 6};
    for (int v : variants) {
        vv_attn_variant_set(v);
        CK(hipMemset(out, 0, nb));
        int rc = vv_attn_prefill3_launch(D, q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, v == 0 ? ref : out, 0);
        CK(hipDeviceSynchronize());
        if (rc) { printf("rc %d\n", rc); return 1; }
        long long nd = 0, nan = 0; double worst = 0;
        if (v != 0) {
            CK(hipMemcpy(a.data(), out, nb, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ref, nb, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nb / 4; ++i) { if (memcmp(&a[i], &b[i], 4)) { ++nd; double d = a[i] - b[i]; if (d < 0) d = -d; if (d > worst) worst = d; } if (a[i] != a[i]) ++nan; }
        }
        for (int i = 0; i < 2; ++i) vv_attn_prefill3_launch(D, q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, out, 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) vv_attn_prefill3_launch(D, q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, out, 0);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("variant %d: %.1f us per launch, %.1f TFLOP/s, differing words vs variant 0: %lld (max |d| %.3g, NaNs %lld)\n", v, ms * 1000 / reps, flops / (ms / reps * 1e-3) / 1e12, nd, worst, nan);
    }
    // phase stamps of the longest workgroup (variants 11 / 12 = timing builds of 1 / 2)
    for (int v : {14, 15}) {
        vv_attn_variant_set(v);
        CK(hipMemset(dbg, 0, 512 * 8));
        vv_attn_prefill3_launch(D, q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, out, 0);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(512);
        CK(hipMemcpy(h.data(), dbg, 512 * 8, hipMemcpyDeviceToHost));
        const char* n0[6] = {"softmax(+V req)", "dma issue", "waits", "barrierA", "matrix+lgkm", "barrierB->next"};
        const char* n1[6] = {"matrix+waits", "barrierA", "softmax(+V req)", "dma issue", "lgkm wait", "barrierB->next"};
        for (int hf = 0; hf < 2; ++hf) {
            double sum[6] = {0}; int cnt = 0;
            for (int s = 4; s < 40; ++s) { for (int k = 0; k < 6; ++k) sum[k] += (double)(h[hf * 256 + s * 6 + k + 1] - h[hf * 256 + s * 6 + k]); ++cnt; }
            printf("variant %d half %d (avg ticks over %d stages):", v, hf, cnt);
            double tot = 0;
            for (int k = 0; k < 6; ++k) { printf("  %s %.0f", hf ? n1[k] : n0[k], sum[k] / cnt); tot += sum[k] / cnt; }
            printf("  | stage %.0f\n", tot);
        }
    }
    return 0;
}
