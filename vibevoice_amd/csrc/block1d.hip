// block1d.hip -- one fused kernel per tokenizer Block1D for the long, thin stages
// (C = 32/64/128 channels, T = 800..3200+ time steps per frame):
//
//   x1  = x  + gamma     * (dwconv7_causal(RMSNorm(x ; w_n)) + b_dw)
//   out = x1 + gamma_ffn * (W2 . GELU(W1 . RMSNorm(x1 ; w_f) + b1) + b2)
//
// (modular_vibevoice_tokenizer.py:924-942 / :786-804).  The unfused path needs four
// launches and round-trips a [T][4C] fp32 hidden tensor through HBM per block; here a
// workgroup owns 16 time steps, keeps x / norms / FFN hidden in LDS, feeds both FFN
// GEMMs to the MFMA pipe straight from LDS-resident B fragments, and reads the
// (<=256 KiB, L2-resident) packed weights as ready-made A fragments.  Input and output
// are different buffers (the stage ping-pongs), so halo rows can be re-normalised from
// the input without racing the neighbours' writes.  Streaming state = the last 6
// *normed* rows, written to `nst + 6*C` and moved to the front by the net's shift kernel.
#include "vv_common.h"

namespace {

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

struct VVBlock {
    const float* xin;      // live rows [T][C] of the input buffer
    float* xout;           // live rows [T][C] of the output buffer
    float* nst;            // [12][C]: rows 0..5 normed history (previous frame), rows 6..11 next state
    const float *norm_w, *ffn_norm_w, *gamma, *ffn_gamma, *dw_w /*[7][C]*/, *dw_b, *b1, *b2;
    const u32x4 *w1, *w2;  // packed [4C][C], [C][4C]
    int T;
    float eps;
    // slot-batched launch (sl.n > 0): blockIdx.y picks the utterance; its buffers sit sx (xin / xout) and snst floats further per slot id
    VVSlotIds sl;
    int64_t sx, snst;
};

template <int XS>
__device__ __forceinline__ void split_store(unsigned char* base, size_t part_stride, int byte_off, float v) {
    __bf16 h = (__bf16)v;
    *reinterpret_cast<__bf16*>(base + byte_off) = h;
    if constexpr (XS > 1) {
        float r = v - (float)h;
        __bf16 m = (__bf16)r;
        *reinterpret_cast<__bf16*>(base + part_stride + byte_off) = m;
        if constexpr (XS > 2) *reinterpret_cast<__bf16*>(base + 2 * part_stride + byte_off) = (__bf16)(r - (float)m);
    }
}

template <int C, int XS>
__global__ __launch_bounds__(256) void vv_block1d_kernel(const VVBlock a) {
    constexpr int TT = 16, HALO = 6, F = 4 * C;
    constexpr int KT1 = C / 32, NT1 = F / 16;      // FFN1: K = C, N = 4C
    constexpr int KT2 = F / 32, NT2 = C / 16;      // FFN2: K = 4C, N = C
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* nrm = reinterpret_cast<float*>(smem);                 // [TT+HALO][C]
    float* xs = nrm + (TT + HALO) * C;                           // [TT][C]  raw x, then x1
    float* rs2 = xs + TT * C;                                    // [TT]
    unsigned char* f1 = reinterpret_cast<unsigned char*>(rs2 + TT);          // XS x KT1 KiB   B frags of n2
    unsigned char* f2 = f1 + (size_t)XS * KT1 * 1024;                        // XS x KT2 KiB   B frags of u
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t0 = blockIdx.x * TT;
    const float* xin = a.xin; float* xout = a.xout; float* nst = a.nst;
    if (a.sl.n > 0) {
        const int64_t id = vv_slot_id(a.sl.id, blockIdx.y);
        xin += id * a.sx; xout += id * a.sx; nst += id * a.snst;
    }
    // The packed weights do not depend on x: every fragment this wave will feed to the MFMA pipe is requested NOW
    // (FFN1: NT1/4 tiles x KT1, FFN2: NT2/4 tiles x KT2 -- at most 32 + 32 fragments for C = 128), so the L2 round
    // trips overlap the norm / conv phases instead of forming a chain of dependent loads inside the GEMM loops.
    constexpr int M1 = (NT1 + 3) / 4, M2 = (NT2 + 3) / 4;
    u32x4 w1r[M1][KT1], w2r[M2][KT2];
#pragma unroll
    for (int i = 0; i < M1; ++i) {
        const int nt = wave + 4 * i;
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt)
            w1r[i][kt] = (nt < NT1) ? a.w1[((int64_t)nt * KT1 + kt) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < M2; ++i) {
        const int nt = wave + 4 * i;
#pragma unroll
        for (int kt = 0; kt < KT2; ++kt)
            w2r[i][kt] = (nt < NT2) ? a.w2[((int64_t)nt * KT2 + kt) * 64 + lane] : u32x4{0u, 0u, 0u, 0u};
    }

    // per-channel parameters, requested up front as well (C divides 256: a thread's conv channel is fixed)
    constexpr int CL = (C + 63) / 64;
    float nw_r[CL], fnw_r[CL];
#pragma unroll
    for (int i = 0; i < CL; ++i) {
        const int c = lane + i * 64;
        nw_r[i] = (c < C) ? a.norm_w[c] : 0.f;
        fnw_r[i] = (c < C) ? a.ffn_norm_w[c] : 0.f;
    }
    const int cc = tid % C;
    float dw_r[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) dw_r[j] = a.dw_w[j * C + cc];
    const float dwb_r = a.dw_b[cc], gam_r = a.gamma[cc];
    const int frow = lane & 15, fq = lane >> 4;
    float4 b1_r[M1], b2_r[M2], fg_r[M2];
#pragma unroll
    for (int i = 0; i < M1; ++i) {
        const int nt = wave + 4 * i;
        b1_r[i] = (nt < NT1) ? *reinterpret_cast<const float4*>(a.b1 + nt * 16 + fq * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < M2; ++i) {
        const int nt = wave + 4 * i;
        b2_r[i] = (nt < NT2) ? *reinterpret_cast<const float4*>(a.b2 + nt * 16 + fq * 4) : float4{0.f, 0.f, 0.f, 0.f};
        fg_r[i] = (nt < NT2) ? *reinterpret_cast<const float4*>(a.ffn_gamma + nt * 16 + fq * 4) : float4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- A: RMSNorm of rows t0-6 .. t0+15 (history rows come from the streaming state); all row loads first ----
    constexpr int RW = (TT + HALO + 3) / 4;
    float xv[RW][CL];
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int rr = wave + 4 * i;
        const int t = t0 - HALO + rr;
#pragma unroll
        for (int q = 0; q < CL; ++q) {
            const int c = lane + q * 64;
            float v = 0.f;
            if (rr < TT + HALO && c < C) {
                if (t < 0) v = nst[(HALO + t) * C + c];
                else if (t < a.T) v = xin[(int64_t)t * C + c];
            }
            xv[i][q] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < RW; ++i) {
        const int rr = wave + 4 * i;
        if (rr >= TT + HALO) break;
        const int t = t0 - HALO + rr;
        if (t < 0) {
#pragma unroll
            for (int q = 0; q < CL; ++q) { const int c = lane + q * 64; if (c < C) nrm[rr * C + c] = xv[i][q]; }
        } else if (t < a.T) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < CL; ++q) s += xv[i][q] * xv[i][q];
            const float r = rsqrtf(wsum(s) / (float)C + a.eps);
#pragma unroll
            for (int q = 0; q < CL; ++q) {
                const int c = lane + q * 64;
                if (c < C) {
                    const float n = xv[i][q] * r * nw_r[q];
                    nrm[rr * C + c] = n;
                    if (rr >= HALO) xs[(rr - HALO) * C + c] = xv[i][q];
                    if (t >= a.T - HALO) nst[(HALO + t - (a.T - HALO)) * C + c] = n;     // next frame's history
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < CL; ++q) {
                const int c = lane + q * 64;
                if (c < C) { nrm[rr * C + c] = 0.f; if (rr >= HALO) xs[(rr - HALO) * C + c] = 0.f; }
            }
        }
    }
    __syncthreads();
    // ---- B: causal depthwise conv k=7 + bias, layer scale, residual -> x1 (in xs) ----
    for (int e = tid; e < TT * C; e += 256) {
        const int r = e / C;                           // channel of e is cc
        float acc = dwb_r;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += dw_r[j] * nrm[(r + j) * C + cc];
        xs[e] += gam_r * acc;
    }
    __syncthreads();
    // ---- C: second RMSNorm -> bf16 B fragments of n2 ----
    for (int r = wave; r < TT; r += 4) {
        float s = 0.f;
        float xr[CL];
#pragma unroll
        for (int q = 0; q < CL; ++q) { const int c = lane + q * 64; xr[q] = (c < C) ? xs[r * C + c] : 0.f; s += xr[q] * xr[q]; }
        const float rr = rsqrtf(wsum(s) / (float)C + a.eps);
#pragma unroll
        for (int q = 0; q < CL; ++q) {
            const int c = lane + q * 64;
            if (c < C) {
                const float n = xr[q] * rr * fnw_r[q];
                const int off = (((c >> 5) * 64 + r + 16 * ((c & 31) >> 3)) * 8 + (c & 7)) * 2;
                split_store<XS>(f1, (size_t)KT1 * 1024, off, n);
            }
        }
    }
    __syncthreads();
    // ---- D: FFN1 + bias + exact GELU -> bf16 B fragments of u ----
#pragma unroll
    for (int i1 = 0; i1 < M1; ++i1) {
        const int nt = wave + 4 * i1;
        if (nt >= NT1) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT1; ++kt) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w1r[i1][kt]);
#pragma unroll
            for (int p = 0; p < XS; ++p) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(f1 + ((size_t)(p * KT1 + kt) * 64 + lane) * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb, acc, 0, 0, 0);
            }
        }
        const int n0 = nt * 16 + fq * 4;           // lane holds u[t=frow][n0..n0+3]
        float u[4];
        const float bb[4] = {b1_r[i1].x, b1_r[i1].y, b1_r[i1].z, b1_r[i1].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) u[r] = gelu_erf(acc[r] + bb[r]);
        const int off = (((n0 >> 5) * 64 + frow + 16 * ((n0 & 31) >> 3)) * 8 + (n0 & 7)) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) split_store<XS>(f2, (size_t)KT2 * 1024, off + r * 2, u[r]);
    }
    __syncthreads();
    // ---- E: FFN2 + bias, layer scale, residual -> out ----
#pragma unroll
    for (int i2 = 0; i2 < M2; ++i2) {
        const int nt = wave + 4 * i2;
        if (nt >= NT2) break;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < KT2; ++kt) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w2r[i2][kt]);
#pragma unroll
            for (int p = 0; p < XS; ++p) {
                const bf16x8 ub = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(f2 + ((size_t)(p * KT2 + kt) * 64 + lane) * 16));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, ub, acc, 0, 0, 0);
            }
        }
        const int c0 = nt * 16 + fq * 4;
        const int t = t0 + frow;
        if (t < a.T) {
            float4 o;
            o.x = xs[frow * C + c0 + 0] + fg_r[i2].x * (acc[0] + b2_r[i2].x);
            o.y = xs[frow * C + c0 + 1] + fg_r[i2].y * (acc[1] + b2_r[i2].y);
            o.z = xs[frow * C + c0 + 2] + fg_r[i2].z * (acc[2] + b2_r[i2].z);
            o.w = xs[frow * C + c0 + 3] + fg_r[i2].w * (acc[3] + b2_r[i2].w);
            *reinterpret_cast<float4*>(xout + (int64_t)t * C + c0) = o;
        }
    }
}

template <int C, int XS>
static void go(const VVBlock& a, hipStream_t s) {
    constexpr int TT = 16, HALO = 6, F = 4 * C;
    const size_t smem = (size_t)((TT + HALO) * C + TT * C + TT) * 4 + (size_t)XS * (C / 32) * 1024 + (size_t)XS * (F / 32) * 1024;
    static bool attr = false;
    if (!attr) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_block1d_kernel<C, XS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((vv_block1d_kernel<C, XS>), dim3((a.T + TT - 1) / TT, a.sl.n > 0 ? a.sl.n : 1), dim3(256), smem, s, a);
}

}  // namespace

extern "C" int vv_block1d_supported(int C) { return C == 32 || C == 64 || C == 128; }
extern "C" int vv_block1d_slots_launch(int C, int xs, const float* xin, float* xout, float* nst, const float* norm_w,
                                       const float* ffn_norm_w, const float* gamma, const float* ffn_gamma,
                                       const float* dw_w, const float* dw_b, const float* b1, const float* b2,
                                       const void* w1, const void* w2, int T, float eps, const int* ids, int n, int64_t sx,
                                       int64_t snst, hipStream_t s);

extern "C" int vv_block1d_launch(int C, int xs, const float* xin, float* xout, float* nst, const float* norm_w,
                                 const float* ffn_norm_w, const float* gamma, const float* ffn_gamma,
                                 const float* dw_w, const float* dw_b, const float* b1, const float* b2,
                                 const void* w1, const void* w2, int T, float eps, hipStream_t s) {
    return vv_block1d_slots_launch(C, xs, xin, xout, nst, norm_w, ffn_norm_w, gamma, ffn_gamma, dw_w, dw_b, b1, b2, w1, w2, T, eps,
                                   nullptr, 0, 0, 0, s);
}

// the same block over n utterance slots (ids != null): xin / xout / nst are slot 0's buffers
extern "C" int vv_block1d_slots_launch(int C, int xs, const float* xin, float* xout, float* nst, const float* norm_w,
                                       const float* ffn_norm_w, const float* gamma, const float* ffn_gamma,
                                       const float* dw_w, const float* dw_b, const float* b1, const float* b2,
                                       const void* w1, const void* w2, int T, float eps, const int* ids, int n, int64_t sx,
                                       int64_t snst, hipStream_t s) {
    if (n < 0 || n > 8) return -1;
    VVBlock a{xin, xout, nst, norm_w, ffn_norm_w, gamma, ffn_gamma, dw_w, dw_b, b1, b2,
              (const u32x4*)w1, (const u32x4*)w2, T, eps};
    a.sl.n = ids ? n : 0;
    for (int i = 0; i < 8; ++i) a.sl.id[i] = (ids && i < n) ? ids[i] : 0;
    a.sx = sx; a.snst = snst;
#define VV_B(C_)                                                     \
    do {                                                             \
        if (xs == 1) go<C_, 1>(a, s);                                \
        else if (xs == 2) go<C_, 2>(a, s);                           \
        else go<C_, 3>(a, s);                                        \
    } while (0)
    if (C == 32) VV_B(32);
    else if (C == 64) VV_B(64);
    else if (C == 128) VV_B(128);
    else return -1;
#undef VV_B
    return vv_launch_rc(0);
}
