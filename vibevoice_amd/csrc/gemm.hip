// gemm.hip -- the weight-streaming skinny GEMM that carries every dense
// contraction of the hot path (LM q/k/v/o/gate/up/down, diffusion-head adaLN +
// SwiGLU, codec pointwise FFNs, causal convs and transposed convs recast as
// GEMMs over time-major buffers, connectors, lm_head restricted rows).
//
//   Y[t][n] (op)= sum_k f(X[t][k]) * W[n][k]      T <= a few thousand, any N, K
//
// MI355X mapping
//  * W is pre-packed at load time into 1 KiB MFMA-fragment tiles (vv_common.h),
//    so every wave-level weight load is one fully coalesced 1 KiB
//    global_load_dwordx4 and goes straight to VGPRs (no LDS round trip: the
//    weights are used once per wave -- HBM-bound, arithmetic intensity ~1).
//  * one wave = NT tiles of 16 output features x 16 activation rows, accumulated
//    with v_mfma_f32_16x16x32_bf16 (A = weight tile, B = activations), fp32 acc.
//  * activations stay fp32 in HBM/L2; they are converted on the fly into 1..3
//    bf16 terms (hi [+ mid [+ lo]]) so the product is exact to bf16 / ~fp24 /
//    fp32 activation precision at 1..3 MFMAs per tile (XS template parameter).
//    The matrix pipe is <10 % busy either way; HBM is the bound.
//  * RMSNorm / adaLN-modulate / SiLU prologues and bias / GELU / SwiGLU /
//    layer-scale-residual / gated-residual epilogues are fused so each
//    activation vector makes one trip.
//  * K can be split across the 4 waves of a block (ksplit) and reduced through
//    LDS in a fixed order -> deterministic, no atomics.
#include "vv_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + expf(-u)); }
__device__ __forceinline__ float gelu_erf_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

template <int XS>
__device__ __forceinline__ void split_bf16(const float (&v)[8], bf16x8 (&out)[XS]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 h = (__bf16)v[j];
        out[0][j] = h;
        if constexpr (XS > 1) {
            float r = v[j] - (float)h;
            __bf16 m = (__bf16)r;
            out[1][j] = m;
            if constexpr (XS > 2) {
                float r2 = r - (float)m;
                out[2][j] = (__bf16)r2;
            }
        }
    }
}

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 u) { return __builtin_bit_cast(bf16x8, u); }

template <int NT, int XS, bool DUAL>
__global__ __launch_bounds__(256) void vv_gemm_kernel(const VVGemm a) {
    constexpr int NM = DUAL ? 2 * NT : NT;
    constexpr int U = (NM >= 2) ? 4 : 8;          // k-steps per batch: ~8 KiB of weights in flight per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int KS = a.ksplit;
    const int NG = 4 / KS;
    const int ng = wave / KS, ks = wave - ng * KS;
    const int n_tiles = (a.N + 15) >> 4;
    const int k_tiles = (a.K + 31) >> 5;
    const int tile0 = ((int)blockIdx.x * NG + ng) * NT;
    const int t0 = (int)blockIdx.y * 16;
    const int kper = (k_tiles + KS - 1) / KS;
    const int kt_begin = ks * kper;
    const int kt_end = min(k_tiles, kt_begin + kper);
    const int row = t0 + (lane & 15);
    const bool row_ok = row < a.T;
    const int kq = (lane >> 4) * 8;
    const bool active = tile0 < n_tiles;
    const bool vec_ok = ((a.K & 3) == 0) && ((a.ldx & 3) == 0) && ((((uintptr_t)a.X) & 15) == 0);

    // ---- prologue: per-row 1/rms over the full K (every wave, redundantly: x is L2-resident) ----
    float rstd = 1.0f;
    if (a.pro == VV_PRO_RMS || a.pro == VV_PRO_RMS_MOD) {
        const int nrows = min(16, a.T - t0);
        for (int r = 0; r < nrows; ++r) {
            const float* xr = a.X + (int64_t)(t0 + r) * a.ldx;
            float s = 0.f;
            if (vec_ok) {
                for (int k = lane * 4; k < a.K; k += 256) {
                    float4 v = *reinterpret_cast<const float4*>(xr + k);
                    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
            } else {
                for (int k = lane; k < a.K; k += 64) { float v = xr[k]; s += v * v; }
            }
            s = wave_sum(s);
            float rs = rsqrtf(s / (float)a.K + a.eps);
            if ((lane & 15) == r) rstd = rs;
        }
    }

    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* xrow = a.X + (int64_t)(row_ok ? row : 0) * a.ldx;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

    if (active) {
        for (int ktb = kt_begin; ktb < kt_end; ktb += U) {
            u32x4 wv[U][NM];
            float xv[U][8];
            // -- issue all weight loads of this batch (coalesced 1 KiB per wave-instruction) --
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int kt = ktb + u;
                const bool kok = kt < kt_end;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    const int tile = tile0 + i;
                    const bool ok = kok && tile < n_tiles;
                    const int64_t off = ((int64_t)tile * k_tiles + kt) * 64 + lane;
                    if (a.nt) {
                        wv[u][i] = ok ? __builtin_nontemporal_load(a.W + off) : zero4;
                        if constexpr (DUAL) wv[u][NT + i] = ok ? __builtin_nontemporal_load(a.W2 + off) : zero4;
                    } else {
                        wv[u][i] = ok ? a.W[off] : zero4;
                        if constexpr (DUAL) wv[u][NT + i] = ok ? a.W2[off] : zero4;
                    }
                }
            }
            // -- activations for this batch (fp32, L2-resident) --
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k0 = (ktb + u) * 32 + kq;
                const bool kok = (ktb + u) < kt_end;
                if (row_ok && kok && vec_ok && k0 + 8 <= a.K) {
                    float4 lo = *reinterpret_cast<const float4*>(xrow + k0);
                    float4 hi = *reinterpret_cast<const float4*>(xrow + k0 + 4);
                    xv[u][0] = lo.x; xv[u][1] = lo.y; xv[u][2] = lo.z; xv[u][3] = lo.w;
                    xv[u][4] = hi.x; xv[u][5] = hi.y; xv[u][6] = hi.z; xv[u][7] = hi.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        xv[u][j] = (row_ok && kok && (k0 + j) < a.K) ? xrow[k0 + j] : 0.f;
                }
            }
            // -- prologue math, bf16 split, MFMA --
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k0 = (ktb + u) * 32 + kq;
                if ((ktb + u) < kt_end) {
                    if (a.pro != VV_PRO_NONE) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int k = k0 + j;
                            const bool ok = row_ok && k < a.K;
                            float v = xv[u][j];
                            if (a.pro == VV_PRO_ADD_SILU) {
                                v = ok ? silu_f(v + a.addvec[k]) : 0.f;
                            } else {
                                v = v * rstd;
                                if (a.nw) v = ok ? v * a.nw[k] : 0.f;
                                if (a.pro == VV_PRO_RMS_MOD && ok) {
                                    const int64_t mo = (int64_t)row * a.ld_mod + k;
                                    v = v * (1.0f + a.mod_scale[mo]) + a.mod_shift[mo];
                                }
                            }
                            xv[u][j] = v;
                        }
                    }
                    bf16x8 xb[XS];
                    split_bf16<XS>(xv[u], xb);
#pragma unroll
                    for (int i = 0; i < NM; ++i) {
                        const bf16x8 wf = as_bf16x8(wv[u][i]);
#pragma unroll
                        for (int p = 0; p < XS; ++p)
                            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb[p], acc[i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- split-K reduction through LDS, fixed order (deterministic) ----
    __shared__ f32x4 red[4][NM][64];
    if (KS > 1) {
#pragma unroll
        for (int i = 0; i < NM; ++i) red[wave][i][lane] = acc[i];
        __syncthreads();
        if (ks == 0) {
            for (int s = 1; s < KS; ++s) {
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i] += red[wave + s][i][lane];
            }
        }
    }
    if (ks != 0 || !active) return;

    // ---- epilogue: lane holds D[n = tile*16 + (lane>>4)*4 + r][t = lane&15] ----
    if (!row_ok) return;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n0 = (tile0 + i) * 16 + (lane >> 4) * 4;
        if (n0 >= a.N) continue;
        float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        float* yp = a.Y + (int64_t)row * a.ldy + n0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + r;
            if (n >= a.N) break;
            float o = v[r];
            switch (a.epi) {
                case VV_EPI_STORE: break;
                case VV_EPI_BIAS: if (a.bias) o += a.bias[n]; break;
                case VV_EPI_BIAS_GELU: if (a.bias) o += a.bias[n]; o = gelu_erf_f(o); break;
                case VV_EPI_SWIGLU:
                    if constexpr (DUAL) { o = silu_f(o) * acc[NT + i][r]; }
                    break;
                case VV_EPI_RESID: {
                    if (a.bias) o += a.bias[n];
                    if (a.nscale) o *= a.nscale[n];
                    o += yp[r];
                } break;
                case VV_EPI_GATED_RESID: {
                    o = yp[r] + a.gate[(int64_t)row * a.ld_gate + n] * o;
                } break;
            }
            yp[r] = o;
        }
    }
}

// ---- weight packing ----------------------------------------------------------
// kind 0: Linear   src[n][k]                     (row-major [N][K])
// kind 1: Conv1d   src[o][c][j]  -> k = j*Cin + c            (time-major window)
// kind 2: ConvT1d  src[c][o][j]  -> n = jj*Cout + o, k = tap*Cin + c,
//                  tap 0 = previous input frame (kernel index jj + stride),
//                  tap 1 = current input frame  (kernel index jj)
template <typename ST>
__global__ void vv_pack_kernel(const ST* __restrict__ src, __bf16* __restrict__ dst,
                               int N, int K, int kind, int Cin, int Cout, int ksz, int stride) {
    const int64_t total = vv_packed_elems(N, K);
    const int k_tiles = (K + 31) >> 5;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(e & 7);
        const int lane = (int)((e >> 3) & 63);
        const int64_t tile = e >> 9;
        const int kt = (int)(tile % k_tiles);
        const int ntile = (int)(tile / k_tiles);
        const int n = ntile * 16 + (lane & 15);
        const int k = kt * 32 + (lane >> 4) * 8 + j;
        float v = 0.f;
        if (n < N && k < K) {
            int64_t si;
            if (kind == 0) {
                si = (int64_t)n * K + k;
            } else if (kind == 1) {
                const int jj = k / Cin, c = k - jj * Cin;
                si = ((int64_t)n * Cin + c) * ksz + jj;
            } else {
                const int jj = n / Cout, o = n - jj * Cout;
                const int tap = k / Cin, c = k - tap * Cin;
                si = ((int64_t)c * Cout + o) * ksz + (tap == 0 ? jj + stride : jj);
            }
            v = (float)src[si];
        }
        dst[e] = (__bf16)v;
    }
}

}  // namespace

// ---- host launchers ------------------------------------------------------------
template <int NT, int XS, bool DUAL>
static void launch_t(const VVGemm& a, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((vv_gemm_kernel<NT, XS, DUAL>), grid, dim3(256), 0, s, a);
}

// Chooses the decomposition.  `xs` in {1,2,3}.
extern "C" int vv_gemm_launch(VVGemm a, int xs, hipStream_t s) {
    const int n_tiles = (a.N + 15) / 16;
    const int k_tiles = (a.K + 31) / 32;
    const int t_tiles = (a.T + 15) / 16;
    const bool dual = a.epi == VV_EPI_SWIGLU;
    if (dual && !a.W2) return -1;
    // K split: enough k-steps per wave to amortise, and enough waves to fill 256 CUs.
    int ks = 1;
    if (a.ksplit > 0) ks = a.ksplit;
    else {
        const long waves1 = (long)n_tiles * t_tiles;
        if (k_tiles >= 64 && waves1 < 4096) ks = 4;
        else if (k_tiles >= 16 && waves1 < 2048) ks = (waves1 < 1024 && k_tiles >= 32) ? 4 : 2;
    }
    a.ksplit = ks;
    const int ng = 4 / ks;
    // two tiles per wave once there are plenty of tiles (halves the x-fragment work per byte)
    int nt = 1;
    if (!dual && (long)n_tiles * t_tiles >= 8192) nt = 2;
    const int per_block = ng * nt;
    dim3 grid((n_tiles + per_block - 1) / per_block, t_tiles);
#define VV_GO(NT_, DUAL_)                                                         \
    do {                                                                          \
        if (xs == 1) launch_t<NT_, 1, DUAL_>(a, grid, s);                         \
        else if (xs == 2) launch_t<NT_, 2, DUAL_>(a, grid, s);                    \
        else launch_t<NT_, 3, DUAL_>(a, grid, s);                                 \
    } while (0)
    if (dual) VV_GO(1, true);
    else if (nt == 2) VV_GO(2, false);
    else VV_GO(1, false);
#undef VV_GO
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int vv_pack_launch(const void* src, int src_is_bf16, void* dst, int N, int K, int kind,
                              int Cin, int Cout, int ksz, int stride, hipStream_t s) {
    const int64_t total = vv_packed_elems(N, K);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (src_is_bf16)
        hipLaunchKernelGGL((vv_pack_kernel<__bf16>), dim3(blocks), dim3(256), 0, s, (const __bf16*)src,
                           (__bf16*)dst, N, K, kind, Cin, Cout, ksz, stride);
    else
        hipLaunchKernelGGL((vv_pack_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float*)src,
                           (__bf16*)dst, N, K, kind, Cin, Cout, ksz, stride);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
