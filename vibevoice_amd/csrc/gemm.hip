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
//    global_load_dwordx4 (non-temporal for streamed-once weights) straight into
//    VGPRs -- no LDS round trip for weights: each is used once per wave, the
//    kernel is HBM-bound (arithmetic intensity ~1 flop/B).  The next batch of
//    weight tiles is prefetched while the current one feeds the matrix pipe.
//  * one wave = NT tiles of 16 output features x 16 activation rows, accumulated
//    with v_mfma_f32_16x16x32_bf16 (A = weight tile, B = activations), fp32 acc.
//  * activations stay fp32 in HBM/L2.  Per batch of U k-steps a wave reads its
//    rows with coalesced float4 loads, applies the prologue (norm weight, adaLN
//    modulate, add+SiLU), splits each value into 1..3 bf16 terms (hi [+mid [+lo]],
//    XS) and parks the terms in a wave-private LDS tile laid out in B-fragment
//    order; fragments come back with one ds_read_b128 per k-step.  Only the rows
//    that exist are converted (T=2 at decode), and LDS traffic is on lgkmcnt, so
//    it never serialises behind the weight stream's vmcnt.
//  * RMSNorm costs no extra pass: sum(x^2) is accumulated during staging and the
//    1/rms factor is applied to the accumulator in the epilogue (it is a per-row
//    scalar); only the adaLN-modulated norm (shift term) needs 1/rms up front.
//  * K is split across KS of the block's waves (up to 16) and reduced through LDS
//    in a fixed order -> deterministic, no atomics, enough waves for skinny N.
#include <cstdlib>
#include "vv_common.h"

#ifdef VV_GEMM_TIMING
#define VV_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define VV_BSTAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 3200) a.dbg[16 + 2 * (blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VV_STAMP(i) do { } while (0)
#define VV_BSTAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + expf(-u)); }
__device__ __forceinline__ float gelu_erf_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

__device__ __forceinline__ bf16x8 as_bf16x8(u32x4 u) { return __builtin_bit_cast(bf16x8, u); }

// 4 fp32 -> XS packed bf16x4 terms (8 bytes each)
template <int XS>
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&out)[XS]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, m, l;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)v[j];
        if constexpr (XS > 1) {
            r[j] = v[j] - (float)h[j];
            m[j] = (__bf16)r[j];
            if constexpr (XS > 2) l[j] = (__bf16)(r[j] - (float)m[j]);
        }
    }
    out[0] = __builtin_bit_cast(uint2, h);
    if constexpr (XS > 1) out[1] = __builtin_bit_cast(uint2, m);
    if constexpr (XS > 2) out[2] = __builtin_bit_cast(uint2, l);
}

template <int NT, int XS, bool DUAL, int WPB, int MAXR, bool VEC>
__global__ __launch_bounds__(WPB * 64) void vv_gemm_kernel(const VVGemm a) {
    constexpr int NM = DUAL ? 2 * NT : NT;
    constexpr bool PREFETCH = (WPB == 8);         // decode variant: second weight buffer; tall variant favours occupancy
    constexpr bool MODREG = (MAXR <= 4);          // adaLN scale/shift prefetched in registers only for few rows
    constexpr int U = 8;                          // k-steps per batch = 256 k = one float4 per lane per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // one batch of s_loads for every kernel argument (fetched lazily they cost 3-4 dependent round trips per launch)
    asm volatile("" ::"s"(a.W), "s"(a.W2), "s"(a.X), "s"(a.Y), "s"(a.nw), "s"(a.mod_scale), "s"(a.mod_shift),
                 "s"(a.addvec), "s"(a.bias), "s"(a.nscale), "s"(a.gate));
    asm volatile("" ::"s"(a.T), "s"(a.N), "s"(a.K), "s"(a.ldx), "s"(a.ldy), "s"(a.ld_mod), "s"(a.ld_gate), "s"(a.x_row_mod), "s"(a.add_rows_per_vec),
                 "s"(a.eps), "s"(a.pro), "s"(a.epi), "s"(a.ksplit), "s"(a.t_pad));
    VV_STAMP(0);
    VV_BSTAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int KS = a.ksplit;
    const int NG = WPB / KS;
    const int ng = wave / KS, ks = wave - ng * KS;
    const int n_tiles = (a.N + 15) >> 4;
    const int k_tiles = (a.K + 31) >> 5;
    const int tile0 = ((int)blockIdx.x * NG + ng) * NT;
    const int t0 = (int)blockIdx.y * 16;
    const int Tt = min(MAXR, a.T - t0);           // rows of this tile (wave-uniform)
    const int kper = (k_tiles + KS - 1) / KS;     // k-steps per wave (the last batch may be partial)
    const int kt_begin = ks * kper;
    const int kt_end = min(k_tiles, kt_begin + kper);
    const bool active = tile0 < n_tiles && kt_begin < kt_end;
    constexpr bool vec_ok = VEC;                  // host guarantees K%4==0, ldx%4==0, 16-B aligned X for VEC kernels
    const int Tpad = a.t_pad;                     // LDS row stride (>= Tt), host-chosen
    // wave-private staging tile: [XS][U][4 q][Tpad] x 16 B
    unsigned char* stg = smem + (size_t)wave * XS * U * 4 * Tpad * 16;
    const int kk = lane * 4;                      // this lane's k offset inside a batch
    const int st_off = (((kk >> 5) * 4 + ((kk & 31) >> 3)) * Tpad) * 16 + (kk & 7) * 2;
    const int frow = lane & 15, fq = lane >> 4;

    // ---- adaLN-modulated norm needs 1/rms before staging: one cheap pass over the tile's rows ----
    float rstd_rows[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) rstd_rows[r] = 1.0f;
    if (a.pro == VV_PRO_RMS_MOD) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < Tt) {
                const float* xr = a.X + (int64_t)(t0 + r) * a.ldx;
                float s = 0.f;
                if constexpr (VEC) {
                    for (int k = lane * 4; k < a.K; k += 256) {
                        float4 v = *reinterpret_cast<const float4*>(xr + k);
                        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    }
                } else {
                    for (int k = lane; k < a.K; k += 64) { float v = xr[k]; s += v * v; }
                }
                rstd_rows[r] = rsqrtf(wave_sum(s) / (float)a.K + a.eps);
            }
        }
    }

    // ---- epilogue operands (bias / residual / gate rows), requested before the weight stream when everything is
    //      16-byte aligned: one overlapped round trip instead of a chain of dependent scalar loads after the MFMAs ----
    const bool vec_epi = VEC && a.epi != VV_EPI_CFG_DPM && !((a.N | a.ldy) & 3) &&
                         !((reinterpret_cast<uintptr_t>(a.Y) | reinterpret_cast<uintptr_t>(a.bias) | reinterpret_cast<uintptr_t>(a.nscale)) & 15) &&
                         (a.epi != VV_EPI_GATED_RESID || (!(a.ld_gate & 3) && !(reinterpret_cast<uintptr_t>(a.gate) & 15)));
    float4 pre_b[NT], pre_y[NT], pre_g[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        pre_b[i] = float4{0.f, 0.f, 0.f, 0.f}; pre_y[i] = float4{0.f, 0.f, 0.f, 0.f}; pre_g[i] = float4{1.f, 1.f, 1.f, 1.f};
        const int n0 = (tile0 + i) * 16 + fq * 4;
        if (vec_epi && ks == 0 && frow < Tt && n0 < a.N) {
            const int64_t yo = (int64_t)(t0 + frow) * a.ldy + n0;
            if (a.bias && (a.epi == VV_EPI_BIAS || a.epi == VV_EPI_BIAS_GELU || a.epi == VV_EPI_RESID)) pre_b[i] = *reinterpret_cast<const float4*>(a.bias + n0);
            if (a.epi == VV_EPI_RESID || a.epi == VV_EPI_GATED_RESID) pre_y[i] = *reinterpret_cast<const float4*>(a.Y + yo);
            if (a.epi == VV_EPI_GATED_RESID) pre_g[i] = *reinterpret_cast<const float4*>(a.gate + (int64_t)(t0 + frow) * a.ld_gate + n0);
            else if (a.epi == VV_EPI_RESID && a.nscale) pre_g[i] = *reinterpret_cast<const float4*>(a.nscale + n0);
        }
    }

    f32x4 acc[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssq[MAXR];
#pragma unroll
    for (int r = 0; r < MAXR; ++r) ssq[r] = 0.f;
    const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};

    auto load_w = [&](int ktb, u32x4 (&dst)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int kt = ktb + u;
            const bool kok = kt < kt_end;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int tile = tile0 + i;
                const bool ok = kok && tile < n_tiles;
                const int64_t off = ((int64_t)tile * k_tiles + kt) * 64 + lane;
                if constexpr (WPB == 8) {        // decode variant: weights are streamed exactly once
                    dst[u][i] = ok ? __builtin_nontemporal_load(a.W + off) : zero4;
                    if constexpr (DUAL) dst[u][NT + i] = ok ? __builtin_nontemporal_load(a.W2 + off) : zero4;
                } else {
                    dst[u][i] = ok ? a.W[off] : zero4;
                    if constexpr (DUAL) dst[u][NT + i] = ok ? a.W2[off] : zero4;
                }
            }
        }
    };

    // ---- staging, split in two so the loads can be issued ahead of the next weight batch ----
    struct XRegs { float4 x[MAXR]; float4 sc[MODREG ? MAXR : 1]; float4 sh[MODREG ? MAXR : 1]; float4 nwv, addv; };
    auto ld4 = [&](const float* p, int k, bool full) -> float4 {
        float4 v = {0.f, 0.f, 0.f, 0.f};
        if constexpr (VEC) { v = *reinterpret_cast<const float4*>(p + k); (void)full; }
        else {
            if (full) v = *reinterpret_cast<const float4*>(p + k);
            else { v.x = p[k]; if (k + 1 < a.K) v.y = p[k + 1]; if (k + 2 < a.K) v.z = p[k + 2]; if (k + 3 < a.K) v.w = p[k + 3]; }
        }
        return v;
    };
    auto stage_load = [&](int ktb, XRegs& R) {
        const int k = ktb * 32 + kk;
        const bool kin = k < a.K && (ktb + (kk >> 5)) < kt_end;
        const bool full = vec_ok && (k + 4 <= a.K);
        R.nwv = float4{1.f, 1.f, 1.f, 1.f};
        R.addv = float4{0.f, 0.f, 0.f, 0.f};
        if (kin && a.nw) R.nwv = ld4(a.nw, k, full);
        if (kin && a.pro == VV_PRO_ADD_SILU && a.add_rows_per_vec <= 0) R.addv = ld4(a.addvec, k, full);
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            R.x[r] = float4{0.f, 0.f, 0.f, 0.f};
            if (r < Tt && kin) {
                const int xr_idx = a.x_row_mod > 0 ? (t0 + r) % a.x_row_mod : (t0 + r);
                R.x[r] = ld4(a.X + (int64_t)xr_idx * a.ldx, k, full);
                if constexpr (MODREG) {
                    if (a.pro == VV_PRO_RMS_MOD) {
                        const int64_t mo = (int64_t)(t0 + r) * a.ld_mod;
                        const bool mfull = full;
                        R.sc[r] = ld4(a.mod_scale + mo, k, mfull);
                        R.sh[r] = ld4(a.mod_shift + mo, k, mfull);
                    }
                }
            }
        }
    };
    auto stage_finish = [&](int ktb, const XRegs& R) {
        const int k = ktb * 32 + kk;
        const bool kin = k < a.K && (ktb + (kk >> 5)) < kt_end;
        const bool k1 = kin && k + 1 < a.K, k2 = kin && k + 2 < a.K, k3 = kin && k + 3 < a.K;
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < Tt) {
                float v[4] = {R.x[r].x, R.x[r].y, R.x[r].z, R.x[r].w};
                if (a.pro == VV_PRO_RMS) {
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] *= R.nwv.x; v[1] *= R.nwv.y; v[2] *= R.nwv.z; v[3] *= R.nwv.w;
                } else if (a.pro == VV_PRO_RMS_MOD) {
                    const float rs = rstd_rows[r];
                    float4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MODREG) { sc = R.sc[r]; sh = R.sh[r]; }
                    else if (kin) {
                        const int64_t mo = (int64_t)(t0 + r) * a.ld_mod;
                        const bool mfull = vec_ok && (k + 4 <= a.K);
                        sc = ld4(a.mod_scale + mo, k, mfull);
                        sh = ld4(a.mod_shift + mo, k, mfull);
                    }
                    v[0] = kin ? (v[0] * rs * R.nwv.x) * (1.f + sc.x) + sh.x : 0.f;
                    v[1] = k1 ? (v[1] * rs * R.nwv.y) * (1.f + sc.y) + sh.y : 0.f;
                    v[2] = k2 ? (v[2] * rs * R.nwv.z) * (1.f + sc.z) + sh.z : 0.f;
                    v[3] = k3 ? (v[3] * rs * R.nwv.w) * (1.f + sc.w) + sh.w : 0.f;
                } else if (a.pro == VV_PRO_ADD_SILU) {
                    float4 av = R.addv;
                    if (a.add_rows_per_vec > 0 && kin) {
                        const int64_t ao = (int64_t)((t0 + r) / a.add_rows_per_vec) * a.K + k;
                        av.x = a.addvec[ao];
                        av.y = k1 ? a.addvec[ao + 1] : 0.f;
                        av.z = k2 ? a.addvec[ao + 2] : 0.f;
                        av.w = k3 ? a.addvec[ao + 3] : 0.f;
                    }
                    v[0] = kin ? silu_f(v[0] + av.x) : 0.f;
                    v[1] = k1 ? silu_f(v[1] + av.y) : 0.f;
                    v[2] = k2 ? silu_f(v[2] + av.z) : 0.f;
                    v[3] = k3 ? silu_f(v[3] + av.w) : 0.f;
                }
                uint2 parts[XS];
                split4<XS>(v, parts);
#pragma unroll
                for (int p = 0; p < XS; ++p)
                    *reinterpret_cast<uint2*>(stg + (size_t)p * U * 4 * Tpad * 16 + st_off + r * 16) = parts[p];
            }
        }
    };

    auto mma_batch = [&](int ktb, const u32x4 (&wb)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ktb + u < kt_end) {
                bf16x8 xb[XS];
#pragma unroll
                for (int p = 0; p < XS; ++p) {
                    u32x4 f = zero4;
                    if (frow < Tt) f = *reinterpret_cast<const u32x4*>(stg + ((size_t)((p * U + u) * 4 + fq) * Tpad + frow) * 16);
                    xb[p] = as_bf16x8(f);
                }
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    const bf16x8 wf = as_bf16x8(wb[u][i]);
#pragma unroll
                    for (int p = 0; p < XS; ++p)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xb[p], acc[i], 0, 0, 0);
                }
            }
        }
    };

    if (active) {
        XRegs R;
        u32x4 wcur[U][NM];
        if constexpr (PREFETCH) {
            // software pipeline with one call site per stage: iteration b issues the loads of batch b+1,
            // runs the MFMAs of batch b, then converts batch b+1 into LDS (x loads were issued before the
            // weights, so that wait leaves the weight prefetch in flight)
            const int nb = (kt_end - kt_begin + U - 1) / U;
            u32x4 wnext[U][NM];
#pragma unroll 1
            for (int b = -1; b < nb; ++b) {
                const int ktn = kt_begin + (b + 1) * U;
                const bool have_next = b + 1 < nb;
                if (have_next) { stage_load(ktn, R); load_w(ktn, wnext); }
                if (b == -1) VV_STAMP(1);
                if (b >= 0) mma_batch(kt_begin + b * U, wcur);
                if (have_next) {
                    stage_finish(ktn, R);
                    if (b == -1) VV_STAMP(2);
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int i = 0; i < NM; ++i) wcur[u][i] = wnext[u][i];
                }
            }
        } else {
#pragma unroll 1
            for (int ktb = kt_begin; ktb < kt_end; ktb += U) {
                stage_load(ktb, R);
                load_w(ktb, wcur);
                stage_finish(ktb, R);
                mma_batch(ktb, wcur);
            }
        }
    }

    VV_STAMP(3);
    // per-row sum of squares of this wave's k-range (PRO_RMS): reduce over lanes
    float my_ssq = 0.f;
    if (a.pro == VV_PRO_RMS) {
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            if (r < Tt) {
                const float s = wave_sum(ssq[r]);
                if (frow == r) my_ssq = s;
            }
        }
    }

    VV_STAMP(4);
    // ---- split-K reduction through LDS, fixed order (deterministic) ----
    if (KS > 1) {
        __syncthreads();                            // staging tiles are dead; reuse the LDS
        f32x4* red = reinterpret_cast<f32x4*>(smem);
        float* redq = reinterpret_cast<float*>(smem + (size_t)WPB * NM * 64 * 16);
#pragma unroll
        for (int i = 0; i < NM; ++i) red[(wave * NM + i) * 64 + lane] = acc[i];
        redq[wave * 64 + lane] = my_ssq;
        __syncthreads();
        if (ks == 0) {
            for (int s = 1; s < KS; ++s) {
#pragma unroll
                for (int i = 0; i < NM; ++i) acc[i] += red[((wave + s) * NM + i) * 64 + lane];
                my_ssq += redq[(wave + s) * 64 + lane];
            }
        }
    }
    VV_STAMP(5);
    if (ks != 0 || tile0 >= n_tiles) return;

    // ---- epilogue: lane holds D[n = tile*16 + (lane>>4)*4 + r][t = lane&15] ----
    const int row = t0 + frow;
    if (frow >= Tt) return;
    float rs = 1.0f;
    if (a.pro == VV_PRO_RMS) rs = rsqrtf(my_ssq / (float)a.K + a.eps);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n0 = (tile0 + i) * 16 + fq * 4;
        if (n0 >= a.N) continue;
        float* yp = a.Y + (int64_t)row * a.ldy + n0;
        float o4[4] = {acc[i][0] * rs, acc[i][1] * rs, acc[i][2] * rs, acc[i][3] * rs};
        float u4[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (DUAL) { u4[0] = acc[NT + i][0] * rs; u4[1] = acc[NT + i][1] * rs; u4[2] = acc[NT + i][2] * rs; u4[3] = acc[NT + i][3] * rs; }
        if (a.epi == VV_EPI_CFG_DPM) {
            // lanes frow = j (cond) and frow = j + n (uncond) hold the two guidance branches of latent row j
            const int nc = a.n_cfg;
            const float ca = a.coef[0], cs_ = a.coef[1], csx = a.coef[2], c0 = a.coef[3], c1 = a.coef[4];
#pragma unroll 1
            for (int r = 0; r < 4; ++r) {
                const float vu = __shfl(o4[r], lane + nc);
                const int n = n0 + r;
                if (frow < nc && n < a.N) {
                    const float v = vu + a.cfg * (o4[r] - vu);
                    const int64_t zi = (int64_t)frow * a.N + n;
                    const float zo = a.z[zi];
                    const float x0 = ca * zo - cs_ * v;
                    float zn = csx * zo + c0 * x0 + c1 * (x0 - a.x0p[zi]);
                    if (a.sde_noise) zn += a.coef[5] * a.sde_noise[zi];      // sde-dpmsolver++ variance noise
                    a.x0p[zi] = x0;
                    a.z[zi] = zn;
                    a.z[zi + (int64_t)nc * a.N] = zn;
                }
            }
            continue;
        }
        if (vec_epi) {
            const float pb[4] = {pre_b[i].x, pre_b[i].y, pre_b[i].z, pre_b[i].w};
            const float py[4] = {pre_y[i].x, pre_y[i].y, pre_y[i].z, pre_y[i].w};
            const float pg[4] = {pre_g[i].x, pre_g[i].y, pre_g[i].z, pre_g[i].w};
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = o4[r];
                if (a.epi == VV_EPI_BIAS) o[r] += pb[r];
                else if (a.epi == VV_EPI_BIAS_GELU) o[r] = gelu_erf_f(o[r] + pb[r]);
                else if (a.epi == VV_EPI_SWIGLU) o[r] = silu_f(o[r]) * u4[r];
                else if (a.epi == VV_EPI_RESID) o[r] = (o[r] + pb[r]) * pg[r] + py[r];
                else if (a.epi == VV_EPI_GATED_RESID) o[r] = py[r] + pg[r] * o[r];
            }
            *reinterpret_cast<float4*>(yp) = float4{o[0], o[1], o[2], o[3]};
            continue;
        }
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + r;
            if (n >= a.N) break;
            float o = o4[r];
            if (a.bias && (a.epi == VV_EPI_BIAS || a.epi == VV_EPI_BIAS_GELU || a.epi == VV_EPI_RESID)) o += a.bias[n];
            if (a.epi == VV_EPI_BIAS_GELU) o = gelu_erf_f(o);
            else if (a.epi == VV_EPI_SWIGLU) o = silu_f(o) * u4[r];
            else if (a.epi == VV_EPI_RESID) { if (a.nscale) o *= a.nscale[n]; o += yp[r]; }
            else if (a.epi == VV_EPI_GATED_RESID) o = yp[r] + a.gate[(int64_t)row * a.ld_gate + n] * o;
            yp[r] = o;
        }
    }
    VV_STAMP(6);
    VV_BSTAMP(1);
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
template <int NT, int XS, bool DUAL, int WPB, int MAXR, bool VEC>
static void launch_t(const VVGemm& a, dim3 grid, size_t smem, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm_kernel<NT, XS, DUAL, WPB, MAXR, VEC>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((vv_gemm_kernel<NT, XS, DUAL, WPB, MAXR, VEC>), grid, dim3(WPB * 64), smem, s, a);
}

static int pow2_floor(int v) { int p = 1; while (p * 2 <= v) p *= 2; return p; }

extern "C" int vv_gemv_ok(const VVGemm* a);
extern "C" int vv_gemv_launch(VVGemm a, int xs, hipStream_t s);

// Chooses the kernel and its decomposition.  `xs` in {1,2,3}.
extern "C" int vv_tile_ok(const VVGemm* a, int xs);
extern "C" int vv_tile_launch(VVGemm a, int xs, hipStream_t s);
extern "C" int vv_gemm_launch(VVGemm a, int xs, hipStream_t s) {
    if (a.sl_n > 0) {                  // slot-batched rows: the MFMA tile form for tall row sets, else the 16-row GEMV form
        if (vv_tile_ok(&a, xs)) {
            const int r = vv_tile_launch(a, xs, s);
            if (r != -3) return r;
        }
        if (xs > 2 || !vv_gemv_ok(&a)) return -4;
        const int r = vv_gemv_launch(a, xs, s);
        return r == -3 ? -4 : r;
    }
    if (vv_tile_ok(&a, xs)) {      // tall activations: MFMA tile GEMM (tile.hip)
        const int r = vv_tile_launch(a, xs, s);
        if (r != -3) return r;
    }
    if (a.ksplit <= 0 && vv_gemv_ok(&a) && (a.T <= 4 || xs <= 2)) {
        const int r = vv_gemv_launch(a, xs, s);
        if (r != -3) return r;                 // -3: no instantiation for this (pair, rows, split mode): general kernel
    }
    if (a.kgrid > 1 || a.n_xa > 0 || a.n_ya > 0) return -4;      // part tensors exist only on the decode GEMV path
    const int n_tiles = (a.N + 15) / 16;
    const int k_tiles = (a.K + 31) / 32;
    const int t_tiles = (a.T + 15) / 16;
    const bool dual = a.epi == VV_EPI_SWIGLU;
    if (dual && !a.W2) return -1;
    const int Tt = a.T < 16 ? a.T : 16;
    const bool vec = ((a.K & 3) == 0) && ((a.ldx & 3) == 0) && ((((uintptr_t)a.X) & 15) == 0) &&
                     (a.pro != VV_PRO_RMS_MOD || (a.ld_mod & 3) == 0);
    // LDS row stride of the staging tile: odd multiples avoid bank conflicts on the 8-byte writes
    a.t_pad = Tt;
    // waves per block: 8 for decode-sized row counts (small staging tiles, deep K split), 4 otherwise
    const int wpb = (Tt <= 4 && vec) ? 8 : 4;
    // two tiles per wave once there are plenty of tiles (halves staging work per weight byte)
    int nt = 1;
    if (!dual && vec && (long)n_tiles * t_tiles >= 4096) nt = 2;
    const long work = ((long)n_tiles + nt - 1) / nt * t_tiles;     // waves if K is not split
    int ks = 1;
    if (a.ksplit > 0) ks = a.ksplit;
    else {
        // enough waves to cover 256 CUs x ~8, at least 8 k-steps (one batch) per wave
        const int want = (int)((2048 + work - 1) / work);
        ks = pow2_floor(want < 1 ? 1 : want);
        const int kmax = pow2_floor(k_tiles / 8 < 1 ? 1 : k_tiles / 8);
        if (ks > kmax) ks = kmax;
    }
    if (ks > wpb) ks = wpb;
    // few tiles: give every tile its own workgroup (more CUs pulling on HBM) rather than packing tiles into one
    if (a.ksplit <= 0 && wpb == 8 && work < 512) { while (ks < wpb && (k_tiles / (ks * 2)) >= 2) ks *= 2; }
    a.ksplit = ks;
    const int ng = wpb / ks;
    const int per_block = ng * nt;
    dim3 grid((n_tiles + per_block - 1) / per_block, t_tiles);
    const int nm = dual ? 2 : nt;
    size_t stage_b = (size_t)wpb * xs * 8 * 4 * a.t_pad * 16;
    size_t red_b = (size_t)wpb * nm * 64 * 16 + (size_t)wpb * 64 * 4;
    size_t smem = stage_b > red_b ? stage_b : red_b;
#define VV_XS(NT_, DUAL_, WPB_, MAXR_, VEC_)                                      \
    do {                                                                          \
        if (xs == 1) launch_t<NT_, 1, DUAL_, WPB_, MAXR_, VEC_>(a, grid, smem, s); \
        else if (xs == 2) launch_t<NT_, 2, DUAL_, WPB_, MAXR_, VEC_>(a, grid, smem, s); \
        else launch_t<NT_, 3, DUAL_, WPB_, MAXR_, VEC_>(a, grid, smem, s);        \
    } while (0)
    if (!vec) {
        if (dual) return -3;                       // SwiGLU operands are always aligned on this path
        VV_XS(1, false, 4, 16, false);             // generic (odd K / unaligned rows): encoder stem, tiny test shapes
    } else if (wpb == 8) {
        if (dual) VV_XS(1, true, 8, 4, true);
        else if (nt == 2) VV_XS(2, false, 8, 4, true);
        else VV_XS(1, false, 8, 4, true);
    } else {
        if (dual) VV_XS(1, true, 4, 16, true);
        else if (nt == 2) VV_XS(2, false, 4, 16, true);
        else VV_XS(1, false, 4, 16, true);
    }
#undef VV_XS
    return vv_launch_rc(0);
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
    return vv_launch_rc(0);
}
