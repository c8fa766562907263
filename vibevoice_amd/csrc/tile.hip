// tile.hip -- MFMA tile GEMM for tall activations (prompt prefill chunks, tokenizer stages with T >= 32).
//
//   Y[t][n] (op)= sum_k f(X[t][k]) * W[n][k]         X fp32 [T][K], W packed bf16 tiles (vv_common.h), Y fp32
//
// The decode GEMV (gemv.hip) gives every 16-feature tile its own workgroup and re-stages the activation rows per tile:
// right when the weight stream is the cost (T <= 16), wrong for prefill, where a 7B prompt chunk is compute.  Here a
// workgroup owns 64 rows x 128 features (64 for the two-matrix SwiGLU form):
//   * its 4 waves stage the 64 x 256-k activation chunk ONCE into LDS as bf16 B fragments (wave w converts row tile w,
//     one coalesced 1 KiB row read per instruction) -- double-buffered, the next chunk's global loads are issued before
//     the current chunk's MFMAs and converted after them;
//   * every wave then feeds each weight fragment (1 KiB, loaded once, one chunk ahead, straight into VGPRs) to 4 MFMAs,
//     one per row tile: 4x the arithmetic per weight byte of the GEMV, 4x fewer weight passes per prompt;
//   * RMSNorm: sum(x^2) per row is gathered while staging and applied to the accumulators in the epilogue.
// Prologues NONE / RMS; epilogues STORE / BIAS / BIAS_GELU / SWIGLU / RESID (what the LM and the tokenizer stages issue).
#include <cstdlib>
#include "vv_common.h"

namespace {

__device__ __forceinline__ float t_silu(float u) { return u / (1.0f + expf(-u)); }
__device__ __forceinline__ float t_gelu(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }
__device__ __forceinline__ float t_wave_sum(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));
    x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int XS>
__device__ __forceinline__ void t_split4(const float (&v)[4], uint2 (&out)[XS]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, m;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)v[j];
        if constexpr (XS > 1) m[j] = (__bf16)(v[j] - (float)h[j]);
    }
    out[0] = __builtin_bit_cast(uint2, h);
    if constexpr (XS > 1) out[1] = __builtin_bit_cast(uint2, m);
}

constexpr int BM = 64;          // rows per workgroup (4 row tiles, one staged by each wave)
constexpr int KC = 256;         // k per staged chunk = 8 k-steps = one float4 per lane and row
constexpr int UU = KC / 32;

template <int XS, int PRO, int EPI>
__global__ __launch_bounds__(256) void vv_gemm_tile_kernel(const u32x4* __restrict__ pW, const u32x4* __restrict__ pW2,
                                                           const float* __restrict__ pX, float* __restrict__ pY,
                                                           const float* __restrict__ pnw, int pT, int pN, int pK, int pldx,
                                                           int pldy, const VVGemm a) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int NTW = DUAL ? 1 : 2;                 // feature tiles per wave
    constexpr int NF = 2;                             // weight fragments per wave and k-step (2 tiles, or gate + up)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // [2 buffers][XS][UU][4 q][BM rows] x 16 B  = B fragments of row tile rt at rows rt*16..rt*16+15
    // each (k-step, k-group) plane of BM x 16 B is padded by 16 B: the 16 planes a staging store instruction touches land
    // in 16 different bank groups instead of one (16-way conflict without the pad)
    constexpr int GS = BM * 16 + 16;
    constexpr int BUF = XS * UU * 4 * GS;
    float* rs_sh = reinterpret_cast<float*>(smem + 2 * BUF);      // [BM]
    asm volatile("" ::"s"(a.bias), "s"(a.nscale), "s"(a.eps));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int t0 = (int)blockIdx.y * BM;
    const int n_tiles = (pN + 15) >> 4;
    const int k_tiles = (pK + 31) >> 5;
    const int tile0 = ((int)blockIdx.x * 4 + wave) * NTW;
    const bool wact = tile0 < n_tiles;                             // wave has at least one live feature tile
    const int n_chunks = (pK + KC - 1) / KC;
    const unsigned kk = lane * 4;
    const unsigned st_off = ((kk >> 5) * 4 + ((kk & 31) >> 3)) * (BM * 16 + 16) + (kk & 7) * 2;

    // ---- staging: wave w owns rows t0 + w*16 .. +15 ----
    // row offsets (wave-uniform), once.  Slot-batched rows (VVGemm::sl_*, the batched tokenizer chains): logical row t belongs to
    // utterance slot j = t / sl_T, local row t % sl_T of that slot's streaming buffer (a side with stride 0 is a dense scratch)
    size_t xoff[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        int t = t0 + wave * 16 + r;
        if (t >= pT) t = pT - 1;                                    // clamped: legal address, never stored
        if (a.sl_n > 0 && a.sl_x) {
            const int j = t / a.sl_T, tt = t - j * a.sl_T;
            xoff[r] = (size_t)vv_slot_id(a.sl_id, j) * a.sl_x + (size_t)tt * pldx;
        } else xoff[r] = (size_t)t * pldx;
    }
    float4 xr[16];
    float4 nwv;
    float ssq[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ssq[r] = 0.f;
    auto x_load = [&](int c) {
        unsigned k = c * KC + kk;
        const bool kin = k < (unsigned)pK;
        if (!kin) k = 0;
        nwv = (PRO == VV_PRO_RMS && pnw) ? *reinterpret_cast<const float4*>(pnw + k) : float4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = *reinterpret_cast<const float4*>(pX + xoff[r] + k);
    };
    auto x_stage = [&](int c, unsigned char* buf) {
        const float msk = (c * KC + kk < (unsigned)pK) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v[4] = {xr[r].x * msk, xr[r].y * msk, xr[r].z * msk, xr[r].w * msk};
            if constexpr (PRO == VV_PRO_RMS) {
                ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                v[0] *= nwv.x; v[1] *= nwv.y; v[2] *= nwv.z; v[3] *= nwv.w;
            }
            uint2 parts[XS];
            t_split4<XS>(v, parts);
#pragma unroll
            for (int p = 0; p < XS; ++p)
                *reinterpret_cast<uint2*>(buf + p * (UU * 4 * GS) + st_off + (wave * 16 + r) * 16) = parts[p];
        }
    };
    // ---- weights: NF fragments per k-step, HALF a chunk (4 k-steps) per register buffer, loaded one half ahead: two
    //      buffers of 32 VGPRs keep the kernel at two workgroups per CU (a second wave per SIMD hides the LDS waits) ----
    constexpr int HU = UU / 2;
    u32x4 wq[2][HU][NF];
    const u32x4* wb0 = pW + (size_t)tile0 * k_tiles * 64 + lane;
    const u32x4* wb1 = DUAL ? pW2 + (size_t)tile0 * k_tiles * 64 + lane
                            : pW + (size_t)min(tile0 + 1, n_tiles - 1) * k_tiles * 64 + lane;   // 2nd tile (clamped, masked at the store)
    auto w_load = [&](int c, int half, u32x4 (&dst)[HU][NF]) {
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int kt = min(c * UU + half * HU + u, k_tiles - 1);
            dst[u][0] = wb0[(size_t)kt * 64];
            dst[u][1] = wb1[(size_t)kt * 64];
        }
    };

    f32x4 acc[NF][4];
#pragma unroll
    for (int i = 0; i < NF; ++i)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[i][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int c, int half, const unsigned char* buf, const u32x4 (&w)[HU][NF]) {
#pragma unroll
        for (int uu = 0; uu < HU; ++uu) {
            const int u = half * HU + uu;
            if (c * UU + u < k_tiles) {
#pragma unroll
                for (int p = 0; p < XS; ++p) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        const bf16x8 xb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(
                            buf + (size_t)((p * UU + u) * 4 + fq) * GS + (rt * 16 + frow) * 16));
#pragma unroll
                        for (int i = 0; i < NF; ++i)
                            acc[i][rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[uu][i]), xb, acc[i][rt], 0, 0, 0);
                    }
                }
            }
        }
    };

    x_load(0);
    if (wact) w_load(0, 0, wq[0]);
    x_stage(0, smem);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        unsigned char* cur = smem + (c & 1) * BUF;
        unsigned char* nxt = smem + ((c + 1) & 1) * BUF;
        const bool n1 = c + 1 < n_chunks;
        if (n1) x_load(c + 1);
        if (wact) { w_load(c, 1, wq[1]); compute(c, 0, cur, wq[0]); }
        if (wact) { if (n1) w_load(c + 1, 0, wq[0]); compute(c, 1, cur, wq[1]); }
        if (n1) x_stage(c + 1, nxt);
        __syncthreads();
    }
    // ---- per-row 1/rms ----
    if constexpr (PRO == VV_PRO_RMS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = t_wave_sum(ssq[r]);
            if (lane == 0) rs_sh[wave * 16 + r] = rsqrtf(s / (float)pK + a.eps);
        }
        __syncthreads();
    }
    if (!wact) return;
    // ---- epilogue: lane holds D[n = tile*16 + fq*4 + j][t = rt*16 + frow] ----
#pragma unroll
    for (int i = 0; i < (DUAL ? 1 : NF); ++i) {
        const int tile = tile0 + i;
        if (tile >= n_tiles) break;
        const int n0 = tile * 16 + fq * 4;
        if (n0 >= pN) continue;
        float4 pb = {0.f, 0.f, 0.f, 0.f}, pg = {1.f, 1.f, 1.f, 1.f};
        if constexpr (EPI == VV_EPI_BIAS || EPI == VV_EPI_BIAS_GELU || EPI == VV_EPI_RESID) {
            if (a.bias) pb = *reinterpret_cast<const float4*>(a.bias + n0);
        }
        if constexpr (EPI == VV_EPI_RESID) {
            if (a.nscale) pg = *reinterpret_cast<const float4*>(a.nscale + n0);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int t = t0 + rt * 16 + frow;
            if (t >= pT) continue;
            const float rs = (PRO == VV_PRO_RMS) ? rs_sh[rt * 16 + frow] : 1.0f;
            float o[4] = {acc[i][rt][0] * rs, acc[i][rt][1] * rs, acc[i][rt][2] * rs, acc[i][rt][3] * rs};
            size_t yo = (size_t)t * pldy;
            if (a.sl_n > 0 && a.sl_y) {
                const int j = t / a.sl_T, tt = t - j * a.sl_T;
                yo = (size_t)vv_slot_id(a.sl_id, j) * a.sl_y + (size_t)tt * pldy;
            }
            float* yp = pY + yo + n0;
            if constexpr (EPI == VV_EPI_BIAS) {
                o[0] += pb.x; o[1] += pb.y; o[2] += pb.z; o[3] += pb.w;
            } else if constexpr (EPI == VV_EPI_BIAS_GELU) {
                o[0] = t_gelu(o[0] + pb.x); o[1] = t_gelu(o[1] + pb.y); o[2] = t_gelu(o[2] + pb.z); o[3] = t_gelu(o[3] + pb.w);
            } else if constexpr (EPI == VV_EPI_SWIGLU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = t_silu(o[j]) * (acc[1][rt][j] * rs);
            } else if constexpr (EPI == VV_EPI_RESID) {
                const float4 py = *reinterpret_cast<const float4*>(yp);
                o[0] = py.x + pg.x * (o[0] + pb.x); o[1] = py.y + pg.y * (o[1] + pb.y);
                o[2] = py.z + pg.z * (o[2] + pb.z); o[3] = py.w + pg.w * (o[3] + pb.w);
            }
            *reinterpret_cast<float4*>(yp) = float4{o[0], o[1], o[2], o[3]};
        }
    }
}

}  // namespace

#define VV_TILE_COMBOS(X)                                                                      \
    X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_RESID)      \
    X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_RMS, VV_EPI_BIAS_GELU) X(VV_PRO_RMS, VV_EPI_SWIGLU)    \
    X(VV_PRO_RMS, VV_EPI_STORE)

// Eligibility: tall, aligned, enough workgroups (>= 48), one of the pairs above, bench / two-term activation modes, no row re-mapping or part tensors.
extern "C" int vv_tile_ok(const VVGemm* a, int xs) {
    if (a->T < 32 || xs > 2) return 0;
    {   // enough workgroups to occupy the chip; smaller problems (tokenizer stages at decode) stay on the row-tiled GEMV
        const int per_wg = 4 * (a->epi == VV_EPI_SWIGLU ? 1 : 2);
        const int64_t wgs = (int64_t)(((a->N + 15) / 16 + per_wg - 1) / per_wg) * ((a->T + BM - 1) / BM);
        static int min_sl = -1;
        if (min_sl < 0) { const char* e = getenv("VVHIP_TILE_MIN_WGS_SLOTS"); min_sl = e ? atoi(e) : 40; }
        // slot-batched tokenizer stages (several utterances' frames in one launch): the 16-row GEMV form re-normalises and
        // re-stages its 16 rows in every one of its (feature tiles x row tiles) workgroups -- 87 us for the C = 256 FFN1 of eight
        // utterances (1600 rows x 1024 features x 256: 0.8 GFLOP) -- so the tile form takes over much earlier there
        const int min_wgs = a->sl_n > 0 ? min_sl : 48;
        if (wgs < min_wgs) return 0;
    }
    if (a->sl_n > 0) {
        if (a->sl_n > 8 || a->sl_T < 1 || a->T != a->sl_n * a->sl_T || a->sl_x < 0 || a->sl_y < 0 || (a->sl_x & 3) || (a->sl_y & 3)) return 0;
    }
    if ((a->K & 3) || (a->ldx & 3) || (a->N & 3) || (a->ldy & 3)) return 0;
    if ((((uintptr_t)a->X) | ((uintptr_t)a->Y)) & 15) return 0;
    if ((a->nw && (((uintptr_t)a->nw) & 15)) || (a->bias && (((uintptr_t)a->bias) & 15)) || (a->nscale && (((uintptr_t)a->nscale) & 15))) return 0;
    if (a->x_row_mod > 0 || a->add_rows_per_vec > 0 || a->kgrid > 1 || a->n_xa > 0 || a->n_ya > 0 || a->ksplit > 0) return 0;
    if (a->K < 32) return 0;
    if (a->epi == VV_EPI_SWIGLU && !a->W2) return 0;
#define X(P, E) if (a->pro == P && a->epi == E) return 1;
    VV_TILE_COMBOS(X)
#undef X
    return 0;
}

template <int XS, int PRO, int EPI>
static int tile_go(const VVGemm& a, hipStream_t s) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    const int n_tiles = (a.N + 15) / 16;
    const int per_wg = 4 * (DUAL ? 1 : 2);
    dim3 grid((n_tiles + per_wg - 1) / per_wg, (a.T + BM - 1) / BM);
    const size_t smem = (size_t)2 * XS * UU * 4 * (BM * 16 + 16) + BM * 4;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm_tile_kernel<XS, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((vv_gemm_tile_kernel<XS, PRO, EPI>), grid, dim3(256), smem, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);
    return vv_launch_rc(0);
}

extern "C" int vv_tile_launch(VVGemm a, int xs, hipStream_t s) {
#define X(P, E) if (a.pro == P && a.epi == E) return xs == 1 ? tile_go<1, P, E>(a, s) : tile_go<2, P, E>(a, s);
    VV_TILE_COMBOS(X)
#undef X
    return -3;
}
