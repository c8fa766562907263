// prefill.hip -- the MFMA-bound half of the path: prompt prefill in the bf16-activation mode (xsplit = 1).
//
// The decode kernels (gemv.hip / attn.hip) are built around streaming weights once per token; a 10 K-token 7B prompt is
// 190 TFLOP of dense work instead, and the round-1 prefill (tile.hip: 64 x 128 tiles, fp32 activations converted per
// tile, weights straight from L2 into VGPRs; attention re-reading K/V for every head and every 16 query rows) sat at
// ~0.4 PFLOP/s for the GEMMs with a third of the time in attention.  This file is the prefill path rebuilt around LDS:
//
//   vv_pack_rows_kernel     fp32 rows [T][K] (+ RMSNorm) -> bf16 MFMA fragments [T/16][K/32][64 lanes][8]: the activation
//                           becomes a ready-made B operand, exactly like the packed weights are ready-made A operands
//                           (vv_common.h), so a GEMM stage is a straight 1 KiB-per-wave-instruction copy into LDS.
//   vv_gemm3_kernel         Y[t][n] (op)= sum_k X[t][k] W[n][k]: 128 rows x 128 features per workgroup, 4 waves of
//                           64 x 64 (4 x 4 MFMA 16x16x32 accumulators), K stepped 64 at a time; both operands arrive in
//                           LDS by global_load_lds_dwordx4 (no VGPR round trip, lane-linear = the fragment order, no bank
//                           conflicts on the ds_read_b128 that follow); epilogues bias / residual / SwiGLU, the SwiGLU one
//                           writing its result as packed bf16 fragments for the down projection.  Workgroup ids are mapped
//                           so that one XCD (own L2) owns a contiguous range of feature blocks.
//   vv_gemm4_kernel         the long-prompt form: 256 x 256 tile, 4-slot ring of 32-wide K stages, the two wave halves in
//                           anti-phase (one computes while the other loads), strip-ordered tiles (an XCD's L2 serves 4 weight
//                           blocks x 8 row blocks at a time), the partial last round split along K; one more epilogue: bias +
//                           RoPE + KV-cache append for the QKV projection.
//   vv_attn_prefill4_kernel causal attention for a chunk of consecutive positions: a workgroup owns 4 units (64 query rows x 1 query
//                           head) of one kv head, streams the causal prefix ONCE, 64 positions per stage through a 4-slot LDS
//                           ring; 8 waves = 4 units x 2 row halves in anti-phase (matrix segment / softmax segment), one
//                           lazy-rescaled softmax step per stage, row sums through the matrix pipe; the result leaves as the
//                           o-projection's packed operand.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "vv_common.h"

namespace {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

// one wave-instruction: 64 lanes x 16 B from per-lane global addresses into LDS at (wave-uniform) dst + lane * 16
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((gvoid_t*)gsrc, (lvoid_t*)lds_dst, 16, 0, 0);
}

// every LDS-DMA copy this wave issued has landed, then the workgroup barrier: after it every wave's copies are visible
__device__ __forceinline__ void stage_sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ float p_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------------ activation packing
// grid = ceil(T / 16) workgroups x 256 threads.  Rows >= T and columns >= K are written as zeros.
__global__ __launch_bounds__(256) void vv_pack_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw,
                                                           float eps, u32x4* __restrict__ xp, int T, int K) {
    __shared__ float rs_sh[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * 16;
    const int KT = (K + 31) >> 5;
    if (nw != nullptr) {            // RMSNorm: 1/rms per row; wave w owns rows 4w .. 4w+3
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + wave * 4 + r;
            float s = 0.f;
            if (t < T) {
                const float* xr = x + (int64_t)t * ldx;
                // six loads in flight per trip (H = 1536: one trip; 3584: two + one pair): a row of a short prompt's 21-workgroup
                // launch is pure load latency otherwise
                int k = lane * 4;
                for (; k + 5 * 256 < K; k += 6 * 256) {
                    float4 v[6];
#pragma unroll
                    for (int u = 0; u < 6; ++u) v[u] = *reinterpret_cast<const float4*>(xr + k + u * 256);
#pragma unroll
                    for (int u = 0; u < 6; ++u) s += v[u].x * v[u].x + v[u].y * v[u].y + v[u].z * v[u].z + v[u].w * v[u].w;
                }
                for (; k + 256 < K; k += 2 * 256) {
                    const float4 v0 = *reinterpret_cast<const float4*>(xr + k), v1 = *reinterpret_cast<const float4*>(xr + k + 256);
                    s += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
                    s += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
                }
                for (; k < K; k += 256) {
                    const float4 v = *reinterpret_cast<const float4*>(xr + k);
                    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                }
            }
            s = p_wave_sum(s);
            if (lane == 0) rs_sh[wave * 4 + r] = rsqrtf(s / (float)K + eps);
        }
        __syncthreads();
    }
    const int row = lane & 15, kq = lane >> 4;
    const int t = t0 + row;
    const float rs = (nw != nullptr) ? rs_sh[row] : 1.0f;
    const float* xr = x + (int64_t)(t < T ? t : 0) * ldx;
    for (int kt = wave; kt < KT; kt += 4) {
        const int k = kt * 32 + kq * 8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = 0.f;
        if (t < T && k < K) {       // K % 8 == 0 is a launch precondition
            const float4 a = *reinterpret_cast<const float4*>(xr + k), b = *reinterpret_cast<const float4*>(xr + k + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            if (nw != nullptr) {
                const float4 wa = *reinterpret_cast<const float4*>(nw + k), wb = *reinterpret_cast<const float4*>(nw + k + 4);
                const float w8[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] * rs * w8[j];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)v[j];
        xp[((int64_t)blockIdx.x * KT + kt) * 64 + lane] = __builtin_bit_cast(u32x4, o);
    }
}

// adaLN input rows of every solver step, straight into MFMA fragments: logical row t = i * rows + r holds
// SiLU(cond_proj[r] + t_emb[i]) (DiT adaLN_modulation's SiLU, modular_vibevoice_diffusion_head.py:198-205).
// grid (ceil(T / 16), ceil(KT / 4)); wave w of a workgroup packs k-tile 4 * blockIdx.y + w of its 16-row tile.
__global__ __launch_bounds__(256) void vv_ada_pack_kernel(const float* __restrict__ cproj, const float* __restrict__ temb,
                                                          u32x4* __restrict__ xp, int rows, int n_steps, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int KT = (H + 31) >> 5, T = rows * n_steps;
    const int kt = blockIdx.y * 4 + wave;
    if (kt >= KT) return;
    const int t = blockIdx.x * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (__bf16)0.f;
    if (t < T && k < H) {               // H % 8 == 0 is a launch precondition
        const int i = t / rows, r = t - i * rows;
        const float* c = cproj + (int64_t)r * H + k;
        const float* e = temb + (int64_t)i * H + k;
        const float4 c0 = *reinterpret_cast<const float4*>(c), c1 = *reinterpret_cast<const float4*>(c + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(e), e1 = *reinterpret_cast<const float4*>(e + 4);
        const float u[8] = {c0.x + e0.x, c0.y + e0.y, c0.z + e0.z, c0.w + e0.w, c1.x + e1.x, c1.y + e1.y, c1.z + e1.z, c1.w + e1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (__bf16)(u[j] / (1.0f + expf(-u[j])));
    }
    xp[((int64_t)blockIdx.x * KT + kt) * 64 + lane] = __builtin_bit_cast(u32x4, o);
}

// packed bf16 fragments -> fp32 rows (tests)
__global__ void vv_unpack_rows_kernel(const __bf16* __restrict__ xp, float* __restrict__ x, int T, int K) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)T * K) return;
    const int t = (int)(e / K), k = (int)(e - (int64_t)t * K);
    const int KT = (K + 31) >> 5;
    const int64_t tile = (int64_t)(t >> 4) * KT + (k >> 5);
    const int lane = (t & 15) + 16 * ((k & 31) >> 3);
    x[e] = (float)xp[(tile * 64 + lane) * 8 + (k & 7)];
}

// ------------------------------------------------------------------------------------------------ GEMM
struct VVGemm3 {
    const u32x4* W;        // packed [N][K]
    const u32x4* W2;       // second matrix (SwiGLU "up"), same shape
    const u32x4* Xp;       // packed activations [T][K]
    float* Y;              // fp32 [T][ldy]            (STORE / BIAS / RESID)
    u32x4* Yp;             // packed bf16 [T][N]       (SWIGLU)
    const float* bias;     // [N] or null
    int T, N, K, ldy;
    int n_blocks, t_blocks;
    // vv_gemm4 only: the tiles past the last whole round of 256 workgroups are split along K (see the kernel)
    int sfw;               // feature blocks per strip of the workgroup -> tile order (L2 blocking)
    int full_idx;          // per XCD: tiles [0, full_idx) of its range are computed whole
    int split;             // k-parts of every remaining tile (1: no split)
    float* ws;             // partial accumulators: [256 slots][32 accumulators][512 threads] f32x4
    unsigned* flags;       // [256] arrival words, zero between launches
    unsigned* err;         // host-visible word: set when a wait timed out
    // VV_EPI_QKV_ROPE (vv_gemm4 only, head_dim 128): N = (Hq + 2 Hkv) * 128 features = [q heads | k heads | v heads]
    const VVRow* rows;     // rows[0] = (cache, first position); row t is position rows[0].pos + t of that cache
    const float2* rope_tab;    // (cos, sin)[pos][64]
    float* q_out;          // fp32 [T][Hq][128]: rotated, scaled queries
    __bf16* kc; __bf16* vc;    // this layer's K / V caches (tile layouts of attn.hip)
    int64_t cache_stride, head_stride;
    int Hq, Hkv;
    float q_scale;
};

__device__ __forceinline__ float g3_silu(float u) { return u / (1.0f + __expf(-u)); }

// TR = row tiles (16 rows each) per wave: the workgroup covers 128 features x 32*TR rows.  TR = 8 (256 rows) doubles the
// MFMAs per staged byte and per barrier: three such workgroups per CU carry enough arithmetic to cover a stage's load latency.
// DB = 1: two stage buffers (TR = 4 only), the next stage's copies issued before the current stage's MFMAs, one barrier per stage.
// Launches of at most ~2 workgroups per CU (short prompts: a few hundred tiles) have no neighbour workgroup to hide a stage's
// load latency behind; with four resident workgroups per CU the single-buffer form overlaps through occupancy and keeps its LDS.
template <int EPI, int TR, int DB = 0>
__global__ __launch_bounds__(256, TR == 8 ? 2 : (DB ? 2 : 4)) void vv_gemm3_kernel(const VVGemm3 a) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int FT = DUAL ? 4 : 8;              // feature tiles (per matrix) per workgroup
    // LDS stage: 16 A fragments then 4*TR B fragments of one 64-wide K step (2 k-tiles): [frag][64 lanes][16 B]
    constexpr int NFR = 16 + 4 * TR;               // fragments per stage
    constexpr int FPW = NFR / 4;                   // copied by each wave
    static_assert(DB == 0 || TR == 4, "two stage buffers: 64 KiB of static LDS, TR = 4 only");
    __shared__ __attribute__((aligned(16))) unsigned char stage_all[NFR * 1024 * (DB + 1)];
    unsigned char* stage = stage_all;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int KT = (a.K + 31) >> 5;
    const int n_tiles = (a.N + 15) >> 4, t_tiles = (a.T + 15) >> 4;
    // ---- workgroup -> (feature block, row block).  Consecutive workgroup ids land on consecutive XCDs; give each XCD a
    // contiguous range of feature blocks and walk the row blocks of one feature block back to back on the same XCD, so the
    // block's weights are fetched into one L2 once and reused by its row blocks.  Bijective for any block count.
    int nb, tb;
    {
        const int total = a.n_blocks * a.t_blocks;
        const int bid = blockIdx.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = total >> 3, r = total & 7;
        const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;     // position in XCD-major order
        nb = lin / a.t_blocks;
        tb = lin - nb * a.t_blocks;
    }
    const int ft0 = nb * FT;                       // first feature tile of this workgroup (in each matrix)
    const int tt0 = tb * 2 * TR;                   // first row tile
    const int wr = wave >> 1, wc = wave & 1;       // 2 x 2 waves: rows wr*16*TR.., features wc*64..

    // ---- stage loader: wave w copies fragments FPW*w .. FPW*w + FPW-1 of the NFR ----
    // fragment f < 16: A, feature-tile slot f >> 1 (DUAL: slots 0-3 gate, 4-7 up), k-tile f & 1;  f >= 16: B, row tile (f-16) >> 1
    const u32x4* src[FPW];
#pragma unroll
    for (int i = 0; i < FPW; ++i) {
        const int f = wave * FPW + i;
        if (f < 16) {
            const int slot = f >> 1;
            const u32x4* base = (DUAL && slot >= 4) ? a.W2 : a.W;
            int ft = ft0 + (DUAL ? (slot & 3) : slot);
            if (ft > n_tiles - 1) ft = n_tiles - 1;                   // clamped: legal address, masked at the store
            src[i] = base + (int64_t)ft * KT * 64 + lane;
        } else {
            int tt = tt0 + ((f - 16) >> 1);
            if (tt > t_tiles - 1) tt = t_tiles - 1;
            src[i] = a.Xp + (int64_t)tt * KT * 64 + lane;
        }
    }
    f32x4 acc[4][TR];                              // [feature tile of this wave][row tile of this wave]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TR; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int n_steps = (KT + 1) >> 1;
    // short prompts: grid.y = K parts (a.split), each writes a dense fp32 partial tensor (vv_gemm3_launch / vv_g3_reduce_kernel)
    const int ks = (int)gridDim.y, part = (int)blockIdx.y;
    const int s_lo = (int)((int64_t)part * n_steps / ks), s_hi = (int)((int64_t)(part + 1) * n_steps / ks);
    auto issue = [&](int s, unsigned char* buf) {
#pragma unroll
        for (int i = 0; i < FPW; ++i) {
            const int f = wave * FPW + i;
            int kt = s * 2 + (f & 1);
            if (kt > KT - 1) kt = KT - 1;                              // odd K tail: re-reads the last k-tile, MFMA skipped
            glds16(src[i] + (int64_t)kt * 64, buf + f * 1024);
        }
    };
    if constexpr (DB) { if (s_lo < s_hi) issue(s_lo, stage_all); }
#pragma unroll 1
    for (int s = s_lo; s < s_hi; ++s) {
        const int kt0 = s * 2;
        if constexpr (DB) {
            stage = stage_all + ((s - s_lo) & 1) * (NFR * 1024);
            stage_sync();                                             // stage s has landed for every wave; everyone is done with stage s - 1
            if (s + 1 < s_hi) issue(s + 1, stage_all + (((s - s_lo) & 1) ^ 1) * (NFR * 1024));
        } else {
            issue(s, stage);
            stage_sync();                                             // the stage has landed for every wave
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kt0 + kk < KT) {
                bf16x8 af[4], bfr[TR];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // this wave's feature tiles: plain: slots wc*4+i;  DUAL: i < 2 gate slots wc*2+i, i >= 2 up slots 4+wc*2+(i-2)
                    const int slot = DUAL ? ((i < 2) ? (wc * 2 + i) : (4 + wc * 2 + (i - 2))) : (wc * 4 + i);
                    af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(stage + ((slot * 2 + kk) * 64 + lane) * 16));
                }
#pragma unroll
                for (int j = 0; j < TR; ++j)
                    bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(stage + ((16 + (wr * TR + j) * 2 + kk) * 64 + lane) * 16));
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < TR; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        if constexpr (!DB) __syncthreads();                           // everyone is done reading before the next stage lands
    }

    // ---- epilogue: lane holds D[n = tile*16 + fq*4 + r][t = ttile*16 + frow] ----
    if constexpr (DUAL) {
        const int KTo = (a.N + 31) >> 5;                               // k-tiles of the OUTPUT operand (its K is our N)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ft = ft0 + wc * 2 + i;
            if (ft >= n_tiles) continue;
            const int n0 = ft * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < TR; ++j) {
                const int tt = tt0 + wr * TR + j;
                if (tt >= t_tiles) continue;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = (n0 + r < a.N) && (tt * 16 + frow < a.T);
                    o[r] = (__bf16)(live ? g3_silu(acc[i][j][r]) * acc[2 + i][j][r] : 0.f);
                }
                // element (t, n) of the packed output: tile (tt, n >> 5), lane (t & 15) + 16 * ((n & 31) >> 3), slot n & 7
                const int64_t tile = (int64_t)tt * KTo + (n0 >> 5);
                const int ol = frow + 16 * ((n0 & 31) >> 3);
                unsigned char* dst = reinterpret_cast<unsigned char*>(a.Yp) + ((tile * 64 + ol) * 16 + (n0 & 7) * 2);
                *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, o);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ft = ft0 + wc * 4 + i;
            if (ft >= n_tiles) continue;
            const int n0 = ft * 16 + fq * 4;
            if (n0 >= a.N) continue;                                   // N % 4 == 0 is a launch precondition
            float4 pb = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == VV_EPI_BIAS) {
                if (a.bias) pb = *reinterpret_cast<const float4*>(a.bias + n0);
            }
#pragma unroll
            for (int j = 0; j < TR; ++j) {
                const int t = (tt0 + wr * TR + j) * 16 + frow;
                if (t >= a.T) continue;
                if (ks > 1) {                                          // K part: the raw sums, bias / residual belong to the reduce
                    *reinterpret_cast<float4*>(a.ws + ((int64_t)part * a.T + t) * a.N + n0) = float4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    continue;
                }
                float* yp = a.Y + (int64_t)t * a.ldy + n0;
                float4 o = {acc[i][j][0] + pb.x, acc[i][j][1] + pb.y, acc[i][j][2] + pb.z, acc[i][j][3] + pb.w};
                if constexpr (EPI == VV_EPI_RESID) {
                    const float4 py = *reinterpret_cast<const float4*>(yp);
                    o.x += py.x; o.y += py.y; o.z += py.z; o.w += py.w;
                }
                *reinterpret_cast<float4*>(yp) = o;
            }
        }
    }
}

// Y[t][n] (+)= bias[n] + sum over the K parts, in part order (deterministic): the second half of a K-split vv_gemm3 launch
__global__ __launch_bounds__(256) void vv_g3_reduce_kernel(const float* __restrict__ ws, int parts, int T, int N, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int ldy, int resid) {
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= (int64_t)T * N) return;
    const int t = (int)(e / N), n = (int)(e - (int64_t)t * N);
    float4 v[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) v[p] = (p < parts) ? *reinterpret_cast<const float4*>(ws + (int64_t)p * T * N + e) : float4{0.f, 0.f, 0.f, 0.f};
    float* yp = Y + (int64_t)t * ldy + n;
    float4 o = bias ? *reinterpret_cast<const float4*>(bias + n) : float4{0.f, 0.f, 0.f, 0.f};
    float4 sum = v[0];
#pragma unroll
    for (int p = 1; p < 8; ++p) { sum.x += v[p].x; sum.y += v[p].y; sum.z += v[p].z; sum.w += v[p].w; }
    o.x += sum.x; o.y += sum.y; o.z += sum.z; o.w += sum.w;
    if (resid) { const float4 py = *reinterpret_cast<const float4*>(yp); o.x += py.x; o.y += py.y; o.z += py.z; o.w += py.w; }
    *reinterpret_cast<float4*>(yp) = o;
}

// workgroup -> (feature block, row block) of the 256 x 256 kernels, XCD-major (see vv_gemm3_kernel), plus the K-split of the
// partial last round: part < 0: a whole tile; else this workgroup's k-part of a split tile, slot0 = the tile's first partial slot.
// Returns false for the padding workgroups of XCDs whose range is one tile shorter (split region only).
__device__ __forceinline__ bool g4_map(const VVGemm3& a, int& nb, int& tb, int& part, int& slot0) {
    const int split = a.split;
    part = 0; slot0 = 0;
    const int total = a.n_blocks * a.t_blocks;
    const int bid = blockIdx.x;
    const int q = total >> 3, r = total & 7;
    int xcd, idx;
    if (split <= 1 || bid < 8 * a.full_idx) {
        xcd = bid & 7; idx = bid >> 3;
        part = -1;
    } else {
        // the last, partial round: `split` consecutive workgroups of an XCD share one tile, each takes 1/split of K; the
        // LAST of them (highest id: dispatched after the others, so what it waits for is already running) adds the others'
        // partial accumulators in part order and runs the epilogue
        const int b2 = bid - 8 * a.full_idx;
        xcd = b2 & 7;
        const int j = b2 >> 3, tr = j / split;
        part = j - tr * split;
        idx = a.full_idx + tr;
        slot0 = xcd * 32 + tr * (split - 1);
    }
    if (idx >= q + (xcd < r ? 1 : 0)) return false;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    // lin -> tile in STRIP order: strips of `sfw` feature blocks, inside a strip row block by row block.  The ~32 workgroups an
    // XCD runs at any time are consecutive lin = sfw feature blocks x 32 / sfw row blocks: per K stage its L2 fetches sfw weight
    // slabs + 32 / sfw activation slabs instead of 1 + 32 (feature-major order: every workgroup's activation rows came over
    // the fabric -- 12.0 GB of L2 misses per 7B gate/up GEMM at T = 10,922, 4.8 TB/s: the kernel was fabric-bound).
    const int sfw = a.sfw;
    const int strip = sfw * a.t_blocks;
    int sf = lin / strip;
    const int n_sf = (a.n_blocks + sfw - 1) / sfw;
    if (sf > n_sf - 1) sf = n_sf - 1;
    const int wdt = (sf == n_sf - 1) ? a.n_blocks - sf * sfw : sfw;      // the last strip may be narrower
    const int rem = lin - sf * strip;
    tb = rem / wdt;
    nb = sf * sfw + (rem - tb * wdt);
    return true;
}

// what follows the K loop of a 256 x 256 workgroup (8 waves as 2 x 4: features wf * 128.., rows wr * 64..): the hand-over of a
// split tile's partial accumulators, then the epilogue.  Lane holds D[n = tile*16 + fq*4 + r][t = ttile*16 + frow].
template <int EPI>
__device__ __forceinline__ void g4_finish(const VVGemm3& a, f32x4 (&acc)[8][4], int part, int slot0, int ft0, int tt0, int wf, int wr) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    const int tid = threadIdx.x, lane = tid & 63;
    const int frow = lane & 15, fq = lane >> 4;
    const int n_tiles = (a.N + 15) >> 4, t_tiles = (a.T + 15) >> 4;
    const int split = a.split;
    if (part >= 0) {
        // ---- split tile: hand the partial accumulators over (cdna_hip_programming.md, Guideline 16: write-through payload,
        // drained by every wave, one agent-scope arrival word per contributor; the consumer polls relaxed, acquires once) ----
        constexpr size_t SLOT = (size_t)32 * 512 * 4;                 // floats per partial slot
        const unsigned voff = (unsigned)tid * 16u;                      // byte offset of this thread inside one accumulator plane
        if (part < split - 1) {
            const char* sbase = reinterpret_cast<const char*>(a.ws + (size_t)(slot0 + part) * SLOT);      // wave-uniform
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(voff), "v"(acc[i][j]), "s"(sbase + (size_t)(i * 4 + j) * 8192) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(a.flags + slot0 + part, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        // A hand-over that times out (a lost producer; the 100 MHz wall clock keeps running while the queue is preempted by
        // another process or a profiler, hence 2 s: far beyond any producer's running time) marks the host-mapped error word
        // and LEAVES THE ARRIVAL WORDS ALONE -- a late producer must never find a word this launch re-armed and the next launch
        // must not consume stale partials.  The tile itself is then summed from whatever the slots hold and is wrong; the host
        // sees the word at the next sync / enqueue (engine.hip: ksplit_check), re-zeroes the words behind a stream sync and
        // reports the prompt pass as failed.  (Skipping the epilogue instead cost the QKV form 200 B of scratch per lane.)
        bool lost = false;                          // meaningful in wave 0, whose lanes also re-arm the words below
        if (tid < 64) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (int p = 0; p < split - 1 && !lost; ++p) {
                for (;;) {
                    if (__hip_atomic_load(a.flags + slot0 + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
                    __builtin_amdgcn_s_sleep(16);
                    if ((unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0) > 200000000ull) { lost = true; break; }
                }
            }
            if (lost && tid == 0) __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
#pragma unroll 1
        for (int p = 0; p < split - 1; ++p) {
            const char* sbase = reinterpret_cast<const char*>(a.ws + (size_t)(slot0 + p) * SLOT);        // wave-uniform
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(v[j]) : "v"(voff), "s"(sbase + (size_t)(i * 4 + j) * 8192) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += v[j];
            }
        }
        if (tid < split - 1 && !lost) __hip_atomic_store(a.flags + slot0 + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    // ---- epilogue: lane holds D[n = tile*16 + fq*4 + r][t = ttile*16 + frow] ----
    if constexpr (EPI == VV_EPI_QKV_ROPE) {
        // this wave's 128 features are ONE head: bias, rotate-half RoPE from the (cos, sin) table (the expressions of
        // vv_rope_append_kernel / vv_attn_fused_kernel: feature i pairs with i + 64 = accumulator tiles i and i + 4 of the same
        // lane), then q -> fp32 rows (scaled), k / v -> bf16 into the cache's tile layouts.  Replaces the fp32 qkv round trip
        // and the vv_rope_append launch of the prompt pass (modeling_vibevoice_inference.py:467-482 through HF Qwen2Attention).
        const VVRow rw = a.rows[0];
        const int hh = (ft0 * 16 + wf * 128) >> 7;
        if (hh >= a.Hq + 2 * a.Hkv) return;
        const int nb0 = hh * 128 + fq * 4;
        float4 bs[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] = a.bias ? *reinterpret_cast<const float4*>(a.bias + nb0 + i * 16) : float4{0.f, 0.f, 0.f, 0.f};
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = (tt0 + wr * 4 + j) * 16 + frow;
            if (t >= a.T) continue;
            const int pos = rw.pos + t;
            if (hh < a.Hq + a.Hkv) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2* tp = a.rope_tab + (int64_t)pos * 64 + i * 16 + fq * 4;
                    const float4 t0 = *reinterpret_cast<const float4*>(tp), t1 = *reinterpret_cast<const float4*>(tp + 2);
                    const float cs[4] = {t0.x, t0.z, t1.x, t1.z}, sn[4] = {t0.y, t0.w, t1.y, t1.w};
                    const float b1[4] = {bs[i].x, bs[i].y, bs[i].z, bs[i].w}, b2[4] = {bs[i + 4].x, bs[i + 4].y, bs[i + 4].z, bs[i + 4].w};
                    float o1[4], o2[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x1 = acc[i][j][r] + b1[r], x2 = acc[i + 4][j][r] + b2[r];
                        o1[r] = x1 * cs[r] - x2 * sn[r];
                        o2[r] = x2 * cs[r] + x1 * sn[r];
                    }
                    const int d = i * 16 + fq * 4;                    // rotation index = feature inside the head (first half)
                    if (hh < a.Hq) {
                        float* qp = a.q_out + ((int64_t)t * a.Hq + hh) * 128 + d;
                        *reinterpret_cast<float4*>(qp) = float4{o1[0] * a.q_scale, o1[1] * a.q_scale, o1[2] * a.q_scale, o1[3] * a.q_scale};
                        *reinterpret_cast<float4*>(qp + 64) = float4{o2[0] * a.q_scale, o2[1] * a.q_scale, o2[2] * a.q_scale, o2[3] * a.q_scale};
                    } else {
                        // K tile layout: element (pos, d) -> tile (pos >> 4) * 4 + (d >> 5), lane (pos & 15) + 16 * ((d & 31) >> 3), slot d & 7
                        __bf16* kb = a.kc + (int64_t)rw.cache * a.cache_stride + (int64_t)(hh - a.Hq) * a.head_stride;
#pragma unroll
                        for (int w = 0; w < 2; ++w) {
                            const int dd = d + w * 64;
                            bf16x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = (__bf16)(w ? o2[r] : o1[r]);
                            const int64_t tile = (int64_t)(pos >> 4) * 4 + (dd >> 5);
                            const int ln = (pos & 15) + 16 * ((dd & 31) >> 3);
                            *reinterpret_cast<uint2*>(kb + (tile * 64 + ln) * 8 + (dd & 7)) = __builtin_bit_cast(uint2, v);
                        }
                    }
                }
            } else {
                // V tile layout: element (pos, d) -> block pos >> 5, tile blk * 8 + (d >> 4), lane (d & 15) + 16 * ((pos & 15) >> 2),
                // slot ((pos >> 4) & 1) * 4 + (pos & 3)
                __bf16* vb = a.vc + (int64_t)rw.cache * a.cache_stride + (int64_t)(hh - a.Hq - a.Hkv) * a.head_stride;
                const int blk = pos >> 5, pp = pos & 15;
                const int jj = ((pos >> 4) & 1) * 4 + (pp & 3), q4 = pp >> 2;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float bb[4] = {bs[i].x, bs[i].y, bs[i].z, bs[i].w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int dd = i * 16 + fq * 4 + r;
                        const int64_t tile = (int64_t)blk * 8 + (dd >> 4);
                        vb[(tile * 64 + (dd & 15) + 16 * q4) * 8 + jj] = (__bf16)(acc[i][j][r] + bb[r]);
                    }
                }
            }
        }
        return;
    }
    if constexpr (DUAL) {
        const int KTo = (a.N + 31) >> 5;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ft = ft0 + wf * 4 + i;
            if (ft >= n_tiles) continue;
            const int n0 = ft * 16 + fq * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tt = tt0 + wr * 4 + j;
                if (tt >= t_tiles) continue;
                typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool live = (n0 + r < a.N) && (tt * 16 + frow < a.T);
                    o[r] = (__bf16)(live ? g3_silu(acc[i][j][r]) * acc[4 + i][j][r] : 0.f);
                }
                const int64_t tile = (int64_t)tt * KTo + (n0 >> 5);
                const int ol = frow + 16 * ((n0 & 31) >> 3);
                unsigned char* dst = reinterpret_cast<unsigned char*>(a.Yp) + ((tile * 64 + ol) * 16 + (n0 & 7) * 2);
                *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, o);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ft = ft0 + wf * 8 + i;
            if (ft >= n_tiles) continue;
            const int n0 = ft * 16 + fq * 4;
            if (n0 >= a.N) continue;
            float4 pb = {0.f, 0.f, 0.f, 0.f};
            if constexpr (EPI == VV_EPI_BIAS) {
                if (a.bias) pb = *reinterpret_cast<const float4*>(a.bias + n0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = (tt0 + wr * 4 + j) * 16 + frow;
                if (t >= a.T) continue;
                float* yp = a.Y + (int64_t)t * a.ldy + n0;
                float4 o = {acc[i][j][0] + pb.x, acc[i][j][1] + pb.y, acc[i][j][2] + pb.z, acc[i][j][3] + pb.w};
                if constexpr (EPI == VV_EPI_RESID) {
                    const float4 py = *reinterpret_cast<const float4*>(yp);
                    o.x += py.x; o.y += py.y; o.z += py.z; o.w += py.w;
                }
                *reinterpret_cast<float4*>(yp) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ GEMM, 256 x 256 tile
// The long-prompt form of vv_gemm3_kernel: a workgroup owns 256 features x 256 rows (8 waves as 2 x 4, 128 x 64 each: 32
// accumulators; SwiGLU: 128 gate + 128 up features -> 128 output features), K in 32-wide stages through a 4-slot ring of
// 32 KiB (16 A + 16 B fragments per slot, LDS-DMA).  The workgroup's two halves (waves 0-3 / 4-7: one wave of each per SIMD)
// run in ANTI-PHASE (MI355X_MICROARCH.md, "Two waves per SIMD"): while one half issues the 32 MFMAs of a stage from fragment
// registers, the other reads ITS fragments of the stage it computes next and issues its share of the DMA four stages ahead;
// one barrier per phase, two phases per stage:
//   phase 2q    half 0: MFMAs of stage q                      half 1: read fragments(q), DMA pieces of stage q-1+4
//   phase 2q+1  half 0: read fragments(q+1), DMA(q+4)         half 1: MFMAs of stage q
//   landing  at the end of every even phase each wave waits until at most 2 of its 4-piece stages are in flight (vmcnt),
//            i.e. stage q+1 has landed; the barrier then makes that true for all waves' pieces.
//   reuse    the slot of stage q is last read in phase 2q (half 1) and refilled in phase 2q+1 (half 0) / 2q+2 (half 1).
// DMA past the k range re-reads the last k-tile (keeps the vmcnt bookkeeping uniform; nobody reads those slots).
// Round 2's form of this kernel (one 64-wide stage per barrier, two 64 KiB buffers, all 8 waves issuing DMA -> fragment reads
// -> MFMAs in step) left the matrix pipe idle while both waves of a SIMD loaded; this schedule, bit-identical results, is
// 7-17 % faster per launch at T = 10,922 (tools/experiments/gemm_pingpong, profiles/r03_gemm_pingpong_ab*.json).  The same
// A/B measured three more schedules (fragment reads / DMA interleaved into the MFMA stream with one barrier per stage;
// register-staged fills instead of LDS-DMA) within a few % of each other and slower: under these GEMMs the package sits at
// its 1.4 kW limit with the shader clock at ~1.9 GHz (rocm-smi while the kernel loops), so what pays is activity removed per
// flop (the strip order of g4_map: 12 -> 4.4 GB of L2 misses per gate/up launch), not a denser issue pattern.
// a bare s_barrier that nothing is scheduled across (the phases ARE the schedule: MFMAs must stay between their two barriers)
__device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// explicit waits as the builtin (the compiler's own counter bookkeeping sees them; an asm wait is invisible to it and it then
// guards the first use of already-waited-for fragments with a full lgkmcnt(0)).  gfx9 encoding: vmcnt [3:0] + [15:14],
// expcnt [6:4], lgkmcnt [11:8]; a field of all ones = no wait on that counter.
template <int N> __device__ __forceinline__ void pp_vmcnt() {
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pp_lgkm0() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}

template <int EPI>
__global__ __launch_bounds__(512, 1) void vv_gemm4_kernel(const VVGemm3 a) {
    constexpr int NSLOT = 4;
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int FT = DUAL ? 8 : 16;
    constexpr int SLOT = 32 * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = (a.K + 31) >> 5;
    const int n_tiles = (a.N + 15) >> 4, t_tiles = (a.T + 15) >> 4;
    int nb, tb, part, slot0;
    if (!g4_map(a, nb, tb, part, slot0)) return;
    const int split = a.split;
    const int ft0 = nb * FT, tt0 = tb * 16;
    const int wf = wave & 1, wr = wave >> 1;
    const int half = wave >> 2;
    // k range, in the 64-wide steps vv_gemm4 splits by (both kernels cut a split tile at the same k)
    const int n_steps = (KT + 1) >> 1;
    const int s0 = part < 0 ? 0 : (int)((int64_t)part * n_steps / split);
    const int s1 = part < 0 ? n_steps : (int)((int64_t)(part + 1) * n_steps / split);
    const int kt0 = s0 * 2, kt1 = (s1 * 2 < KT) ? s1 * 2 : KT;
    const int n = kt1 - kt0;                       // stages (k-tiles) of this workgroup
    // loader: wave w copies pieces 4w .. 4w+3 of a stage's 32.  f < 16: A slot f (DUAL: 0-7 gate, 8-15 up); else B row tile f - 16
    const u32x4* src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = wave * 4 + i;
        if (f < 16) {
            const u32x4* base = (DUAL && f >= 8) ? a.W2 : a.W;
            int ft = ft0 + (DUAL ? (f & 7) : f);
            if (ft > n_tiles - 1) ft = n_tiles - 1;
            src[i] = base + ((int64_t)ft * KT + kt0) * 64 + lane;
        } else {
            int tt = tt0 + (f - 16);
            if (tt > t_tiles - 1) tt = t_tiles - 1;
            src[i] = a.Xp + ((int64_t)tt * KT + kt0) * 64 + lane;
        }
    }
    auto issue = [&](int g, int slot) {            // stage g (k-tile kt0 + g, clamped) into ring slot `slot`
        const int gg = g < n - 1 ? g : n - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(src[i] + (int64_t)gg * 64, ring + slot * SLOT + (wave * 4 + i) * 1024);
    };
    bf16x8 af[8], bfr[4];
    auto read = [&](int slot) {
        const unsigned char* st = ring + slot * SLOT;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int sl = DUAL ? ((i < 4) ? (wf * 4 + i) : (8 + wf * 4 + (i - 4))) : (wf * 8 + i);
            af[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + (sl * 64 + lane) * 16));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bfr[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(st + ((16 + wr * 4 + j) * 64 + lane) * 16));
    };
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    };
#pragma unroll
    for (int g = 0; g < NSLOT; ++g) issue(g, g);
    pp_vmcnt<(NSLOT - 1) * 4>();
    pp_barrier();                                                     // stage 0 has landed
    // two straight-line loops (same barrier count): a branch on the half around the MFMAs inside ONE loop makes the 128
    // accumulator registers phi values and the allocator spills them
    if (half == 0) {
        read(0);
        pp_lgkm0();
        int sq = 0;                                                   // ring slot of stage q
#pragma unroll 1
        for (int q = 0; q < n; ++q) {
            const int sq1 = (sq + 1 == NSLOT) ? 0 : sq + 1;
            compute();                                                // phase 2q
            pp_vmcnt<(NSLOT - 2) * 4>();
            pp_barrier();
            if (q + 1 < n) read(sq1);                                 // phase 2q + 1
            issue(q + NSLOT, sq);
            pp_lgkm0();
            pp_barrier();
            sq = sq1;
        }
    } else {
        int sq = 0;
#pragma unroll 1
        for (int q = 0; q < n; ++q) {
            const int sq1 = (sq + 1 == NSLOT) ? 0 : sq + 1;
            const int sqm = (sq == 0) ? NSLOT - 1 : sq - 1;           // slot of stage q - 1, refilled with stage q - 1 + NSLOT
            read(sq);                                                 // phase 2q
            if (q >= 1) issue(q - 1 + NSLOT, sqm);
            pp_lgkm0();
            pp_vmcnt<(NSLOT - 2) * 4>();
            pp_barrier();
            compute();                                                // phase 2q + 1
            pp_barrier();
            sq = sq1;
        }
    }
    pp_vmcnt<0>();                                                    // no DMA may land in an LDS allocation that has been released
    g4_finish<EPI>(a, acc, part, slot0, ft0, tt0, wf, wr);
}

// ------------------------------------------------------------------------------------------------ prefill attention
__device__ __forceinline__ float a3_xrow_max(float v) {
    // max over the 4 lanes that share (lane & 15): rows of 16 lanes exchanged pairwise, then the two halves of the wave.
    // NB: going through `unsigned` temporaries is deliberate -- bit-casting element 1 of the builtin's result directly reads
    // element 0 (clang); the per-row parity tests (tests/test_gpu_shipped.py, test_gpu_geometry.py) pin the exchange.
    unsigned x = __float_as_uint(v);
    auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
    unsigned r0 = r[0], r1 = r[1];
    v = fmaxf(__uint_as_float(r0), __uint_as_float(r1));
    x = __float_as_uint(v);
    auto t = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    unsigned t0 = t[0], t1 = t[1];
    return fmaxf(__uint_as_float(t0), __uint_as_float(t1));
}

// rows = consecutive positions of ONE cache (rows[0] first); q = the rotated queries (vv_rope_append_kernel), fp32 [R][Hq][D]; the
// softmax scale is applied here.  A workgroup = 4 units (64 query rows x one query head) of one kv head, 8 waves = unit x row half (two
// 16-row tiles per wave), the causal prefix streamed ONCE, 64 positions per stage through a 4-slot LDS ring (LDS-DMA).
// Round 2's form of this kernel (all waves in step: S^T MFMAs -> softmax -> P.V MFMAs between two barriers per stage) ran the
// matrix pipe 31-33 % busy: both waves of a SIMD contended for it, then both for the VALU.  Here the two waves of a SIMD (w and
// w + 4: one head's two row halves) alternate a MATRIX segment with a VALU segment, one barrier per phase (the schedule of
// vv_gemm4_kernel; MI355X_MICROARCH.md, "Two waves per SIMD"):
//   per wave and stage s:   V(s) = softmax of S(s) -> P(s)            M(s) = O += P(s).V(s), then S(s+1) = K(s+1) q^T
//   phase 2s    half 0: V(s), DMA pieces of stage s+2        half 1: M(s-1)
//   phase 2s+1  half 0: M(s)                                half 1: V(s), DMA pieces of stage s+3
//   ring     4 slots of one stage (K fragments, then V fragments); stage s is live from S(s) of half 0 (phase 2s-1) to the V(s)
//            fragment reads of half 1 (phase 2s+1); at the end of every even phase each wave waits until at most ONE of its
//            stages is in flight (vmcnt), which is "stage s+1 has landed" for both halves.
//   traffic  V(s) fragments are requested at the head of the VALU segment (they land under the softmax), each K(s+1) fragment
//            between the P.V MFMAs, into the registers of the V fragment just multiplied: no MFMA run starts with an LDS trip.
// and the softmax is cut to what the phase stamps showed it must be (the VALU segment was the longer one: ~1750 of a 4700-tick
// stage, tools/experiments/attn_pingpong):
//   * the S^T accumulators START at -m (m = the query row's reference exponent, in the MFMA's C operand), so the scores arrive
//     as S - m and are exponentiated as they are: no subtraction per score;
//   * LAZY rescaling: m is the row maximum of stage 0 and is raised only when a later stage exceeds it by more than 2^8 (a
//     wave-uniform vote per stage).  P is formed relative to m -- exp2(S - m) <= 256 cannot overflow, and O / l does not depend
//     on m -- so the common stage has no running-maximum update and no rescale of O at all;
//   * the row sums l = sum P go through the matrix pipe (4 MFMAs per stage against an all-ones A operand) instead of 32 VALU
//     adds per lane and a final cross-lane reduction.
// Copies past the last stage re-read it (uniform vmcnt bookkeeping); the ragged last stage is walked whole and masked.
// 7B prompt of 10,922 rows: 1205 -> 1035 us per layer (0.71 -> 0.83 PFLOP/s); what is left per stage: the matrix segment's own
// MFMAs (1088 cycles) + ~25 cycles of issue per LDS fragment read + two barriers, and whichever wave loses the SIMD's
// arbitration runs ~25 % longer (s_setprio only moves that to the other half).
template <int D>
__global__ __launch_bounds__(512) void vv_attn_prefill4_kernel(
    const float* __restrict__ q, const VVRow* __restrict__ rows, const __bf16* __restrict__ kc,
    const __bf16* __restrict__ vc, int R, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride,
    float* __restrict__ out, u32x4* __restrict__ out_packed) {
    constexpr int RT = 2;
    constexpr int KT = D / 32, DT = D / 16;
    constexpr int KF = 2 * KT;                       // K fragments of a 32-position block (2 position tiles x KT)
    constexpr int SF = 2 * (KF + DT);                // fragments of one 64-position stage: K of 2 blocks, then V of 2 blocks
    constexpr int BUF = SF * 1024;
    constexpr int NW = 8, NS = 4;
    constexpr int PW = SF / NW;                      // copies per wave and stage
    static_assert((2 * KF) % NW == 0 && (2 * DT) % NW == 0, "every wave copies the same number of K and of V fragments");
    static_assert(KF == DT, "K and V fragments of a block share one register set");
    extern __shared__ __attribute__((aligned(16))) unsigned char kv[];          // NS stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int half = wave >> 2;
    // The work of one kv head is n_rt x G units (64-row tile, query head of the group), tile-major; a workgroup takes FOUR
    // consecutive units (a wave pair each).  With 4 heads per workgroup taken from ONE tile a group of 7 left every eighth wave pair
    // idle (6: every fourth); packed, a workgroup may straddle two adjacent tiles (more when G < 4): it walks the longer
    // prefix and the earlier tile's rows see the extra stage fully masked (what the ragged diagonal stage already is for them).
    const VVRow rw = rows[0];
    const int G = Hq / Hkv;
    const int n_rt = (R + 63) >> 6, n_units = n_rt * G;
    const int b = (int)gridDim.x - 1 - (int)blockIdx.x;                         // longest workgroups first
    const int u = b * 4 + (wave & 3);                                           // this wave pair's unit
    const bool act = u < n_units;
    const int qt_lo = (b * 4) / G, qt_hi = min(b * 4 + 3, n_units - 1) / G;     // tiles this workgroup touches
    const int qt = act ? u / G : qt_hi;
    const int g = act ? u - qt * G : 0;                                         // this wave's query head inside the group
    const int r0 = qt * 64, kvh = blockIdx.y;
    const int rw0 = r0 + half * (RT * 16);           // first query row of this wave
    const int h = kvh * G + g;
    const int col = lane & 15, qg = lane >> 4;
    const int pend = rw.pos + min(qt_hi * 64 + 63, R - 1) + 1;        // positions this workgroup walks: [0, pend)
    const int n = (pend + 63) >> 6;                                   // stages
    const int first_masked = (rw.pos + qt_lo * 64) >> 6;              // stages below this one are visible to every query row
    const u32x4* kt_base = reinterpret_cast<const u32x4*>(kc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride);
    const u32x4* vt_base = reinterpret_cast<const u32x4*>(vc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride);
    constexpr float LOG2E = 1.4426950408889634f;

    const u32x4* ksrc = kt_base + wave * 64 + lane;
    const u32x4* vsrc = vt_base + wave * 64 + lane;
    auto issue = [&](int st) {                       // stage st (clamped) into slot st & 3: wave w moves K / V fragments w, w + 8, ..
        const int sc_ = st < n - 1 ? st : n - 1;
        const u32x4* ks = ksrc + (int64_t)sc_ * (2 * KF * 64);
        const u32x4* vs = vsrc + (int64_t)sc_ * (2 * DT * 64);
        unsigned char* kd = kv + (st & (NS - 1)) * BUF + wave * 1024;
#pragma unroll
        for (int i = 0; i < (2 * KF) / NW; ++i) glds16(ks + i * (NW * 64), kd + i * (NW * 1024));
#pragma unroll
        for (int i = 0; i < (2 * DT) / NW; ++i) glds16(vs + i * (NW * 64), kd + (2 * KF + i * NW) * 1024);
    };
    issue(0);
    issue(1);
    if (half == 1) issue(2);

    bf16x8 qf[RT][KT];
    int plim[RT];                                    // last position this lane's query row (column of S^T) may attend
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int row = min(rw0 + rt * 16 + col, R - 1);
        plim[rt] = rw.pos + row;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const float* qp = q + ((int64_t)row * Hq + h) * D + kt * 32 + qg * 8;
            const float4 a0 = *reinterpret_cast<const float4*>(qp), a1 = *reinterpret_cast<const float4*>(qp + 4);
            const float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[rt][kt][j] = (__bf16)(v[j] * LOG2E);
        }
    }
    f32x4 o[RT][DT], sc[RT][4];
    f32x4 ol[RT];                                    // row sums (every lane: its query row's)
    float negm[RT];                                  // -m (m = the row's reference exponent, log2 domain): the S^T accumulators' start value
    bf16x8 pb[RT][2];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        ol[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        negm[rt] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[rt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    bf16x8 fr[2][DT];                                // ONE set of fragment registers: V(s), then K(s+1)
    auto load_k1 = [&](int s, int blk, int f) {
        const unsigned char* cur = kv + (s & (NS - 1)) * BUF;
        fr[blk][f] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(cur + ((blk * KF + f) * 64 + lane) * 16));
    };
    auto load_v = [&](int s) {
        const unsigned char* vb_ = kv + (s & (NS - 1)) * BUF + 2 * KF * 1024;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                fr[blk][dt] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(vb_ + ((blk * DT + dt) * 64 + lane) * 16));
    };
    // ---- S^T - m = K q^T - m (K fragments in fr): sc[rt][2 blk + hf] = positions 64 s + 32 blk + 16 hf + 4 qg + r, query row = lane & 15 ----
    auto qk_mfma = [&]() {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int i = 0; i < 4; ++i) sc[rt][i] = f32x4{negm[rt], negm[rt], negm[rt], negm[rt]};
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    sc[rt][2 * blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[blk][kt], qf[rt][kt], sc[rt][2 * blk], 0, 0, 0);
                    sc[rt][2 * blk + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[blk][KT + kt], qf[rt][kt], sc[rt][2 * blk + 1], 0, 0, 0);
                }
    };
    // the matrix segment: O += P . V of the stage whose V fragments are in fr, each V fragment's registers refilled with a K
    // fragment of stage sk right after its two MFMAs, l += sum P (all-ones A operand), then S^T of stage sk
    auto matrix = [&](bool do_pv, int sk) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (do_pv) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        o[rt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[blk][dt], pb[rt][blk], o[rt][dt], 0, 0, 0);
                }
                if (sk >= 0) load_k1(sk, blk, dt);
            }
        if (do_pv) {
            u32x4 one4 = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};      // eight bf16 1.0; rebuilt per stage: registers
            asm volatile("" : "+v"(one4));                                        // are scarcer here than 4 v_mov
            const bf16x8 ones = __builtin_bit_cast(bf16x8, one4);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) ol[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[rt][blk], ol[rt], 0, 0, 0);
        }
        if (sk >= 0) qk_mfma();
    };
    // ---- the softmax step of stage s: sc (= S - m) -> pb; on the rare re-base also m, O, l ----
    auto softmax = [&](int s) {
        const int p0 = s * 64;
        const bool fast = s < first_masked;          // the whole stage lies below the causal diagonal of every query row
        float mx[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            if (!fast) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        if (p0 + i * 16 + qg * 4 + rr > plim[rt]) sc[rt][i][rr] = -INFINITY;
            }
            float v = sc[rt][0][0];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) v = fmaxf(v, sc[rt][i][rr]);
            mx[rt] = a3_xrow_max(v);
        }
        const bool rebase = (s == 0) || __builtin_amdgcn_ballot_w64(mx[0] > 8.0f || mx[1] > 8.0f) != 0;   // wave-uniform
        if (!rebase) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        pb[rt][i >> 1][(i & 1) * 4 + rr] = (__bf16)__builtin_amdgcn_exp2f(sc[rt][i][rr]);
        } else {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                // stage 0: the reference becomes the stage maximum (finite: position 0 is visible to every row); later: rows
                // above the threshold move up to their stage maximum, the others (and rows whose stage is all masked) stay
                const float dl = (s == 0) ? mx[rt] : (mx[rt] > 8.0f ? mx[rt] : 0.f);
                const float alpha = __builtin_amdgcn_exp2f(-dl);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        pb[rt][i >> 1][(i & 1) * 4 + rr] = (__bf16)__builtin_amdgcn_exp2f(sc[rt][i][rr] - dl);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) o[rt][dt] *= alpha;
                ol[rt] *= alpha;
                negm[rt] -= dl;
            }
        }
    };
    // two straight-line loops with the same barrier count (a branch on the half inside one loop makes the accumulators phi values)
    if (half == 0) {
        pp_vmcnt<PW>();                                               // stages 0, 1 in flight: stage 0 has landed
        pp_barrier();
        if (act) matrix(false, 0);                                    // phase -1: S(0)
        pp_lgkm0();
        pp_barrier();
#pragma unroll 1
        for (int s = 0; s < n; ++s) {
            if (act) { load_v(s); softmax(s); }                       // phase 2s
            issue(s + 2);
            pp_lgkm0();
            pp_vmcnt<PW>();
            pp_barrier();
            if (act) matrix(true, s + 1 < n ? s + 1 : -1);            // phase 2s + 1: O += P(s).V(s), S(s+1)
            pp_lgkm0();
            pp_barrier();
        }
    } else {
        pp_vmcnt<2 * PW>();                                           // stages 0, 1, 2 in flight
        pp_barrier();
        pp_barrier();                                                 // phase -1: nothing to do yet
#pragma unroll 1
        for (int s = 0; s < n; ++s) {
            if (act) matrix(s >= 1, s);                               // phase 2s: O += P(s-1).V(s-1), S(s)
            pp_lgkm0();
            pp_vmcnt<PW>();
            pp_barrier();
            if (act) { load_v(s); softmax(s); }                       // phase 2s + 1
            issue(s + 3);
            pp_lgkm0();
            pp_barrier();
        }
        if (act) matrix(true, -1);                                    // phase 2n: V(n-1) is in registers
    }
    pp_vmcnt<0>();                                                    // no copy may land in an LDS allocation that has been released
    if (act) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const int row = rw0 + rt * 16 + col;
            if (row < R) {
                const float inv = 1.0f / ol[rt][0];
                if (out_packed) {
                    // straight into the o-projection's B operand (vv_pack_rows_kernel's layout, K = Hq * D): element (t, k) lives in
                    // tile (t >> 4, k >> 5), lane (t & 15) + 16 * ((k & 31) >> 3), slot k & 7; this lane holds 4 consecutive k
                    const int KTo = (Hq * D) >> 5;
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const int k0 = h * D + dt * 16 + qg * 4;
                        bf16x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = (__bf16)(o[rt][dt][r] * inv);
                        const int64_t tile = (int64_t)(row >> 4) * KTo + (k0 >> 5);
                        const int ol_ = (row & 15) + 16 * ((k0 & 31) >> 3);
                        unsigned char* dst = reinterpret_cast<unsigned char*>(out_packed) + ((tile * 64 + ol_) * 16 + (k0 & 7) * 2);
                        *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, v);
                    }
                } else {
                    float* orow = out + ((int64_t)row * Hq + h) * D + qg * 4;
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        *reinterpret_cast<float4*>(orow + dt * 16) =
                            float4{o[rt][dt][0] * inv, o[rt][dt][1] * inv, o[rt][dt][2] * inv, o[rt][dt][3] * inv};
                }
            }
        }
    }
}

}  // namespace

// the K-split of the partial last round (see vv_gemm3_launch): fills a.full_idx / a.split / workspace pointers, updates the grid size
static void g4_split_plan(VVGemm3& a, const VVGemmWs* ws, int K, unsigned& n_wgs) {
    if (!(ws && ws->partials && ws->flags && ws->err)) return;
    const int total = a.n_blocks * a.t_blocks, q = total >> 3, r = total & 7;
    const int full_idx = (total / 256) * 32;
    const int max_rem = q + (r ? 1 : 0) - full_idx;           // 0..32 tiles per XCD in the partial round
    const int n_steps = ((K + 31) / 32 + 1) / 2;
    int split = max_rem > 0 ? 32 / max_rem : 1;
    if (split > 8) split = 8;
    if (split > n_steps / 8) split = n_steps / 8;             // a part keeps >= 8 of the 64-wide steps
    if (split == 2 && n_steps < 256) split = 1;               // halves of a short K do not pay (see vv_gemm3_launch)
    if (split > 1) {
        a.full_idx = full_idx; a.split = split; a.ws = ws->partials; a.flags = ws->flags; a.err = ws->err;
        n_wgs = (unsigned)(8 * full_idx + 8 * max_rem * split);
    }
}

extern "C" {

int vv_pack_rows_launch(const float* x, int ldx, const float* nw, float eps, void* xp, int T, int K, hipStream_t s) {
    if ((K & 7) || (ldx & 3) || (((uintptr_t)x) & 15) || (((uintptr_t)xp) & 15) || (nw && (((uintptr_t)nw) & 15))) return -1;
    hipLaunchKernelGGL(vv_pack_rows_kernel, dim3((T + 15) / 16), dim3(256), 0, s, x, ldx, nw, eps, (u32x4*)xp, T, K);
    return vv_launch_rc(0);
}

int vv_unpack_rows_launch(const void* xp, float* x, int T, int K, hipStream_t s) {
    const int64_t n = (int64_t)T * K;
    hipLaunchKernelGGL(vv_unpack_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const __bf16*)xp, x, T, K);
    return vv_launch_rc(0);
}

// Y (op)= Xp . W^T.  epi: VV_EPI_STORE / BIAS / RESID -> fp32 Y[T][ldy];  VV_EPI_SWIGLU -> packed bf16 Yp [T][N] (W = gate, W2 = up)
int vv_ada_pack_launch(const float* cproj, const float* temb, void* xp, int rows, int n_steps, int H, hipStream_t s) {
    if ((H & 7) || rows < 1 || n_steps < 1) return -1;
    const int T = rows * n_steps, KT = (H + 31) >> 5;
    hipLaunchKernelGGL(vv_ada_pack_kernel, dim3((T + 15) / 16, (KT + 3) / 4), dim3(256), 0, s, cproj, temb, (u32x4*)xp, rows, n_steps, H);
    return vv_launch_rc(0);
}

int vv_gemm3_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, int T, int N, int K,
                    int ldy, int epi, const VVGemmWs* ws, hipStream_t s) {
    if (T < 1 || N < 1 || K < 32 || (N & 3)) return -1;
    VVGemm3 a;
    a.W = (const u32x4*)W; a.W2 = (const u32x4*)W2; a.Xp = (const u32x4*)Xp; a.Y = Y; a.Yp = (u32x4*)Yp; a.bias = bias;
    a.T = T; a.N = N; a.K = K; a.ldy = ldy;
    a.full_idx = 0; a.split = 1; a.ws = nullptr; a.flags = nullptr; a.err = nullptr; a.sfw = 4;
    const int n_tiles = (N + 15) / 16;
    // long prompts: the 256 x 256 kernel whenever its grid fills the chip at least once; smaller problems keep the 128-feature
    // kernel, whose tiles quantise better.  Per launch at T = 10,922 (7B layer), 128-feature kernel -> round-2 256 x 256 ->
    // round-3 schedule + strip order + K-split: gate/up 3188 -> 2650 -> 2350 us, down 1751 -> 1470 -> 1390, qkv 518 -> 404 -> 315,
    // o 432 -> 340 -> 271.
    const int64_t wgs4 = (int64_t)((n_tiles + ((epi == VV_EPI_SWIGLU) ? 8 : 16) - 1) / ((epi == VV_EPI_SWIGLU) ? 8 : 16)) * ((T + 255) / 256);
    if (wgs4 >= 256 &&
        (epi == VV_EPI_SWIGLU || epi == VV_EPI_RESID || epi == VV_EPI_BIAS || epi == VV_EPI_STORE)) {
        const int ft4 = (epi == VV_EPI_SWIGLU) ? 8 : 16;
        a.n_blocks = (n_tiles + ft4 - 1) / ft4;
        a.t_blocks = (T + 255) / 256;
        if (epi == VV_EPI_SWIGLU && (!W2 || !Yp)) return -1;
        if (epi != VV_EPI_SWIGLU && (!Y || (ldy & 3))) return -1;
        // One workgroup per CU (128 KiB of LDS): the grid runs in rounds of 256 and the tiles past the last whole round (T =
        // 10,922 at 7B widths: 602 tiles for down / o = 2.35 rounds, 774 for QKV = 3.02) cost most of a round however few they
        // are.  Each XCD's share of that remainder (<= 32 tiles) is split along K by the largest factor that still fits one
        // round (g4_map / g4_finish).  Measured (tools/experiments/gemm_pingpong): it pays when the remainder is a handful of
        // tiles (QKV: split 7, -6 %) or the parts stay long (down: K = 18,944 halved, -3.5 %); halving K = 3584 tiles does not
        // (o: +6 %; gate/up: neutral) -- a partial round is not a lost round on this chip: the package is at its power limit
        // under these kernels and the clock rises when fewer CUs compute -- so those launches stay whole.
        unsigned n_wgs = (unsigned)(a.n_blocks * a.t_blocks);
        g4_split_plan(a, ws, K, n_wgs);
        const dim3 grid4(n_wgs);
        static bool attr4 = false;
        if (!attr4) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm4_kernel<VV_EPI_SWIGLU>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm4_kernel<VV_EPI_RESID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm4_kernel<VV_EPI_BIAS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr4 = true;
        }
        const size_t smem4 = 4 * 32 * 1024;
        if (epi == VV_EPI_SWIGLU) hipLaunchKernelGGL((vv_gemm4_kernel<VV_EPI_SWIGLU>), grid4, dim3(512), smem4, s, a);
        else if (epi == VV_EPI_RESID) hipLaunchKernelGGL((vv_gemm4_kernel<VV_EPI_RESID>), grid4, dim3(512), smem4, s, a);
        else hipLaunchKernelGGL((vv_gemm4_kernel<VV_EPI_BIAS>), grid4, dim3(512), smem4, s, a);
        return vv_launch_rc(0);
    }
    const int ft = (epi == VV_EPI_SWIGLU) ? 4 : 8;
    a.n_blocks = (n_tiles + ft - 1) / ft;
    // 256-row workgroups once the problem is tall enough to fill the chip with them
    const int tr = (int64_t)a.n_blocks * ((T + 255) / 256) >= 768 ? 8 : 4;
    a.t_blocks = (T + 32 * tr - 1) / (32 * tr);
    dim3 grid((unsigned)(a.n_blocks * a.t_blocks));
    // Short prompts (a 330-token request: 36 tiles for the o / down projections of a 1.5B layer, 48 for QKV): a few dozen workgroups
    // stream the whole matrix -- 170 us for the 27.5 MB down projection.  K is split over grid.y (each part >= 4 of the 64-wide
    // steps) until ~192 workgroups pull on HBM; the parts go to dense fp32 tensors and vv_g3_reduce_kernel sums them in part order
    // with the bias / residual: deterministic, no hand-off inside a launch.  VVHIP_G3_KSPLIT=0 switches it off.
    int ks = 1;
    if (epi != VV_EPI_SWIGLU && ws && ws->g3_partials && grid.x < 128 && (N & 3) == 0 && Y) {
        static int on = -1;
        if (on < 0) { const char* e = getenv("VVHIP_G3_KSPLIT"); on = (e && e[0] == '0') ? 0 : 1; }
        const int n_steps = (((K + 31) >> 5) + 1) >> 1;
        ks = on ? (int)((192 + grid.x - 1) / grid.x) : 1;
        if (ks > 8) ks = 8;
        if (ks > n_steps / 4) ks = n_steps / 4;
        if (ks < 2 || (size_t)ks * T * N * 4 > ws->g3_bytes) ks = 1;
    }
    const float* bias_r = bias; const int resid_r = (epi == VV_EPI_RESID) ? 1 : 0;
    if (ks > 1) { grid.y = (unsigned)ks; a.ws = ws->g3_partials; a.split = ks; }
    static int db_on = -1;
    if (db_on < 0) { const char* e = getenv("VVHIP_G3_DB"); db_on = (e && e[0] == '0') ? 0 : 1; }
    const bool db = db_on && tr == 4 && (int64_t)grid.x * (ks > 1 ? ks : 1) <= 512;      // at most ~2 workgroups per CU: nothing else hides a stage's latency
#define VV_G3(E_) do { if (tr == 8) hipLaunchKernelGGL((vv_gemm3_kernel<E_, 8>), grid, dim3(256), 0, s, a); \
                       else if (db) hipLaunchKernelGGL((vv_gemm3_kernel<E_, 4, 1>), grid, dim3(256), 0, s, a); \
                       else hipLaunchKernelGGL((vv_gemm3_kernel<E_, 4>), grid, dim3(256), 0, s, a); } while (0)
    if (epi == VV_EPI_SWIGLU) {
        if (!W2 || !Yp) return -1;
        VV_G3(VV_EPI_SWIGLU);
    } else if (epi == VV_EPI_RESID) {
        if (!Y || (ldy & 3)) return -1;
        VV_G3(VV_EPI_RESID);
    } else if (epi == VV_EPI_BIAS || epi == VV_EPI_STORE) {
        if (!Y || (ldy & 3)) return -1;
        VV_G3(VV_EPI_BIAS);                        // bias == null: plain store
    } else {
        return -3;
    }
#undef VV_G3
    if (ks > 1) {
        const int64_t n4 = ((int64_t)T * N + 3) / 4;
        hipLaunchKernelGGL(vv_g3_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, ws->g3_partials, ks, T, N, bias_r, Y, ldy, resid_r);
    }
    return vv_launch_rc(0);
}

// causal prefill attention of R consecutive rows of one cache (64 query rows x the GQA group of one kv head per workgroup)
// The QKV projection of a prompt pass with RoPE + KV-cache append in the epilogue (head_dim 128, vv_gemm4 shapes only).
// Returns 1 when the fused kernel was launched, 0 when the shape does not qualify (the caller runs the plain GEMM + vv_rope_append),
// < 0 on error.  rows_dev[0] = (cache, first position): read on the device, so a captured graph replays at other positions.
int vv_gemm_qkv_rope_launch(const void* W, const void* Xp, const float* bias, int T, int K, int D, int Hq, int Hkv, const VVRow* rows_dev,
                            const void* rope_tab, float* q_out, void* kc, void* vc, int64_t cache_stride, int64_t head_stride,
                            const VVGemmWs* ws, hipStream_t s) {
    const int N = (Hq + 2 * Hkv) * D;
    if (D != 128 || T < 1 || K < 32 || !rows_dev || !rope_tab || !q_out || !kc || !vc) return 0;
    const int n_blocks = N / 256, t_blocks = (T + 255) / 256;
    if ((N & 255) || (int64_t)n_blocks * t_blocks < 256) return 0;
    VVGemm3 a;
    memset(&a, 0, sizeof(a));
    a.W = (const u32x4*)W; a.Xp = (const u32x4*)Xp; a.bias = bias;
    a.T = T; a.N = N; a.K = K; a.ldy = N;
    a.n_blocks = n_blocks; a.t_blocks = t_blocks; a.sfw = 4; a.split = 1;
    a.rows = rows_dev; a.rope_tab = (const float2*)rope_tab; a.q_out = q_out; a.kc = (__bf16*)kc; a.vc = (__bf16*)vc;
    a.cache_stride = cache_stride; a.head_stride = head_stride; a.Hq = Hq; a.Hkv = Hkv; a.q_scale = 1.0f / sqrtf((float)D);
    unsigned n_wgs = (unsigned)(n_blocks * t_blocks);
    g4_split_plan(a, ws, K, n_wgs);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_gemm4_kernel<VV_EPI_QKV_ROPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    hipLaunchKernelGGL((vv_gemm4_kernel<VV_EPI_QKV_ROPE>), dim3(n_wgs), dim3(512), (size_t)4 * 32 * 1024, s, a);
    return vv_launch_rc(1);
}

// out_packed != null: the result goes out as packed bf16 B fragments [R][Hq * D] (the o-projection GEMM's operand) instead of fp32 rows
int vv_attn_prefill4_launch(int D, const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq, int Hkv,
                            int64_t cache_stride, int64_t head_stride, float* out, void* out_packed, hipStream_t s) {
    if (Hq % Hkv != 0 || (D != 128 && D != 64)) return -1;
    if (out_packed && ((Hq * D) & 31)) return -1;
    const int G = Hq / Hkv;
    const dim3 grid((unsigned)((((int64_t)(R + 63) / 64) * G + 3) / 4), Hkv, 1);     // four (row tile, query head) units per workgroup
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_attn_prefill4_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_attn_prefill4_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const int sm = 4 * 2 * (2 * (D / 32) + D / 16) * 1024;          // 4 stages of K + V fragments
    if (D == 128) hipLaunchKernelGGL((vv_attn_prefill4_kernel<128>), grid, dim3(512), sm, s, q, rows, (const __bf16*)kc, (const __bf16*)vc, R, Hq, Hkv, cache_stride, head_stride, out, (u32x4*)out_packed);
    else hipLaunchKernelGGL((vv_attn_prefill4_kernel<64>), grid, dim3(512), sm, s, q, rows, (const __bf16*)kc, (const __bf16*)vc, R, Hq, Hkv, cache_stride, head_stride, out, (u32x4*)out_packed);
    return vv_launch_rc(0);
}

}  // extern "C"
