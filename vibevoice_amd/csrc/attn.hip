// attn.hip -- KV-cached GQA attention for the LM decode step (and chunked prefill).
//
// KV-cache layout (ours to choose; chosen so every wave-level K/V load is one
// coalesced 1 KiB MFMA fragment, no LDS transposes, no cross-lane shuffles):
//   K: per (cache, layer, kv_head): tiles [pos/16][d/32] of 16 positions x 32 dims;
//      lane l of a tile holds K[pos%16 = l&15][d%32 = (l>>4)*8 + 0..7]          (A operand of S^T = K q^T)
//   V: stored TRANSPOSED per 32-position block: tiles [pos/32][d/16] of 16 dims x 32 slots;
//      lane l holds V^T[d%16 = l&15][slot = (l>>4)*8 + 0..7] where, inside a block,
//      slot 8q+r   <-> position 4q+r      (r<4)
//      slot 8q+4+r <-> position 16+4q+r
//      -- exactly the order in which the two S^T accumulator tiles of that block sit
//      in a lane's registers, so P feeds the P.V MFMA as the B operand unshuffled.
//
// Decode attention is split along the sequence (flash-decoding): grid =
// (splits, kv_heads, rows); each block's 4 waves walk interleaved 32-position
// blocks with an online softmax per query head (query heads of the GQA group are
// the 16 MFMA columns, so K/V are read once per group); partial (m, l, O) go to a
// workspace and a second tiny kernel merges the splits.
#include "vv_common.h"

namespace {

template <int XS>
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8 (&out)[XS]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        __bf16 h = (__bf16)v[j];
        out[0][j] = h;
        if constexpr (XS > 1) {
            float r = v[j] - (float)h;
            __bf16 m = (__bf16)r;
            out[1][j] = m;
            if constexpr (XS > 2) out[2][j] = (__bf16)(r - (float)m);
        }
    }
}

// qkv: [R][(Hq + 2 Hkv) * D] fp32 (bias already added).  One wave per (row, head).
template <int D>
__global__ __launch_bounds__(64) void vv_rope_append_kernel(
    const float* __restrict__ qkv, const VVRow* __restrict__ rows, const float* __restrict__ inv_freq,
    float* __restrict__ q_out, __bf16* __restrict__ kc, __bf16* __restrict__ vc,
    int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, float q_scale) {
    const int r = blockIdx.x, h = blockIdx.y, lane = threadIdx.x;
    const VVRow rw = rows[r];
    const int pos = rw.pos;
    const float* src = qkv + ((int64_t)r * (Hq + 2 * Hkv) + h) * D;
    constexpr int HALF = D / 2;
    if (h < Hq + Hkv) {
        // rotate-half RoPE; cos/sin evaluated in fp32 from the host-provided inv_freq table
        for (int i = lane; i < HALF; i += 64) {
            const float ang = (float)pos * inv_freq[i];
            const float c = cosf(ang), s = sinf(ang);
            const float x1 = src[i], x2 = src[i + HALF];
            const float o1 = x1 * c - x2 * s;
            const float o2 = x2 * c + x1 * s;
            if (h < Hq) {
                float* q = q_out + ((int64_t)r * Hq + h) * D;
                q[i] = o1 * q_scale;
                q[i + HALF] = o2 * q_scale;
            } else {
                __bf16* kb = kc + (int64_t)rw.cache * cache_stride + (int64_t)(h - Hq) * head_stride;
                const int pt = pos >> 4, pl = pos & 15;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int d = w ? i + HALF : i;
                    const float v = w ? o2 : o1;
                    const int64_t tile = (int64_t)pt * (D / 32) + (d >> 5);
                    const int ln = pl + 16 * ((d & 31) >> 3);
                    kb[(tile * 64 + ln) * 8 + (d & 7)] = (__bf16)v;
                }
            }
        }
    } else {
        __bf16* vb = vc + (int64_t)rw.cache * cache_stride + (int64_t)(h - Hq - Hkv) * head_stride;
        const int blk = pos >> 5, p = pos & 31;
        const int half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
        const int j = half * 4 + rr;
        for (int d = lane; d < D; d += 64) {
            const int64_t tile = (int64_t)blk * (D / 16) + (d >> 4);
            const int ln = (d & 15) + 16 * q4;
            vb[(tile * 64 + ln) * 8 + j] = (__bf16)src[d];
        }
    }
}

template <int D, int XS>
__global__ __launch_bounds__(256) void vv_attn_split_kernel(
    const float* __restrict__ q, const VVRow* __restrict__ rows, const __bf16* __restrict__ kc,
    const __bf16* __restrict__ vc, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride,
    float* __restrict__ part_m, float* __restrict__ part_l, float* __restrict__ part_o) {
    constexpr int KT = D / 32;     // k-steps of the QK^T contraction
    constexpr int DT = D / 16;     // 16-dim output tiles of P.V
    const int S = gridDim.x;
    const int split = blockIdx.x, kvh = blockIdx.y, r = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const VVRow rw = rows[r];
    const int len = rw.pos + 1;
    const int G = Hq / Hkv;
    const int g = lane & 15;
    const int qg = lane >> 4;
    int chunk = (len + S - 1) / S;
    chunk = (chunk + 127) & ~127;
    const int start = split * chunk;
    const int end = min(len, start + chunk);
    if (start >= len) return;                   // whole block: the merge kernel never reads unused splits

    const u32x4* kt_base = reinterpret_cast<const u32x4*>(kc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride);
    const u32x4* vt_base = reinterpret_cast<const u32x4*>(vc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride);

    // q as B operand: lane holds q[g][kt*32 + qg*8 + j]
    bf16x8 qf[KT][XS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        float v[8];
        const float* qp = q + ((int64_t)r * Hq + kvh * G + g) * D + kt * 32 + qg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (g < G) ? qp[j] : 0.f;
        split8<XS>(v, qf[kt]);
    }

    float m = -INFINITY, lsum = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int p0 = start + wave * 32; p0 < end; p0 += 128) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        const int64_t t0 = (int64_t)(p0 >> 4) * KT;
        u32x4 ka[KT], kb[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            ka[kt] = kt_base[(t0 + kt) * 64 + lane];
            kb[kt] = kt_base[(t0 + KT + kt) * 64 + lane];
        }
        u32x4 vt[DT];
        const int64_t vt0 = (int64_t)(p0 >> 5) * DT;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vt[dt] = vt_base[(vt0 + dt) * 64 + lane];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int p = 0; p < XS; ++p) {
                s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ka[kt]), qf[kt][p], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, kb[kt]), qf[kt][p], s1, 0, 0, 0);
            }
        }
        float sv[8];
        float mx = -INFINITY;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int pa = p0 + qg * 4 + rr;
            sv[rr] = (pa < end) ? s0[rr] : -INFINITY;
            sv[4 + rr] = (pa + 16 < end) ? s1[rr] : -INFINITY;
            mx = fmaxf(mx, fmaxf(sv[rr], sv[4 + rr]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m, mx);               // finite: at least one position of this block is valid
        const float alpha = (m == -INFINITY) ? 0.f : expf(m - mn);
        float pv[8];
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            pv[j] = (sv[j] == -INFINITY) ? 0.f : expf(sv[j] - mn);
            ps += pv[j];
        }
        lsum = lsum * alpha + ps;
        m = mn;
        bf16x8 pb[XS];
        split8<XS>(pv, pb);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            o[dt] *= alpha;
#pragma unroll
            for (int p = 0; p < XS; ++p)
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vt[dt]), pb[p], o[dt], 0, 0, 0);
        }
    }
    // lane-group partial sums of l -> full per-head sum
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);

    // ---- combine the 4 waves (fixed order) ----
    __shared__ float sm[4][16], sl[4][16];
    __shared__ f32x4 so[4][DT][64];
    if (lane < 16) { sm[wave][lane] = m; sl[wave][lane] = lsum; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) so[wave][dt][lane] = o[dt];
    __syncthreads();
    if (wave != 0) return;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, sm[w][g]);
    float L = 0.f;
    f32x4 O[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float mw = sm[w][g];
        const float f = (mw == -INFINITY) ? 0.f : expf(mw - M);
        L += sl[w][g] * f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) O[dt] += so[w][dt][lane] * f;
    }
    const int64_t pidx = ((int64_t)r * Hkv + kvh) * S + split;
    if (lane < 16) { part_m[pidx * 16 + lane] = M; part_l[pidx * 16 + lane] = L; }
    if (g < G) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            float4 v = {O[dt][0], O[dt][1], O[dt][2], O[dt][3]};
            *reinterpret_cast<float4*>(part_o + (pidx * 16 + g) * D + dt * 16 + qg * 4) = v;
        }
    }
}

// grid (R, Hq), block D threads
template <int D>
__global__ void vv_attn_merge_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                     const float* __restrict__ part_o, const VVRow* __restrict__ rows,
                                     float* __restrict__ out, int Hq, int Hkv, int S) {
    const int r = blockIdx.x, h = blockIdx.y, d = threadIdx.x;
    const int G = Hq / Hkv;
    const int kvh = h / G, g = h - kvh * G;
    const int64_t base = ((int64_t)r * Hkv + kvh) * S;
    // same chunking as the split kernel: only the first `used` splits hold data
    const int len = rows[r].pos + 1;
    int chunk = (len + S - 1) / S;
    chunk = (chunk + 127) & ~127;
    const int used = (len + chunk - 1) / chunk;
    float M = -INFINITY;
    for (int s = 0; s < used; ++s) M = fmaxf(M, part_m[(base + s) * 16 + g]);
    float L = 0.f, acc = 0.f;
    for (int s = 0; s < used; ++s) {
        const float ms = part_m[(base + s) * 16 + g];
        if (ms == -INFINITY) continue;
        const float f = expf(ms - M);
        L += part_l[(base + s) * 16 + g] * f;
        acc += part_o[((base + s) * 16 + g) * D + d] * f;
    }
    out[((int64_t)r * Hq + h) * D + d] = acc / L;
}

}  // namespace

extern "C" int vv_rope_append_launch(int D, const float* qkv, const VVRow* rows, const float* inv_freq, float* q_out,
                                     void* kc, void* vc, int R, int Hq, int Hkv, int64_t cache_stride,
                                     int64_t head_stride, hipStream_t s) {
    dim3 grid(R, Hq + 2 * Hkv);
    const float scale = 1.0f / sqrtf((float)D);
    if (D == 128)
        hipLaunchKernelGGL((vv_rope_append_kernel<128>), grid, dim3(64), 0, s, qkv, rows, inv_freq, q_out,
                           (__bf16*)kc, (__bf16*)vc, Hq, Hkv, cache_stride, head_stride, scale);
    else if (D == 64)
        hipLaunchKernelGGL((vv_rope_append_kernel<64>), grid, dim3(64), 0, s, qkv, rows, inv_freq, q_out,
                           (__bf16*)kc, (__bf16*)vc, Hq, Hkv, cache_stride, head_stride, scale);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <int D, int XS>
static void attn_go(const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq, int Hkv,
                    int64_t cs, int64_t hs, int S, float* pm, float* pl, float* po, float* out, hipStream_t s) {
    hipLaunchKernelGGL((vv_attn_split_kernel<D, XS>), dim3(S, Hkv, R), dim3(256), 0, s, q, rows,
                       (const __bf16*)kc, (const __bf16*)vc, Hq, Hkv, cs, hs, pm, pl, po);
    hipLaunchKernelGGL((vv_attn_merge_kernel<D>), dim3(R, Hq), dim3(D), 0, s, pm, pl, po, rows, out, Hq, Hkv, S);
}

extern "C" int vv_attn_launch(int D, int xs, const float* q, const VVRow* rows, const void* kc, const void* vc,
                              int R, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, int S,
                              float* pm, float* pl, float* po, float* out, hipStream_t s) {
    if (Hq % Hkv != 0 || Hq / Hkv > 16) return -1;
#define VV_A(D_)                                                                                              \
    do {                                                                                                      \
        if (xs == 1) attn_go<D_, 1>(q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, S, pm, pl, po, out, s); \
        else if (xs == 2) attn_go<D_, 2>(q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, S, pm, pl, po, out, s); \
        else attn_go<D_, 3>(q, rows, kc, vc, R, Hq, Hkv, cache_stride, head_stride, S, pm, pl, po, out, s);   \
    } while (0)
    if (D == 128) VV_A(128);
    else if (D == 64) VV_A(64);
    else return -1;
#undef VV_A
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
