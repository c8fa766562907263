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
#include <cstdlib>
#include "vv_common.h"

namespace {

// positions per split: at least 512 (below that one workgroup walking the sequence beats a cross-workgroup merge),
// a multiple of 128 (4 waves x 32), and few enough splits to fit the grid
__device__ __forceinline__ int attn_chunk(int len, int S, int gran = 128) {
    int chunk = (len + S - 1) / S;
    chunk = (chunk + gran - 1) / gran * gran;          // a multiple of (waves x 32): gran is 128 or 256, 512 is a multiple of both
    return chunk < 512 ? 512 : chunk;
}

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

// The V^T fragments of a sequence's ragged last 32-position block: slots of positions >= end hold whatever the cache held (an
// earlier utterance's values, imported data, a profiling replay's output).  Their softmax weight is exactly 0, but 0 x NaN is NaN:
// zero them.  Slot j of lane group qg is position p0 + 4 qg + (j & 3) + 16 (j >> 2) (the layout at the top of this file).
template <int DT>
__device__ __forceinline__ void vv_zero_v_past_end(u32x4 (&vt)[DT], int p0, int qg, int end) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        bf16x8 t = __builtin_bit_cast(bf16x8, vt[dt]);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (p0 + qg * 4 + (j & 3) + ((j >> 2) << 4) >= end) t[j] = (__bf16)0.0f;
        vt[dt] = __builtin_bit_cast(u32x4, t);
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
    const int chunk = attn_chunk(len, S);
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
        if (p0 + 32 > end) vv_zero_v_past_end<DT>(vt, p0, qg, end);
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

// RoPE table: tab[pos][i] = (cos, sin)(pos * inv_freq[i]); same fp32 expressions as vv_rope_append_kernel.
__global__ void vv_rope_table_kernel(const float* __restrict__ inv_freq, float2* __restrict__ tab, int n_pos, int half) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n_pos * half) return;
    const int pos = (int)(e / half), i = (int)(e - (int64_t)pos * half);
    const float ang = (float)pos * inv_freq[i];
    tab[e] = float2{cosf(ang), sinf(ang)};
}

// One launch per layer for decode steps (every row owns a different KV cache):
//   RoPE(q) from the table, RoPE(k)/v of the new token appended to the cache by the workgroup whose split holds `pos`
//   (and patched into its own K/V fragments, so nothing waits on that store), split-KV attention as above.  A sequence that
//   fits one split is finished here; otherwise each split writes its (m, l, o) partial and vv_attn_merge2_kernel -- a separate
//   wide launch -- combines them in a fixed order (an in-kernel last-arriver merge paid a release fence + device-scope ticket
//   per split and merged through a serial chain of L2 round trips: 25.9 us per layer at 32K positions against 15.5 + 4.9).
//   Replaces rope_append + split + merge (3 launches) by 1 (short contexts) or 2.
// WAVES = waves per workgroup = 32-position blocks in flight per workgroup pass (4: the 8-wave form measured no better at
// 32K positions, twice -- a CU's streaming rate does not grow with its wave count -- and slower at short contexts).
template <int D, int XS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vv_attn_fused_kernel(
    const float* __restrict__ qkv, const VVRow* __restrict__ rows, const float2* __restrict__ rope_tab,
    __bf16* __restrict__ kc, __bf16* __restrict__ vc, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride,
    float q_scale, float* __restrict__ part_m, float* __restrict__ part_l, float* __restrict__ part_o,
    float* __restrict__ out, unsigned char* __restrict__ out_packed) {
    // out_packed (batch decode, <= 16 rows): the finished rows also go out as ONE packed bf16 tile [Hq D / 32][64 lanes][8] -- the
    // o-projection's MFMA B operand (gemv16p.hip), element (row r, k) at lane (r & 15) + 16 ((k & 31) >> 3), slot k & 7 of k-tile k >> 5
    constexpr int KT = D / 32, DT = D / 16, HALF = D / 2;
    const int S = gridDim.x;
    const int split = blockIdx.x, kvh = blockIdx.y, r = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const VVRow rw = rows[r];
    const int pos = rw.pos, len = pos + 1;
    const int G = Hq / Hkv;
    const int g = lane & 15, qg = lane >> 4;
    // split s owns the 32-position blocks b = s, s + S, s + 2S, ...: at any moment the workgroups of a launch read one compact
    // region of the cache, i.e. the concurrent requests spread over every HBM channel instead of marching through S far-apart
    // regions in lockstep (the online softmax does not care about the order)
    const int n_blocks = (len + 31) >> 5;
    const int used = min(S, n_blocks);
    if (split >= used) return;
    const bool owner = (split == ((pos >> 5) % S));      // the block holding the new token
    const int end = len;
    const int p_first = (split + S * wave) * 32;
    const int p_step = S * WAVES * 32;
    const int QW = (Hq + 2 * Hkv) * D;
    const float* qrow = qkv + (int64_t)r * QW;
    const float2* tp = rope_tab + (int64_t)pos * HALF;
    __bf16* kbase = kc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride;
    __bf16* vbase = vc + (int64_t)rw.cache * cache_stride + (int64_t)kvh * head_stride;

    const u32x4* kt_base = reinterpret_cast<const u32x4*>(kbase);
    const u32x4* vt_base = reinterpret_cast<const u32x4*>(vbase);
    float m = -INFINITY, lsum = 0.f;
    f32x4 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // K/V fragments of the next 32-position block are requested before the current one is consumed; the FIRST block's loads go
    // out here, before the q / RoPE work: they depend on nothing but the row table
    u32x4 nka[KT], nkb[KT], nvt[DT];
#define VV_KVLD(p) (*(p))       // cacheable: a short context's K/V is re-read from L2 / MALL frame after frame (nt measured slower)
    auto kv_load = [&](int p0) {
        const int64_t t0 = (int64_t)(p0 >> 4) * KT;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            nka[kt] = VV_KVLD(kt_base + (t0 + kt) * 64 + lane);
            nkb[kt] = VV_KVLD(kt_base + (t0 + KT + kt) * 64 + lane);
        }
        const int64_t vt0 = (int64_t)(p0 >> 5) * DT;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) nvt[dt] = VV_KVLD(vt_base + (vt0 + dt) * 64 + lane);
    };
#undef VV_KVLD
    if (p_first < end) kv_load(p_first);

    // ---- new token's K (rotated) / V: LDS copy for the patch below + the cache append (fire and forget) ----
    // (all LDS of this kernel is one dynamic block: the 8-wave form's partial tiles alone are 64 KiB, the static limit)
    extern __shared__ __attribute__((aligned(16))) unsigned char af_lds[];
    f32x4 (*so)[DT][64] = reinterpret_cast<f32x4 (*)[DT][64]>(af_lds);                              // [WAVES][DT][64]
    float (*sm)[16] = reinterpret_cast<float (*)[16]>(af_lds + (size_t)WAVES * DT * 64 * 16);     // [WAVES][16]
    float (*sl)[16] = sm + WAVES;                                                                  // [WAVES][16]
    __bf16* knew = reinterpret_cast<__bf16*>(sl + WAVES);                                          // [D]
    __bf16* vnew = knew + D;                                                                       // [D]
    if (owner && tid < D) {
        const int d = tid, i = d & (HALF - 1);
        const float* ksrc = qrow + (int64_t)(Hq + kvh) * D;
        const float2 cs = tp[i];
        const float x = ksrc[d], xp = ksrc[d < HALF ? d + HALF : d - HALF];
        const float kr = d < HALF ? x * cs.x - xp * cs.y : x * cs.x + xp * cs.y;
        const __bf16 kb = (__bf16)kr;
        knew[d] = kb;
        {
            const int pt = pos >> 4, pl = pos & 15;
            const int64_t tile = (int64_t)pt * (D / 32) + (d >> 5);
            kbase[(tile * 64 + pl + 16 * ((d & 31) >> 3)) * 8 + (d & 7)] = kb;
        }
        const __bf16 vb = (__bf16)qrow[(int64_t)(Hq + Hkv + kvh) * D + d];
        vnew[d] = vb;
        {
            const int blk = pos >> 5, p = pos & 31;
            const int half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
            const int64_t tile = (int64_t)blk * (D / 16) + (d >> 4);
            vbase[(tile * 64 + (d & 15) + 16 * q4) * 8 + half * 4 + rr] = vb;
        }
    }

    // ---- q fragments: RoPE from the table, scaled, split into bf16 terms.  lane holds q[g][kt*32 + qg*8 + j] ----
    bf16x8 qf[KT][XS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int d0 = kt * 32 + qg * 8;
        const int i0 = d0 & (HALF - 1);
        const bool lo = d0 < HALF;
        float v[8];
        if (g < G) {
            const float* qp = qrow + (int64_t)(kvh * G + g) * D;
            const float4 xa = *reinterpret_cast<const float4*>(qp + d0), xb = *reinterpret_cast<const float4*>(qp + d0 + 4);
            const int dp = lo ? d0 + HALF : d0 - HALF;
            const float4 pa = *reinterpret_cast<const float4*>(qp + dp), pb = *reinterpret_cast<const float4*>(qp + dp + 4);
            const float4 c0 = *reinterpret_cast<const float4*>(tp + i0), c1 = *reinterpret_cast<const float4*>(tp + i0 + 2);
            const float4 c2 = *reinterpret_cast<const float4*>(tp + i0 + 4), c3 = *reinterpret_cast<const float4*>(tp + i0 + 6);
            const float x[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
            const float xp[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
            const float cc[8] = {c0.x, c0.z, c1.x, c1.z, c2.x, c2.z, c3.x, c3.z};
            const float ss[8] = {c0.y, c0.w, c1.y, c1.w, c2.y, c2.w, c3.y, c3.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (lo ? x[j] * cc[j] - xp[j] * ss[j] : x[j] * cc[j] + xp[j] * ss[j]) * q_scale;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        split8<XS>(v, qf[kt]);
    }
    __syncthreads();                         // knew / vnew visible to the wave that meets the new token

    for (int p0 = p_first; p0 < end; p0 += p_step) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
        u32x4 ka[KT], kb[KT], vt[DT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) { ka[kt] = nka[kt]; kb[kt] = nkb[kt]; }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) vt[dt] = nvt[dt];
        if (p0 + p_step < end) kv_load(p0 + p_step);
        if (owner && (pos >> 5) == (p0 >> 5)) {
            // the append above may not have landed: take the new token's row / column from LDS instead
            const int p = pos & 31;
            if ((lane & 15) == (p & 15)) {
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const u32x4 nk = *reinterpret_cast<const u32x4*>(&knew[kt * 32 + (lane >> 4) * 8]);
                    if (p < 16) ka[kt] = nk; else kb[kt] = nk;
                }
            }
            const int half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
            if ((lane >> 4) == q4) {
                const int j = half * 4 + rr;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    bf16x8 t = __builtin_bit_cast(bf16x8, vt[dt]);
                    const __bf16 nv = vnew[dt * 16 + (lane & 15)];
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) if (jj == j) t[jj] = nv;
                    vt[dt] = __builtin_bit_cast(u32x4, t);
                }
            }
        }
        if (p0 + 32 > end) vv_zero_v_past_end<DT>(vt, p0, qg, end);     // cache slots past the sequence hold anything: never 0 x NaN
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
        const float mn = fmaxf(m, mx);
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
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);

    // ---- combine the waves (fixed order), ALL waves at work: wave w finishes the output tiles dt = w, w + WAVES, ... (the
    // one-wave form of this step read WAVES x DT partial tiles through a single wave: the longest serial piece of a
    // short-context launch, and what made the 8-wave form -- a load chain half as long -- the slower one)
    if (lane < 16) { sm[wave][lane] = m; sl[wave][lane] = lsum; }
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) so[wave][dt][lane] = o[dt];
    __syncthreads();
    const int64_t gidx = (int64_t)r * Hkv + kvh;
    float* orow = out + ((int64_t)r * Hq + kvh * G + g) * D + qg * 4;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) M = fmaxf(M, sm[w][g]);
    float L = 0.f, fw[WAVES];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const float mw = sm[w][g];
        fw[w] = (mw == -INFINITY) ? 0.f : expf(mw - M);
        L += sl[w][g] * fw[w];
    }
    const int64_t pidx = gidx * S + split;
    if (used > 1 && wave == 0 && lane < 16) { part_m[pidx * 16 + lane] = M; part_l[pidx * 16 + lane] = L; }
    const float inv = 1.0f / L;
#pragma unroll
    for (int dt0 = 0; dt0 < DT; dt0 += WAVES) {
        const int dt = dt0 + wave;
        if (dt >= DT) break;
        f32x4 O = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < WAVES; ++w) O += so[w][dt][lane] * fw[w];
        if (g < G) {
            if (used == 1) {                      // short sequence: this workgroup saw everything
                *reinterpret_cast<float4*>(orow + dt * 16) = float4{O[0] * inv, O[1] * inv, O[2] * inv, O[3] * inv};
                if (out_packed) {
                    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)(O[e] * inv);
                    const int k = (kvh * G + g) * D + dt * 16 + qg * 4;
                    *reinterpret_cast<uint2*>(out_packed + ((((int64_t)(k >> 5) * 64 + (r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7)) * 2)) = __builtin_bit_cast(uint2, pk);
                }
            } else                                  // several splits: vv_attn_merge2_kernel (a separate wide launch that follows in
                                                  // the stream: no fence, no ticket) combines the partials in a fixed order
                *reinterpret_cast<float4*>(part_o + (pidx * 16 + g) * D + dt * 16 + qg * 4) = float4{O[0], O[1], O[2], O[3]};
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
    const int chunk = attn_chunk(len, S);
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

// Merge of the split partials written by vv_attn_fused_kernel: grid (R, Hq), 512 threads = D output
// dimensions x 512/D split groups.  Group j takes splits j, j + NG, ...: eight (m, l, o) triples are requested at once, so a
// 32-way merge is ONE round trip per thread instead of a chain of them inside the last attention workgroup (which also paid a
// release fence + device-scope ticket per split: 0.33 us per split at 32K positions).  Rows that needed one split were
// finished by the attention kernel itself and are skipped.  Fixed assignment and summation order: deterministic.
template <int D>
__global__ __launch_bounds__(512) void vv_attn_merge2_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                             const float* __restrict__ part_o, const VVRow* __restrict__ rows,
                                                             float* __restrict__ out, int Hq, int Hkv, int S, unsigned char* __restrict__ out_packed) {
    constexpr int NG = 512 / D, MB = 8;
    __shared__ float sm[NG][D], sl[NG][D], sa[NG][D];
    const int r = blockIdx.x, h = blockIdx.y;
    const int d = threadIdx.x % D, grp = threadIdx.x / D;
    const int G = Hq / Hkv;
    const int kvh = h / G, g = h - kvh * G;
    const int len = rows[r].pos + 1;
    const int used = min(S, (len + 31) >> 5);
    if (used <= 1) return;
    const int64_t base = ((int64_t)r * Hkv + kvh) * S;
    float mrun = -INFINITY, lrun = 0.f, arun = 0.f;
    for (int b0 = grp; b0 < used; b0 += NG * MB) {
        float mv[MB], lv[MB], ov[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u) {
            const int s2 = b0 + u * NG;
            const int64_t pi = base + (s2 < used ? s2 : b0);
            mv[u] = part_m[pi * 16 + g];
            lv[u] = part_l[pi * 16 + g];
            ov[u] = part_o[(pi * 16 + g) * D + d];
        }
        float mb = mrun;
#pragma unroll
        for (int u = 0; u < MB; ++u) if (b0 + u * NG < used) mb = fmaxf(mb, mv[u]);
        if (mb != -INFINITY) {
            const float fs = (mrun == -INFINITY) ? 0.f : expf(mrun - mb);
            lrun *= fs; arun *= fs;
#pragma unroll
            for (int u = 0; u < MB; ++u) {
                const bool live = (b0 + u * NG < used) && mv[u] != -INFINITY;
                const float f = live ? expf(mv[u] - mb) : 0.f;
                lrun += lv[u] * f;
                arun += ov[u] * f;
            }
            mrun = mb;
        }
    }
    sm[grp][d] = mrun; sl[grp][d] = lrun; sa[grp][d] = arun;
    __syncthreads();
    if (grp != 0) return;
    float M = -INFINITY;
#pragma unroll
    for (int j = 0; j < NG; ++j) M = fmaxf(M, sm[j][d]);
    float L = 0.f, A = 0.f;
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const float f = (sm[j][d] == -INFINITY) ? 0.f : expf(sm[j][d] - M);
        L += sl[j][d] * f;
        A += sa[j][d] * f;
    }
    out[((int64_t)r * Hq + h) * D + d] = A / L;
    if (out_packed) {                                // batch decode: the o-projection's packed bf16 operand (see vv_attn_fused_kernel)
        const int k = h * D + d;
        *reinterpret_cast<__bf16*>(out_packed + ((((int64_t)(k >> 5) * 64 + (r & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7)) * 2)) = (__bf16)(A / L);
    }
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
    return vv_launch_rc(0);
}

template <int D, int XS>
static void attn_go(const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq, int Hkv,
                    int64_t cs, int64_t hs, int S, float* pm, float* pl, float* po, float* out, hipStream_t s) {
    hipLaunchKernelGGL((vv_attn_split_kernel<D, XS>), dim3(S, Hkv, R), dim3(256), 0, s, q, rows,
                       (const __bf16*)kc, (const __bf16*)vc, Hq, Hkv, cs, hs, pm, pl, po);
    hipLaunchKernelGGL((vv_attn_merge_kernel<D>), dim3(R, Hq), dim3(D), 0, s, pm, pl, po, rows, out, Hq, Hkv, S);
}

extern "C" int vv_rope_table_launch(const float* inv_freq, void* tab, int n_pos, int half, hipStream_t s) {
    const int64_t n = (int64_t)n_pos * half;
    hipLaunchKernelGGL(vv_rope_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, inv_freq, (float2*)tab, n_pos, half);
    return vv_launch_rc(0);
}

// Decode-step attention in one launch; requires every row to own a different cache (the new token of row r must not
// be visible to -- or needed by -- another row of the same launch).
// waves: 4, or 8 (the caller's choice for contexts that fit ONE split but are several 32-position blocks long: twice the
// K/V requests in flight at once, half the dependent load -> consume iterations per wave)
template <int D, int XS, int W>
static void attn_fused_go(dim3 grid, hipStream_t s, const float* qkv, const VVRow* rows, const void* rope_tab, void* kc, void* vc,
                          int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, float scale, float* pm, float* pl, float* po, float* out, void* out_packed) {
    constexpr size_t smem = (size_t)W * (D / 16) * 64 * 16 + (size_t)2 * W * 16 * 4 + (size_t)2 * D * 2;
    static bool attr = false;
    if (!attr) {
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_attn_fused_kernel<D, XS, W>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr = true;
    }
    hipLaunchKernelGGL((vv_attn_fused_kernel<D, XS, W>), grid, dim3(W * 64), smem, s, qkv, rows, (const float2*)rope_tab, (__bf16*)kc, (__bf16*)vc,
                       Hq, Hkv, cache_stride, head_stride, scale, pm, pl, po, out, (unsigned char*)out_packed);
}
extern "C" int vv_attn_fused_launch(int D, int xs, const float* qkv, const VVRow* rows, const void* rope_tab, void* kc, void* vc,
                                    int R, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, int S, int waves,
                                    float* pm, float* pl, float* po, float* out, void* out_packed, hipStream_t s) {
    if (Hq % Hkv != 0 || Hq / Hkv > 16 || (waves != 4 && waves != 8)) return -1;
    if (out_packed && (R > 16 || ((Hq * D) & 31))) return -1;
    const float scale = 1.0f / sqrtf((float)D);
    const dim3 grid(S, Hkv, R);
#define VV_F(D_, XS_)                                                                                                                   \
    do {                                                                                                                                \
        if (waves == 8) attn_fused_go<D_, XS_, 8>(grid, s, qkv, rows, rope_tab, kc, vc, Hq, Hkv, cache_stride, head_stride, scale, pm, pl, po, out, out_packed); \
        else attn_fused_go<D_, XS_, 4>(grid, s, qkv, rows, rope_tab, kc, vc, Hq, Hkv, cache_stride, head_stride, scale, pm, pl, po, out, out_packed);            \
    } while (0)
    if (D == 128) { if (xs == 1) VV_F(128, 1); else if (xs == 2) VV_F(128, 2); else VV_F(128, 3); }
    else if (D == 64) { if (xs == 1) VV_F(64, 1); else if (xs == 2) VV_F(64, 2); else VV_F(64, 3); }
    else return -1;
#undef VV_F
    if (S > 1) {      // rows that needed one split were finished by the attention kernel; the merge kernel skips them
        if (D == 128) hipLaunchKernelGGL((vv_attn_merge2_kernel<128>), dim3(R, Hq), dim3(512), 0, s, pm, pl, po, rows, out, Hq, Hkv, S, (unsigned char*)out_packed);
        else hipLaunchKernelGGL((vv_attn_merge2_kernel<64>), dim3(R, Hq), dim3(512), 0, s, pm, pl, po, rows, out, Hq, Hkv, S, (unsigned char*)out_packed);
    }
    return vv_launch_rc(0);
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
    return vv_launch_rc(0);
}
