// misc.hip -- the small HBM/L2-bound kernels around the GEMMs: embedding gather,
// row RMSNorm, causal depthwise conv (+layer-scale residual), streaming-state row
// shifts, CFG + DPM-Solver++ update, affine/copy helpers.  All are one element (or
// one float4) per lane, coalesced along the channel axis of time-major buffers.
#include "vv_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// out[i][:] = (float) table[ids[i]][:]     grid (n), block 256
__global__ void vv_embed_kernel(const __bf16* __restrict__ table, const int* __restrict__ ids,
                                float* __restrict__ out, int H) {
    const int i = blockIdx.x;
    const __bf16* src = table + (int64_t)ids[i] * H;
    for (int k = threadIdx.x; k < H; k += blockDim.x) out[(int64_t)i * H + k] = (float)src[k];
}

// y[t][:] = x[t][:] * rsqrt(mean(x^2)+eps) * w.  Rows are independent; a row is handled by one wave
// (rows_per_block = 4) or by the whole 256-thread block (rows_per_block = 1, wide rows) with float4
// loads kept in registers between the two passes.
template <int RPB>
__global__ __launch_bounds__(256) void vv_rmsnorm_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                              const float* __restrict__ w, int T, int C, float eps) {
    constexpr int NTH = 256 / RPB;                 // threads cooperating on one row
    constexpr int MAXV = (RPB == 1) ? 4 : 8;       // float4 per thread kept in registers (C <= NTH*4*MAXV)
    __shared__ float part[4];
    const int sub = threadIdx.x / NTH, tl = threadIdx.x % NTH;
    const int t = blockIdx.x * RPB + sub;
    const bool live = t < T;
    const float* xr = x + (int64_t)(live ? t : 0) * ldx;
    const bool vec = ((C & 3) == 0) && ((ldx & 3) == 0) && ((ldy & 3) == 0) && C <= NTH * 4 * MAXV;
    float4 v[MAXV];
    float s = 0.f;
    if (vec) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (tl + i * NTH) * 4;
            v[i] = (live && c < C) ? *reinterpret_cast<const float4*>(xr + c) : float4{0.f, 0.f, 0.f, 0.f};
            s += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
        }
    } else if (live) {
        for (int c = tl; c < C; c += NTH) { const float q = xr[c]; s += q * q; }
    }
    s = wave_sum(s);
    if (RPB == 1) {
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        s = part[0] + part[1] + part[2] + part[3];
    }
    if (!live) return;
    const float rs = rsqrtf(s / (float)C + eps);
    float* yr = y + (int64_t)t * ldy;
    if (vec) {
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = (tl + i * NTH) * 4;
            if (c < C) {
                float4 ww = w ? *reinterpret_cast<const float4*>(w + c) : float4{1.f, 1.f, 1.f, 1.f};
                float4 o = {v[i].x * rs * ww.x, v[i].y * rs * ww.y, v[i].z * rs * ww.z, v[i].w * rs * ww.w};
                *reinterpret_cast<float4*>(yr + c) = o;
            }
        }
    } else {
        for (int c = tl; c < C; c += NTH) yr[c] = xr[c] * rs * (w ? w[c] : 1.f);
    }
}

// Causal depthwise conv k=7 over time on the normed buffer nb (6 history rows in front),
// fused with bias, layer scale and the residual:  x[t][c] += gamma[c]*(sum_j w[j][c]*nb[t+j][c] + b[c])
__global__ void vv_dwconv_res_kernel(const float* __restrict__ nb, const float* __restrict__ x, float* __restrict__ xo,
                                     const float* __restrict__ w /*[7][C]*/, const float* __restrict__ b,
                                     const float* __restrict__ gamma, int T, int C) {
    const int64_t total = (int64_t)T * C;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(e / C), c = (int)(e - (int64_t)t * C);
        float acc = b[c];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += w[j * C + c] * nb[(int64_t)(t + j) * C + c];
        xo[e] = x[e] + gamma[c] * acc;
    }
}

// Small-T fused variant of the two kernels above: ONE workgroup normalises all T rows into nb
// (behind its 6 history rows) and then applies the depthwise conv + residual in place.  Used for
// the T*C <= 64K stages (C >= 256), where two launches cost more than the arithmetic.
__global__ __launch_bounds__(1024) void vv_normdw_kernel(float* __restrict__ x, float* __restrict__ nb,
                                                         const float* __restrict__ nw, const float* __restrict__ w,
                                                         const float* __restrict__ b, const float* __restrict__ gamma,
                                                         int T, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = wave; t < T; t += 16) {
        const float* xr = x + (int64_t)t * C;
        float s = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        const float rs = rsqrtf(wave_sum(s) / (float)C + eps);
        float* nr = nb + (int64_t)(6 + t) * C;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 ww = *reinterpret_cast<const float4*>(nw + c);
            *reinterpret_cast<float4*>(nr + c) = float4{v.x * rs * ww.x, v.y * rs * ww.y, v.z * rs * ww.z, v.w * rs * ww.w};
        }
    }
    __syncthreads();
    const int total = T * C;
    for (int e = threadIdx.x; e < total; e += 1024) {
        const int t = e / C, c = e - t * C;
        float acc = b[c];
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += w[j * C + c] * nb[(int64_t)(t + j) * C + c];
        x[e] += gamma[c] * acc;
    }
}

// Channel-sliced form of the same fusion for the T <= 8 stages (C = 1024 / 2048): C/256 workgroups instead of one.
// Every workgroup re-derives the T row norms from the full input rows (a few KB, L2), then owns 256 channels:
// norm -> nb rows 6.., depthwise conv over [6 history rows ++ new rows], layer scale, residual.  Output goes to a
// DIFFERENT buffer (the stage ping-pongs): other workgroups are still reading full rows of xin for their norms.
// Every global load of a thread is issued before the first wait.
__device__ __forceinline__ float wave_sum_dpp_m(float v) {
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
template <int NCH>      // NCH = C / 1024 float4 chunks per thread and row
__global__ __launch_bounds__(256) void vv_normdw_sliced_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                               float* __restrict__ nb, const float* __restrict__ nw,
                                                               const float* __restrict__ w, const float* __restrict__ b,
                                                               const float* __restrict__ gamma, int T, int C, float eps,
                                                               const VVSlotIds sl, int64_t sx, int64_t snb) {
    constexpr int TM = 8;
    __shared__ float red[4][TM];
    if (sl.n > 0) {        // slot-batched launch: blockIdx.y picks the utterance, buffers are sx / snb floats apart
        const int64_t id = vv_slot_id(sl.id, blockIdx.y);
        xin += id * sx; xout += id * sx; nb += id * snb;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 256 + tid;
    float4 full[TM][NCH];
    float xc[TM], hist[6], wj[7];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
            full[t][i] = (t < T) ? *reinterpret_cast<const float4*>(xin + (int64_t)t * C + (i * 256 + tid) * 4) : float4{0.f, 0.f, 0.f, 0.f};
        xc[t] = (t < T) ? xin[(int64_t)t * C + c] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) hist[j] = nb[(int64_t)j * C + c];
#pragma unroll
    for (int j = 0; j < 7; ++j) wj[j] = w[j * C + c];
    const float nwc = nw[c], bc = b[c], gc = gamma[c];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NCH; ++i) s += full[t][i].x * full[t][i].x + full[t][i].y * full[t][i].y + full[t][i].z * full[t][i].z + full[t][i].w * full[t][i].w;
        s = wave_sum_dpp_m(s);
        if (lane == 0) red[wave][t] = s;
    }
    __syncthreads();
    float nrm[6 + TM];
#pragma unroll
    for (int j = 0; j < 6; ++j) nrm[j] = hist[j];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const float ss = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        nrm[6 + t] = xc[t] * rsqrtf(ss / (float)C + eps) * nwc;
    }
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        if (t < T) {
            float acc = bc;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += wj[j] * nrm[t + j];
            xout[(int64_t)t * C + c] = xc[t] + gc * acc;
            nb[(int64_t)(6 + t) * C + c] = nrm[6 + t];
        }
    }
}

// Row-tiled form of the same fusion for the middle stages (C = 256 / 512, T = 40 .. 1000): one launch instead of
// vv_rmsnorm_rows_kernel + vv_dwconv_res_kernel.  A workgroup owns RB output rows: it normalises rows t0-6 .. t0+RB-1 into
// LDS (the six halo rows are re-derived from the input, or taken from the streaming history for t < 0), then applies the
// depthwise conv + layer scale + residual.  Output goes to a DIFFERENT buffer (the stage ping-pongs): the neighbours are
// still reading the halo rows of xin.  The owner of a row >= T-6 also writes its normed row into nb (the next frame's history).
template <int RB, int NCH>          // NCH = C / 256 float4 chunks per lane and row
__global__ __launch_bounds__(256) void vv_normdw_rows_kernel(const float* __restrict__ xin, float* __restrict__ xout,
                                                             float* __restrict__ nb, const float* __restrict__ nw,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ gamma, int T, int C, float eps,
                                                             const VVSlotIds sl, int64_t sx, int64_t snb) {
    extern __shared__ __attribute__((aligned(16))) float nrm_sh[];          // [RB + 6][C] normed rows, then [RB][C] raw rows
    if (sl.n > 0) {        // slot-batched launch: blockIdx.y picks the utterance
        const int64_t id = vv_slot_id(sl.id, blockIdx.y);
        xin += id * sx; xout += id * sx; nb += id * snb;
    }
    float* raw_sh = nrm_sh + (size_t)(RB + 6) * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * RB;
    // ---- every global load of phase 1 is issued before the first wait: wave w owns rows rr = 4w .. 4w+3 of the RB + 6 ----
    constexpr int RPW = (RB + 6 + 3) / 4;
    float4 v[RPW][NCH], nwv[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) nwv[q] = *reinterpret_cast<const float4*>(nw + (q * 64 + lane) * 4);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int rr = wave * RPW + i, t = t0 - 6 + rr;
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = (q * 64 + lane) * 4;
            float4 x4 = {0.f, 0.f, 0.f, 0.f};
            if (rr < RB + 6) {
                if (t < 0) x4 = *reinterpret_cast<const float4*>(nb + (int64_t)(6 + t) * C + c);          // history: already normed
                else if (t < T) x4 = *reinterpret_cast<const float4*>(xin + (int64_t)t * C + c);
            }
            v[i][q] = x4;
        }
    }
    // phase-2 operands of this thread's channel group (C/4 divides 256: the group is fixed per thread)
    const int C4 = C >> 2;
    const int c4 = (tid % C4) * 4;
    float4 wj[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) wj[j] = *reinterpret_cast<const float4*>(w + (size_t)j * C + c4);
    const float4 bb = *reinterpret_cast<const float4*>(b + c4), gg = *reinterpret_cast<const float4*>(gamma + c4);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int rr = wave * RPW + i, t = t0 - 6 + rr;
        if (rr >= RB + 6) break;
        float rs = 1.0f;
        const bool cur = (t >= 0 && t < T);
        if (cur) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NCH; ++q) s += v[i][q].x * v[i][q].x + v[i][q].y * v[i][q].y + v[i][q].z * v[i][q].z + v[i][q].w * v[i][q].w;
            rs = rsqrtf(wave_sum(s) / (float)C + eps);
        }
        const bool keep = cur && (rr >= 6) && (t >= T - 6);                 // owned row that belongs to the next frame's history
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
            const int c = (q * 64 + lane) * 4;
            float4 o = v[i][q];
            if (cur) o = float4{o.x * rs * nwv[q].x, o.y * rs * nwv[q].y, o.z * rs * nwv[q].z, o.w * rs * nwv[q].w};
            *reinterpret_cast<float4*>(nrm_sh + (size_t)rr * C + c) = o;
            if (rr >= 6) *reinterpret_cast<float4*>(raw_sh + (size_t)(rr - 6) * C + c) = v[i][q];
            if (keep) *reinterpret_cast<float4*>(nb + (int64_t)(6 + t) * C + c) = o;
        }
    }
    __syncthreads();
    for (int e = tid; e < RB * C4; e += 256) {
        const int r = e / C4;
        const int t = t0 + r;
        if (t >= T) break;
        float4 acc = bb;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float4 nj = *reinterpret_cast<const float4*>(nrm_sh + (size_t)(r + j) * C + c4);
            acc.x += wj[j].x * nj.x; acc.y += wj[j].y * nj.y; acc.z += wj[j].z * nj.z; acc.w += wj[j].w * nj.w;
        }
        const float4 xv = *reinterpret_cast<const float4*>(raw_sh + (size_t)r * C + c4);
        *reinterpret_cast<float4*>(xout + (int64_t)t * C + c4) = float4{xv.x + gg.x * acc.x, xv.y + gg.y * acc.y, xv.z + gg.z * acc.z, xv.w + gg.w * acc.w};
    }
}

// element W[n][k] of a packed matrix (vv_common.h) with KT k-tiles
__device__ __forceinline__ float packed_w(const __bf16* __restrict__ wp, int KT, int n, int k) {
    return (float)wp[(((int64_t)(n >> 4) * KT + (k >> 5)) * 64 + (n & 15) + 16 * ((k & 31) >> 3)) * 8 + (k & 7)];
}

// Encoder stem: causal conv k = 7 over a MONO signal (Cin = 1): out[t][n] = b[n] + sum_j W[n][j] * in[t + j], `in` = the
// time-major input buffer with its 6 history samples in front.  K = 7 is not an MFMA shape (the general GEMM kernel spent
// 10 us on it); one output per thread, coalesced along n.  W packed [N][7].
__global__ void vv_stem_conv_kernel(const float* __restrict__ in, const __bf16* __restrict__ wp, const float* __restrict__ bias,
                                    float* __restrict__ out, int T, int N, const VVSlotIds sl, int64_t s_in, int64_t s_out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= T * N) return;
    if (sl.n > 0) { const int64_t id = vv_slot_id(sl.id, blockIdx.y); in += id * s_in; out += id * s_out; }
    const int t = e / N, n = e - t * N;
    float acc = bias[n];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc += packed_w(wp, 1, n, j) * in[t + j];
    out[e] = acc;
}

// Decoder head: causal conv k = 7 from Cin channels to ONE output channel: out[t] = b + sum_{k < 7 Cin} W[0][k] * X[t * Cin + k]
// (X = the last stage's buffer, 6 history rows in front: a window is 7 Cin contiguous floats).  Four lanes per output sample.
__global__ __launch_bounds__(256) void vv_head_conv1_kernel(const float* __restrict__ x, const __bf16* __restrict__ wp,
                                                            const float* __restrict__ bias, float* __restrict__ out, int T, int Cin,
                                                            const VVSlotIds sl, int64_t s_in, int64_t s_out) {
    extern __shared__ float wsh[];                       // [7 * Cin]
    // slot-batched: input = utterance id's stage buffer, output = row blockIdx.y of a dense [n][s_out] tensor
    if (sl.n > 0) { x += (int64_t)vv_slot_id(sl.id, blockIdx.y) * s_in; out += (int64_t)blockIdx.y * s_out; }
    const int K = 7 * Cin, KT = (K + 31) >> 5;
    for (int k = threadIdx.x; k < K; k += 256) wsh[k] = packed_w(wp, KT, 0, k);
    __syncthreads();
    const int g = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    float acc = 0.f;
    if (g < T) {
        const float* xr = x + (int64_t)g * Cin;
        for (int k = part * 4; k < K; k += 16) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            acc += v.x * wsh[k] + v.y * wsh[k + 1] + v.z * wsh[k + 2] + v.w * wsh[k + 3];
        }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (g < T && part == 0) out[g] = acc + bias[0];
}

// Streaming state carry: for every stateful buffer move rows [T, T+hist) -> [0, hist).
// One lane owns one column and walks rows in ascending order, so overlapping moves
// (T < hist) are race-free.                        grid (n_entries, col_chunks), block 256
struct VVShift { float* buf; int T, hist, C; };
__global__ void vv_shift_rows_kernel(const VVShift* __restrict__ tab) {
    const VVShift e = tab[blockIdx.x];
    for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < e.C; c += gridDim.y * blockDim.x)
        for (int r = 0; r < e.hist; ++r) e.buf[(int64_t)r * e.C + c] = e.buf[(int64_t)(r + e.T) * e.C + c];
}

// Zero the history rows of every stateful buffer (<speech_end>: cache.set_to_zero).
__global__ void vv_zero_hist_kernel(const VVShift* __restrict__ tab) {
    const VVShift e = tab[blockIdx.x];
    const int64_t n = (int64_t)e.hist * e.C;
    for (int64_t i = blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.y * blockDim.x) e.buf[i] = 0.f;
}

// Classifier-free guidance + one DPM-Solver++(2M) update on the n x L latent block.
//   v  = v_u + cfg (v_c - v_u)                  (rows [0,n) cond, [n,2n) uncond of `eps`)
//   x0 = a x - s v
//   x' = cs x + c0 x0 + c1 (x0 - x0_prev)       (c1 = 0 on first-order steps)
// coef = {a, s, cs, c0, c1} for this step.
__global__ void vv_cfg_dpm_kernel(const float* __restrict__ eps, float* __restrict__ x, float* __restrict__ x0_prev,
                                  const float* __restrict__ coef, float cfg, int n, int L, const float* __restrict__ sde_noise) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * L) return;
    const float a = coef[0], s = coef[1], cs = coef[2], c0 = coef[3], c1 = coef[4];
    const float vc = eps[i], vu = eps[i + n * L];
    const float v = vu + cfg * (vc - vu);
    const float xi = x[i];
    const float x0 = a * xi - s * v;
    float xn = cs * xi + c0 * x0 + c1 * (x0 - x0_prev[i]);
    if (sde_noise) xn += coef[5] * sde_noise[i];               // sde-dpmsolver++: coef row = {a, s, cs, c0, c1, cn}
    x0_prev[i] = x0;
    x[i] = xn;
    x[i + n * L] = xn;        // both CFG halves see the same latent (modeling_vibevoice_inference.py:703-704)
}

// y = x * mul + add   (latent un-scaling, copies)
__global__ void vv_affine_kernel(const float* __restrict__ x, float* __restrict__ y, float mul, float add, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * mul + add;
}

// Plain copies and zero fills as kernels of our own.  The launch sequences of a decode step are captured into hipGraphs, and a MEMSET NODE of a
// replayed graph was observed (round 6, ROCm 7.0.2 + 7.2 user space, every GPU of the pool) to fill its range with stale 16-byte patterns
// instead of zeros once other graph executables had been created and destroyed in the process (profiles/r06_memset_node_*):
// the sampler's "previous x0" buffer then held whatever words the pattern was, harmless as denormals (x 0 = 0), a NaN in the processes
// where a word of the pattern had all exponent bits set.  Nothing the engine captures uses hipMemsetAsync / hipMemcpyAsync device-to-device any more.
__global__ void vv_copy_words_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void vv_copy_quads_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void vv_zero_words_kernel(unsigned* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = 0u;
}
// the sampler's start of a frame: both CFG halves of the noisy latent = the noise rows, previous x0 prediction = 0 (one launch for
// what used to be two copy nodes and a memset node)
__global__ void vv_sampler_init_kernel(const float* __restrict__ noise, float* __restrict__ z, float* __restrict__ x0p, int nL) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nL) { const float v = noise[i]; z[i] = v; z[nL + i] = v; x0p[i] = 0.f; }
}

// y[i] = a[i] + b[i]
__global__ void vv_add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = a[i] + b[i];
}

// sinusoidal timestep features: out[i][0:128]=cos(t_i f_k), [128:256]=sin(t_i f_k)
__global__ void vv_tfreq_kernel(const float* __restrict__ t, float* __restrict__ out, int n) {
    const int i = blockIdx.x, k = threadIdx.x;   // block 128
    const float f = expf(-logf(10000.f) * (float)k / 128.f);
    const float arg = t[i] * f;
    out[i * 256 + k] = cosf(arg);
    out[i * 256 + 128 + k] = sinf(arg);
}

// y[t][c] = x[t][c] + v[c]
__global__ void vv_add_rows_kernel(const float* __restrict__ x, const float* __restrict__ v, float* __restrict__ y, int n, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * C) y[i] = x[i] + v[i % C];
}
__global__ void vv_relu_kernel(float* __restrict__ x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fmaxf(x[i], 0.f);
}
// HF KV layout [kvh][L][D] -> tiled cache layout (attn.hip header).  grid (L, kvh), block D threads
template <typename ST>
__global__ void vv_kv_import_kernel(const ST* __restrict__ k, const ST* __restrict__ v, __bf16* __restrict__ kc,
                                    __bf16* __restrict__ vc, int L, int D, int64_t head_stride, int pos0) {
    const int h = blockIdx.y, d = threadIdx.x;
    const int64_t si = ((int64_t)h * L + blockIdx.x) * D + d;
    const int pos = pos0 + (int)blockIdx.x;                  // cache position of source row blockIdx.x
    __bf16* kb = kc + (int64_t)h * head_stride;
    __bf16* vb = vc + (int64_t)h * head_stride;
    {
        const int64_t tile = (int64_t)(pos >> 4) * (D / 32) + (d >> 5);
        const int ln = (pos & 15) + 16 * ((d & 31) >> 3);
        kb[(tile * 64 + ln) * 8 + (d & 7)] = (__bf16)(float)k[si];
    }
    {
        const int p = pos & 31, half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
        const int64_t tile = (int64_t)(pos >> 5) * (D / 16) + (d >> 4);
        const int ln = (d & 15) + 16 * q4;
        vb[(tile * 64 + ln) * 8 + half * 4 + rr] = (__bf16)(float)v[si];
    }
}

// one cached position copied onto another, every layer and kv head of one cache: grid (layers, kvh), block D threads
__global__ void vv_kv_move_kernel(__bf16* __restrict__ kc, __bf16* __restrict__ vc, int D, int64_t layer_stride, int64_t head_stride,
                                  int src, int dst) {
    const int d = threadIdx.x;
    __bf16* kb = kc + (int64_t)blockIdx.x * layer_stride + (int64_t)blockIdx.y * head_stride;
    __bf16* vb = vc + (int64_t)blockIdx.x * layer_stride + (int64_t)blockIdx.y * head_stride;
    auto kidx = [&](int pos) {
        const int64_t tile = (int64_t)(pos >> 4) * (D / 32) + (d >> 5);
        return (tile * 64 + (pos & 15) + 16 * ((d & 31) >> 3)) * 8 + (d & 7);
    };
    auto vidx = [&](int pos) {
        const int p = pos & 31, half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
        const int64_t tile = (int64_t)(pos >> 5) * (D / 16) + (d >> 4);
        return (tile * 64 + (d & 15) + 16 * q4) * 8 + half * 4 + rr;
    };
    kb[kidx(dst)] = kb[kidx(src)];
    vb[vidx(dst)] = vb[vidx(src)];
}

// Prompt pass: the V slots between the end of the chunk and the end of its last 64-position attention stage are zeroed in every layer and
// kv head of the chunk's cache.  vv_attn_prefill4 multiplies whole 64-position stages; the positions past the chunk get a softmax weight
// of exactly 0 (their scores are selected to -inf), but 0 x NaN = NaN and a re-used / imported / replay-polluted slot may hold anything
// (the decode kernels zero the fragments in registers instead: vv_zero_v_past_end).  The chunk's last row is read from the DEVICE row
// table (the launch sits in a hipGraph keyed by shapes, not by positions).  grid (layers, kvh), block D threads.
__global__ void vv_kv_zero_v_tail_kernel(__bf16* __restrict__ vc, const VVRow* __restrict__ rows, int R, int D, int64_t cache_stride,
                                         int64_t layer_stride, int64_t head_stride, int max_ctx) {
    const int d = threadIdx.x;
    const VVRow last = rows[R - 1];
    const int p0 = last.pos + 1;
    const int p1 = min(max_ctx, (p0 + 63) & ~63);
    __bf16* vb = vc + (int64_t)last.cache * cache_stride + (int64_t)blockIdx.x * layer_stride + (int64_t)blockIdx.y * head_stride;
    for (int pos = p0; pos < p1; ++pos) {
        const int p = pos & 31, half = p >> 4, pp = p & 15, q4 = pp >> 2, rr = pp & 3;
        const int64_t tile = (int64_t)(pos >> 5) * (D / 16) + (d >> 4);
        vb[(tile * 64 + (d & 15) + 16 * q4) * 8 + half * 4 + rr] = (__bf16)0.0f;
    }
}

// 16-bit PCM of one chunk per workgroup, the arithmetic of the reference's convert_to_16_bit_wav (demo/gradio_demo.py:1058-1073):
// peak = max|x|; if peak > 1: x /= peak (fp32, IEEE division); (x * 32767) truncated toward zero to int16.
__global__ __launch_bounds__(256) void vv_pcm16_kernel(const float* __restrict__ x, short* __restrict__ out, int samples) {
    const float* xr = x + (size_t)blockIdx.x * samples;
    short* orow = out + (size_t)blockIdx.x * samples;
    float mx = 0.f;
    for (int i = threadIdx.x; i < samples; i += 256) mx = fmaxf(mx, fabsf(xr[i]));
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    __shared__ float wmx[4];
    if ((threadIdx.x & 63) == 0) wmx[threadIdx.x >> 6] = mx;
    __syncthreads();
    const float peak = fmaxf(fmaxf(wmx[0], wmx[1]), fmaxf(wmx[2], wmx[3]));
    for (int i = threadIdx.x; i < samples; i += 256) {
        float v = xr[i];
        if (peak > 1.0f) v = __fdiv_rn(v, peak);
        orow[i] = (short)(int)__fmul_rn(v, 32767.0f);
    }
}

// adaLN input rows for every solver step at once: out[i*rows + r] = SiLU(cond_proj[r] + t_emb[i])  (float4 lanes)
__global__ void vv_ada_in_kernel(const float* __restrict__ cproj, const float* __restrict__ temb, float* __restrict__ out,
                                 int rows, int n_steps, int H) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;          // float4 index
    const int H4 = H >> 2;
    if (q >= rows * n_steps * H4) return;
    const int t = q / H4, k4 = q - t * H4;
    const int i = t / rows, r = t - i * rows;
    const float4 c = reinterpret_cast<const float4*>(cproj)[r * H4 + k4];
    const float4 e = reinterpret_cast<const float4*>(temb)[i * H4 + k4];
    float4 o;
    float u;
    u = c.x + e.x; o.x = u / (1.0f + expf(-u));
    u = c.y + e.y; o.y = u / (1.0f + expf(-u));
    u = c.z + e.z; o.z = u / (1.0f + expf(-u));
    u = c.w + e.w; o.w = u / (1.0f + expf(-u));
    reinterpret_cast<float4*>(out)[q] = o;
}

__global__ void vv_silu_kernel(float* __restrict__ x, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float u = x[i]; x[i] = u / (1.f + expf(-u)); }
}

// fp32 <-> storage conversions used when uploading parameters
__global__ void vv_cvt_bf16_to_f32_kernel(const __bf16* __restrict__ s, float* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (float)s[i];
}
__global__ void vv_cvt_f32_to_bf16_kernel(const float* __restrict__ s, __bf16* __restrict__ d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = (__bf16)s[i];
}
// depthwise weight [C][1][7] -> [7][C]
__global__ void vv_dw_transpose_kernel(const float* __restrict__ s, float* __restrict__ d, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C * 7) { const int c = i / 7, j = i - c * 7; d[j * C + c] = s[i]; }
}

// lm_head over the WHOLE vocabulary for n hidden rows (n <= 16): logits[r][v] = sum_k hidden[r][k] * W[v][k], W row-major bf16 (the
// embedding / lm_head table as uploaded).  Only the full-vocabulary logits processors need it (top-k / top-p / min-p /
// repetition penalty act on every token before the valid-id constraint, modeling_vibevoice_inference.py:310-319,488-496); the
// greedy / plain-sampling path evaluates the <= 16 valid rows with vv_lm_logits.  One wave per vocabulary row: the row is read
// once from HBM (8 bf16 per lane and step, fp32 products and sums), the n hidden rows come from L2; HBM-bound (2 V H bytes).
__global__ __launch_bounds__(256) void vv_logits_full_kernel(const __bf16* __restrict__ table, const float* __restrict__ hidden,
                                                             float* __restrict__ out, int n, int V, int H) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    const __bf16* wrow = table + (int64_t)v * H;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = lane * 8; k < H; k += 512) {           // H % 8 == 0 is a launch precondition
        const u32x4 wv = *reinterpret_cast<const u32x4*>(wrow + k);
        const bf16x8 w8 = __builtin_bit_cast(bf16x8, wv);
        float wf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = (float)w8[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (r < n) {
                const float4 a = *reinterpret_cast<const float4*>(hidden + (int64_t)r * H + k);
                const float4 b = *reinterpret_cast<const float4*>(hidden + (int64_t)r * H + k + 4);
                acc[r] += a.x * wf[0] + a.y * wf[1] + a.z * wf[2] + a.w * wf[3] + b.x * wf[4] + b.y * wf[5] + b.z * wf[6] + b.w * wf[7];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < n) {
            float t = acc[r];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (lane == 0) out[(int64_t)r * V + v] = t;
        }
    }
}

}  // namespace

static inline int okk() { return vv_launch_rc(0); }

extern "C" {

int vv_logits_full_launch(const void* table, const float* hidden, float* out, int n, int V, int H, hipStream_t s) {
    if (n < 1 || n > 16 || V < 1 || (H & 7)) return -1;
    hipLaunchKernelGGL(vv_logits_full_kernel, dim3((V + 3) / 4), dim3(256), 0, s, (const __bf16*)table, hidden, out, n, V, H);
    return okk();
}
int vv_embed_launch(const void* table, const int* ids, float* out, int n, int H, hipStream_t s) {
    hipLaunchKernelGGL(vv_embed_kernel, dim3(n), dim3(256), 0, s, (const __bf16*)table, ids, out, H);
    return okk();
}
int vv_rmsnorm_rows_launch(const float* x, int ldx, float* y, int ldy, const float* w, int T, int C, float eps, hipStream_t s) {
    if (C > 1024 || T < 64)
        hipLaunchKernelGGL((vv_rmsnorm_rows_kernel<1>), dim3(T), dim3(256), 0, s, x, ldx, y, ldy, w, T, C, eps);
    else
        hipLaunchKernelGGL((vv_rmsnorm_rows_kernel<4>), dim3((T + 3) / 4), dim3(256), 0, s, x, ldx, y, ldy, w, T, C, eps);
    return okk();
}
int vv_dwconv_res_launch(const float* nb, const float* x, float* xo, const float* w, const float* b, const float* gamma, int T, int C, hipStream_t s) {
    int64_t total = (int64_t)T * C;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(vv_dwconv_res_kernel, dim3(blocks), dim3(256), 0, s, nb, x, xo, w, b, gamma, T, C);
    return okk();
}
int vv_normdw_launch(float* x, float* nb, const float* nw, const float* w, const float* b, const float* gamma,
                     int T, int C, float eps, hipStream_t s) {
    hipLaunchKernelGGL(vv_normdw_kernel, dim3(1), dim3(1024), 0, s, x, nb, nw, w, b, gamma, T, C, eps);
    return okk();
}
int vv_normdw_sliced_ok(int T, int C) { return T >= 1 && T <= 8 && (C == 1024 || C == 2048); }
int vv_normdw_sliced_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                            const float* gamma, int T, int C, float eps, hipStream_t s) {
    if (!vv_normdw_sliced_ok(T, C) || xin == xout) return -1;
    VVSlotIds none; none.n = 0;
    if (C == 1024) hipLaunchKernelGGL((vv_normdw_sliced_kernel<1>), dim3(C / 256), dim3(256), 0, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, none, (int64_t)0, (int64_t)0);
    else hipLaunchKernelGGL((vv_normdw_sliced_kernel<2>), dim3(C / 256), dim3(256), 0, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, none, (int64_t)0, (int64_t)0);
    return okk();
}
// the same kernel over n utterance slots: xin / xout / nb are slot 0's buffers, slot k's are sx / snb floats further
int vv_normdw_sliced_slots_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                                  const float* gamma, int T, int C, float eps, const int* ids, int n, int64_t sx, int64_t snb,
                                  hipStream_t s) {
    if (!vv_normdw_sliced_ok(T, C) || xin == xout || n < 1 || n > 8) return -1;
    VVSlotIds sl; sl.n = n;
    for (int i = 0; i < 8; ++i) sl.id[i] = i < n ? ids[i] : 0;
    if (C == 1024) hipLaunchKernelGGL((vv_normdw_sliced_kernel<1>), dim3(C / 256, n), dim3(256), 0, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, sl, sx, snb);
    else hipLaunchKernelGGL((vv_normdw_sliced_kernel<2>), dim3(C / 256, n), dim3(256), 0, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, sl, sx, snb);
    return okk();
}
// y[id[j] * stride + c] = x[j * L + c] * mul + add: the batch's latents into the per-utterance decoder input buffers
static __global__ void vv_affine_slots_kernel(const float* __restrict__ x, float* __restrict__ y, float mul, float add, int L,
                                       const VVSlotIds sl, int64_t stride) {
    const int j = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < L) y[(int64_t)vv_slot_id(sl.id, j) * stride + c] = x[(int64_t)j * L + c] * mul + add;
}
int vv_affine_slots_launch(const float* x, float* y, float mul, float add, int L, const int* ids, int n, int64_t stride, hipStream_t s) {
    if (n < 1 || n > 8) return -1;
    VVSlotIds sl; sl.n = n;
    for (int i = 0; i < 8; ++i) sl.id[i] = i < n ? ids[i] : 0;
    hipLaunchKernelGGL(vv_affine_slots_kernel, dim3((L + 255) / 256, n), dim3(256), 0, s, x, y, mul, add, L, sl, stride);
    return okk();
}
int vv_normdw_rows_ok(int T, int C) { return T >= 1 && (C == 256 || C == 512 || C == 1024); }
int vv_normdw_rows_slots_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                                const float* gamma, int T, int C, float eps, const int* ids, int n, int64_t sx, int64_t snb,
                                hipStream_t s);
int vv_normdw_rows_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                          const float* gamma, int T, int C, float eps, hipStream_t s) {
    return vv_normdw_rows_slots_launch(xin, xout, nb, nw, w, b, gamma, T, C, eps, nullptr, 0, 0, 0, s);
}
// ids != null: the same kernel over n utterance slots (xin / xout / nb are slot 0's buffers, slot k's sx / snb floats further)
int vv_normdw_rows_slots_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                                const float* gamma, int T, int C, float eps, const int* ids, int n, int64_t sx, int64_t snb,
                                hipStream_t s) {
    if (!vv_normdw_rows_ok(T, C) || xin == xout || n < 0 || n > 8) return -1;
    constexpr int RB = 8;
    const size_t smem = (size_t)(2 * RB + 6) * C * 4;
    VVSlotIds sl; sl.n = ids ? n : 0;
    for (int i = 0; i < 8; ++i) sl.id[i] = (ids && i < n) ? ids[i] : 0;
    const dim3 grid((T + RB - 1) / RB, sl.n > 0 ? sl.n : 1);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_normdw_rows_kernel<RB, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    if (C == 256) hipLaunchKernelGGL((vv_normdw_rows_kernel<RB, 1>), grid, dim3(256), smem, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, sl, sx, snb);
    else if (C == 512) hipLaunchKernelGGL((vv_normdw_rows_kernel<RB, 2>), grid, dim3(256), smem, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, sl, sx, snb);
    else hipLaunchKernelGGL((vv_normdw_rows_kernel<RB, 4>), grid, dim3(256), smem, s, xin, xout, nb, nw, w, b, gamma, T, C, eps, sl, sx, snb);
    return okk();
}
int vv_stem_conv_slots_launch(const float* in, const void* wp, const float* bias, float* out, int T, int N, const int* ids, int n,
                              int64_t s_in, int64_t s_out, hipStream_t s) {
    if (n < 0 || n > 8) return -1;
    VVSlotIds sl; sl.n = ids ? n : 0;
    for (int i = 0; i < 8; ++i) sl.id[i] = (ids && i < n) ? ids[i] : 0;
    hipLaunchKernelGGL(vv_stem_conv_kernel, dim3((T * N + 255) / 256, sl.n > 0 ? sl.n : 1), dim3(256), 0, s, in, (const __bf16*)wp, bias, out, T, N, sl, s_in, s_out);
    return okk();
}
int vv_stem_conv_launch(const float* in, const void* wp, const float* bias, float* out, int T, int N, hipStream_t s) {
    return vv_stem_conv_slots_launch(in, wp, bias, out, T, N, nullptr, 0, 0, 0, s);
}
int vv_head_conv1_slots_launch(const float* x, const void* wp, const float* bias, float* out, int T, int Cin, const int* ids, int n,
                               int64_t s_in, int64_t s_out, hipStream_t s) {
    if ((Cin & 3) || (((uintptr_t)x) & 15) || 7 * Cin * 4 > 48 * 1024 || n < 0 || n > 8 || (s_in & 3)) return -1;
    VVSlotIds sl; sl.n = ids ? n : 0;
    for (int i = 0; i < 8; ++i) sl.id[i] = (ids && i < n) ? ids[i] : 0;
    hipLaunchKernelGGL(vv_head_conv1_kernel, dim3((T + 63) / 64, sl.n > 0 ? sl.n : 1), dim3(256), (size_t)7 * Cin * 4, s, x, (const __bf16*)wp, bias, out, T, Cin, sl, s_in, s_out);
    return okk();
}
int vv_head_conv1_launch(const float* x, const void* wp, const float* bias, float* out, int T, int Cin, hipStream_t s) {
    return vv_head_conv1_slots_launch(x, wp, bias, out, T, Cin, nullptr, 0, 0, 0, s);
}
int vv_shift_rows_launch(const void* tab, int n_entries, int maxC, hipStream_t s) {
    int cy = (maxC + 255) / 256; if (cy < 1) cy = 1; if (cy > 16) cy = 16;
    hipLaunchKernelGGL(vv_shift_rows_kernel, dim3(n_entries, cy), dim3(256), 0, s, (const VVShift*)tab);
    return okk();
}
int vv_zero_hist_launch(const void* tab, int n_entries, hipStream_t s) {
    hipLaunchKernelGGL(vv_zero_hist_kernel, dim3(n_entries, 8), dim3(256), 0, s, (const VVShift*)tab);
    return okk();
}
int vv_cfg_dpm_launch(const float* eps, float* x, float* x0_prev, const float* coef, float cfg, int n, int L, const float* sde_noise, hipStream_t s) {
    hipLaunchKernelGGL(vv_cfg_dpm_kernel, dim3((n * L + 255) / 256), dim3(256), 0, s, eps, x, x0_prev, coef, cfg, n, L, sde_noise);
    return okk();
}
int vv_affine_launch(const float* x, float* y, float mul, float add, int n, hipStream_t s) {
    hipLaunchKernelGGL(vv_affine_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, y, mul, add, n);
    return okk();
}
int vv_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s) {      // bytes: a multiple of 4
    if (bytes == 0) return 0;
    if ((bytes & 3) || ((uintptr_t)dst & 3) || ((uintptr_t)src & 3)) { g_vv_launch_err = (int)hipErrorInvalidValue; return -1; }
    if (!(bytes & 15) && !((uintptr_t)dst & 15) && !((uintptr_t)src & 15)) {
        const size_t n = bytes >> 4;
        hipLaunchKernelGGL(vv_copy_quads_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n);
    } else {
        const size_t n = bytes >> 2;
        hipLaunchKernelGGL(vv_copy_words_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, (const unsigned*)src, (unsigned*)dst, n);
    }
    return okk();
}
int vv_zero_launch(void* dst, size_t bytes, hipStream_t s) {                        // bytes: a multiple of 4
    if (bytes == 0) return 0;
    if ((bytes & 3) || ((uintptr_t)dst & 3)) { g_vv_launch_err = (int)hipErrorInvalidValue; return -1; }
    const size_t n = bytes >> 2;
    hipLaunchKernelGGL(vv_zero_words_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, s, (unsigned*)dst, n);
    return okk();
}
int vv_sampler_init_launch(const float* noise, float* z, float* x0p, int nL, hipStream_t s) {
    hipLaunchKernelGGL(vv_sampler_init_kernel, dim3((nL + 255) / 256), dim3(256), 0, s, noise, z, x0p, nL);
    return okk();
}
int vv_add_launch(const float* a, const float* b, float* y, int n, hipStream_t s) {
    hipLaunchKernelGGL(vv_add_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, y, n);
    return okk();
}
int vv_tfreq_launch(const float* t, float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(vv_tfreq_kernel, dim3(n), dim3(128), 0, s, t, out, n);
    return okk();
}
int vv_add_rows_launch(const float* x, const float* v, float* y, int n, int C, hipStream_t s) {
    hipLaunchKernelGGL(vv_add_rows_kernel, dim3((n * C + 255) / 256), dim3(256), 0, s, x, v, y, n, C);
    return okk();
}
int vv_relu_launch(float* x, int n, hipStream_t s) {
    hipLaunchKernelGGL(vv_relu_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n);
    return okk();
}
int vv_kv_import_launch(const void* k, const void* v, int src_bf16, void* kc, void* vc, int L, int Hkv, int D,
                        int64_t head_stride, int pos0, hipStream_t s) {
    if (src_bf16) hipLaunchKernelGGL((vv_kv_import_kernel<__bf16>), dim3(L, Hkv), dim3(D), 0, s, (const __bf16*)k, (const __bf16*)v, (__bf16*)kc, (__bf16*)vc, L, D, head_stride, pos0);
    else hipLaunchKernelGGL((vv_kv_import_kernel<float>), dim3(L, Hkv), dim3(D), 0, s, (const float*)k, (const float*)v, (__bf16*)kc, (__bf16*)vc, L, D, head_stride, pos0);
    return okk();
}
int vv_kv_move_launch(void* kc, void* vc, int layers, int Hkv, int D, int64_t layer_stride, int64_t head_stride, int src, int dst, hipStream_t s) {
    hipLaunchKernelGGL(vv_kv_move_kernel, dim3(layers, Hkv), dim3(D), 0, s, (__bf16*)kc, (__bf16*)vc, D, layer_stride, head_stride, src, dst);
    return okk();
}
int vv_kv_zero_v_tail_launch(void* vc, const VVRow* rows, int R, int layers, int Hkv, int D, int64_t cache_stride, int64_t layer_stride,
                             int64_t head_stride, int max_ctx, hipStream_t s) {
    hipLaunchKernelGGL(vv_kv_zero_v_tail_kernel, dim3(layers, Hkv), dim3(D), 0, s, (__bf16*)vc, rows, R, D, cache_stride, layer_stride, head_stride, max_ctx);
    return okk();
}
int vv_pcm16_launch(const float* x, short* out, int n, int samples, hipStream_t s) {
    hipLaunchKernelGGL(vv_pcm16_kernel, dim3(n), dim3(256), 0, s, x, out, samples);
    return okk();
}
int vv_ada_in_launch(const float* cproj, const float* temb, float* out, int rows, int n_steps, int H, hipStream_t s) {
    if (H & 3) return -1;
    const int n4 = rows * n_steps * (H >> 2);
    hipLaunchKernelGGL(vv_ada_in_kernel, dim3((n4 + 255) / 256), dim3(256), 0, s, cproj, temb, out, rows, n_steps, H);
    return vv_launch_rc(0);
}
int vv_silu_launch(float* x, int n, hipStream_t s) {
    hipLaunchKernelGGL(vv_silu_kernel, dim3((n + 255) / 256), dim3(256), 0, s, x, n);
    return okk();
}
int vv_cvt_launch(const void* src, void* dst, int64_t n, int to_bf16, hipStream_t s) {
    int blocks = (int)((n + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    if (to_bf16) hipLaunchKernelGGL(vv_cvt_f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, s, (const float*)src, (__bf16*)dst, n);
    else hipLaunchKernelGGL(vv_cvt_bf16_to_f32_kernel, dim3(blocks), dim3(256), 0, s, (const __bf16*)src, (float*)dst, n);
    return okk();
}
int vv_dw_transpose_launch(const float* src, float* dst, int C, hipStream_t s) {
    hipLaunchKernelGGL(vv_dw_transpose_kernel, dim3((C * 7 + 255) / 256), dim3(256), 0, s, src, dst, C);
    return okk();
}

}  // extern "C"
