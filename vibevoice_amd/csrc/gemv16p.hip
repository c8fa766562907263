// gemv16p.hip -- batch decode (5..16 activation rows, bf16 mode): weight-streaming GEMV over PRE-PACKED activations.
//
// The 16-row form of vv_gemv_kernel (gemv.hip) makes every workgroup (one per 16 output features, 200 .. 1200 of them)
// load the launch's fp32 activation rows, normalise / modulate them and convert them to bf16 MFMA fragments in LDS.  With
// 2 rows that work is noise; with 16 rows it is 16 x K x {4 B load, ~10 VALU ops} PER WORKGROUP -- for the diffusion head's
// gate/up projection (K = 3584, 672 workgroups) three times the bytes of the weight slice the workgroup streams, and the
// kernel turns VALU-bound: 56 us against 29 us for the same weights at 2 rows.  Here the activation is packed ONCE:
//
//   vv_pack16_kernel     one workgroup per row: RMSNorm (optionally adaLN-modulated: x^ = rs x nw (1 + scale) + shift) ->
//                        bf16 MFMA B fragments [K/32][64 lanes][8] (the layout of vv_common.h, rows = fragment columns);
//   vv_gemv16p_kernel    one workgroup per 16 features, waves split K: each k-step is one 1 KiB non-temporal weight load +
//                        one 1 KiB fragment load of the packed activation (L2-resident, 2 B per element) -> MFMA.  No LDS
//                        staging, no VALU work in the stream.  Epilogues: bias, residual, gated residual (fp32 rows out)
//                        and SwiGLU, which writes its result straight as packed fragments for the down projection.
//
// Round 6: most vv_pack16 launches of a batch step are gone (163 launches x 4.7 us of a 12.2 ms step).  RMSNorm's 1/rms is a per-ROW
// scale, so it commutes with the contraction: a residual epilogue (PK) that has just produced 16 features of the new residual rows
// also writes them, times the NEXT norm's weight, as packed bf16 fragments, and the row's partial sum of squares over its 16 features
// to ssq_out[tile][16]; the consumer (RS) sums the partials of all tiles in a fixed order while its first weight batch is in flight
// and scales its accumulator rows by rsqrt(sum / K + eps) in the epilogue -- what the 2-row decode GEMV has always done inside one
// kernel (gemv.hip: "sum(x^2) gathered during staging, applied to the accumulator").  Deterministic: fixed summation order, no atomics.
// The diffusion head's adaLN form y = rs W (x w (1 + scale)) + W shift takes the shift rows as a SECOND packed operand (packed once
// per frame for every solver step and layer: they depend on the condition and t only).
#include <cstdlib>
#include "vv_common.h"

namespace {

__device__ __forceinline__ float p16_silu(float u) { return u / (1.0f + expf(-u)); }

__device__ __forceinline__ float p16_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// MODE 0: plain conversion;  MODE 1: RMSNorm with weight nw;  MODE 2: adaLN-modulated RMSNorm (scale / shift rows, stride ld_mod).
// grid = 16 * ceil(T / 16) workgroups (rows >= T are written as zeros), 256 threads; K % 32 == 0, K <= 8192.
template <int MODE>
__global__ __launch_bounds__(256) void vv_pack16_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ nw, float eps,
                                                        const float* __restrict__ sc, const float* __restrict__ sh, int ld_mod,
                                                        unsigned char* __restrict__ xp, int T, int K) {
    constexpr int MAXQ = 8;                              // float4 chunks per thread
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    const int KT = K >> 5, K4 = K >> 2;
    float4 v[MAXQ];
    float ssq = 0.f;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int q = tid + i * 256;
        v[i] = (t < T && q < K4) ? *reinterpret_cast<const float4*>(x + (int64_t)t * ldx + q * 4) : float4{0.f, 0.f, 0.f, 0.f};
        ssq += v[i].x * v[i].x + v[i].y * v[i].y + v[i].z * v[i].z + v[i].w * v[i].w;
    }
    ssq = p16_wave_sum(ssq);
    if (lane == 0) red[wave] = ssq;
    __syncthreads();
    const float rs = (MODE == 0) ? 1.0f : rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)K + eps);
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int q = tid + i * 256;
        if (q >= K4) break;
        const int k = q * 4;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (t < T) {
            const float4 w4 = (MODE != 0 && nw) ? *reinterpret_cast<const float4*>(nw + k) : float4{1.f, 1.f, 1.f, 1.f};
            o[0] = v[i].x * rs * w4.x; o[1] = v[i].y * rs * w4.y; o[2] = v[i].z * rs * w4.z; o[3] = v[i].w * rs * w4.w;
            if constexpr (MODE == 2) {
                const float4 s4 = *reinterpret_cast<const float4*>(sc + (int64_t)t * ld_mod + k);
                const float4 h4 = *reinterpret_cast<const float4*>(sh + (int64_t)t * ld_mod + k);
                o[0] = o[0] * (1.f + s4.x) + h4.x; o[1] = o[1] * (1.f + s4.y) + h4.y;
                o[2] = o[2] * (1.f + s4.z) + h4.z; o[3] = o[3] * (1.f + s4.w) + h4.w;
            }
        }
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 b;
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = (__bf16)o[j];
        // element (t, k): tile (t >> 4) * KT + (k >> 5), lane (t & 15) + 16 * ((k & 31) >> 3), slot k & 7
        const int64_t tile = (int64_t)(t >> 4) * KT + (k >> 5);
        const int ol = (t & 15) + 16 * ((k & 31) >> 3);
        *reinterpret_cast<uint2*>(xp + ((tile * 64 + ol) * 16 + (k & 7) * 2)) = __builtin_bit_cast(uint2, b);
    }
}


// Plain conversion of MANY [T][K] fp32 row blocks to packed 16-row tiles in one launch: tile j reads rows at
// x + (j / n_inner) * stride_outer + (j % n_inner) * stride_inner (row stride ldx) and writes xp + j * tile_bytes.  grid (16, n_tiles):
// blockIdx.x = row, blockIdx.y = tile.  Used once per frame for the diffusion head's adaLN shift rows of every (solver step, layer).
__global__ __launch_bounds__(256) void vv_pack16_tiles_kernel(const float* __restrict__ x, int ldx, int64_t stride_outer, int n_inner,
                                                              int64_t stride_inner, unsigned char* __restrict__ xp, int64_t tile_bytes, int T, int K) {
    const int t = blockIdx.x, j = blockIdx.y;
    const float* xr = x + (int64_t)(j / n_inner) * stride_outer + (int64_t)(j % n_inner) * stride_inner + (int64_t)t * ldx;
    unsigned char* out = xp + (int64_t)j * tile_bytes;
    const int KT = K >> 5;
    for (int q = threadIdx.x; q < (K >> 2); q += 256) {
        const int k = q * 4;
        const float4 v = (t < T) ? *reinterpret_cast<const float4*>(xr + k) : float4{0.f, 0.f, 0.f, 0.f};
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 b;
        b[0] = (__bf16)v.x; b[1] = (__bf16)v.y; b[2] = (__bf16)v.z; b[3] = (__bf16)v.w;
        const int64_t tile = (k >> 5);
        const int ol = (t & 15) + 16 * ((k & 31) >> 3);
        *reinterpret_cast<uint2*>(out + ((tile * 64 + ol) * 16 + (k & 7) * 2)) = __builtin_bit_cast(uint2, b);
    }
    (void)KT;
}

constexpr int PU = 8;      // k-steps per batch

// EPI: VV_EPI_BIAS (bias may be null = store), VV_EPI_RESID (Y += acc (+ bias)), VV_EPI_GATED_RESID (Y += gate * acc),
//      VV_EPI_SWIGLU (Yp = bf16(silu(gate_acc) * up_acc), packed).  WPB = waves per workgroup = K split.
// RS / SH / PK: see VVGemv16p.
template <int EPI, int WPB, int RS = 0, int SH = 0, int PK = 0>
__global__ __launch_bounds__(WPB * 64) void vv_gemv16p_kernel(const VVGemv16p a) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int NM = DUAL ? 2 : 1;
    constexpr int NA = NM * (SH ? 2 : 1);          // accumulator sets: [NM, 2NM) = the shift operand's products
    __shared__ f32x4 red[WPB][NA][64];
    __shared__ float rs_sh[RS ? WPB : 1][16];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned tile = blockIdx.x;
    const unsigned k_tiles = (unsigned)(a.K + 31) >> 5;
    const unsigned kper = (k_tiles + WPB - 1) / WPB;
    const unsigned kt0 = wave * kper, kt1 = min(k_tiles, kt0 + kper);
    const bool has_k = kt0 < kt1;
    const int frow = lane & 15, fq = lane >> 4;
    const u32x4* wbase = a.W + (size_t)tile * k_tiles * 64 + lane;
    const u32x4* wbase2 = DUAL ? a.W2 + (size_t)tile * k_tiles * 64 + lane : nullptr;
    const u32x4* xbase = a.Xp + lane;
    const u32x4* sbase = SH ? a.Xs + lane : nullptr;
    constexpr int NX = SH ? 2 : 1;

    auto load = [&](unsigned ktb, u32x4 (&w)[PU][NM], u32x4 (&xf)[PU][NX]) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const unsigned kt = min(ktb + u, kt1 - 1);          // clamped: tail k-steps re-read the last tile, MFMA skipped
            w[u][0] = __builtin_nontemporal_load(wbase + kt * 64);
            if constexpr (DUAL) w[u][1] = __builtin_nontemporal_load(wbase2 + kt * 64);
            xf[u][0] = xbase[kt * 64];
            if constexpr (SH) xf[u][1] = sbase[kt * 64];
        }
    };
    f32x4 acc[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](unsigned ktb, const u32x4 (&w)[PU][NM], const u32x4 (&xf)[PU][NX]) {
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            if (ktb + u < kt1) {
                const bf16x8 xb = __builtin_bit_cast(bf16x8, xf[u][0]);
#pragma unroll
                for (int i = 0; i < NM; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u][i]), xb, acc[i], 0, 0, 0);
                if constexpr (SH) {
                    const bf16x8 sb = __builtin_bit_cast(bf16x8, xf[u][1]);
#pragma unroll
                    for (int i = 0; i < NM; ++i)
                        acc[NM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u][i]), sb, acc[NM + i], 0, 0, 0);
                }
            }
        }
    };
    u32x4 wA[PU][NM], xA[PU][NX];
    if (has_k) load(kt0, wA, xA);
    __builtin_amdgcn_sched_barrier(0);

    // ---- RS: this thread's share of the producers' partial sums of squares, requested behind the first weight batch.  float4 i of
    // ssq_in covers rows 4 (i & 3) .. + 3 of tile i >> 2; the stride (threads per workgroup) is a multiple of 4, so a thread stays in
    // one row group (lane & 3); lanes of a wave are combined by a fixed xor tree, waves in order after the barrier ----
    if constexpr (RS) {
        float4 ps = {0.f, 0.f, 0.f, 0.f};
        const int n4 = a.ssq_tiles * 4;
        for (int i = threadIdx.x; i < n4; i += WPB * 64) {
            const float4 v = reinterpret_cast<const float4*>(a.ssq_in)[i];
            ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
        }
#pragma unroll
        for (int o = 4; o < 64; o <<= 1) {
            ps.x += __shfl_xor(ps.x, o); ps.y += __shfl_xor(ps.y, o); ps.z += __shfl_xor(ps.z, o); ps.w += __shfl_xor(ps.w, o);
        }
        if (lane < 4) *reinterpret_cast<float4*>(&rs_sh[wave][lane * 4]) = ps;
    }

    // ---- epilogue operands of wave 0: requested now, consumed one weight stream later ----
    const int n0 = tile * 16 + fq * 4;
    const bool epi_lane = (wave == 0) && frow < a.T && n0 < a.N;           // N % 4 == 0 is a launch precondition
    float4 pre_y = {0.f, 0.f, 0.f, 0.f}, pre_b = {0.f, 0.f, 0.f, 0.f}, pre_g = {1.f, 1.f, 1.f, 1.f};
    float4 pre_nw = {1.f, 1.f, 1.f, 1.f}, pre_sc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (PK) {
        if (epi_lane) {
            if (a.pk_nw) pre_nw = *reinterpret_cast<const float4*>(a.pk_nw + n0);
            if (a.pk_sc) pre_sc = *reinterpret_cast<const float4*>(a.pk_sc + (size_t)frow * a.ld_pk + n0);
        }
    }
    if (epi_lane) {
        if constexpr (EPI == VV_EPI_BIAS || EPI == VV_EPI_RESID) {
            if (a.bias) pre_b = *reinterpret_cast<const float4*>(a.bias + n0);
        }
        if constexpr (EPI == VV_EPI_RESID || EPI == VV_EPI_GATED_RESID)
            pre_y = *reinterpret_cast<const float4*>(a.Y + (size_t)frow * a.ldy + n0);
        if constexpr (EPI == VV_EPI_GATED_RESID)
            pre_g = *reinterpret_cast<const float4*>(a.gate + (size_t)frow * a.ld_gate + n0);
    }

    if (has_k) {
        if constexpr (DUAL) {
            // two weight streams: a second buffer set would halve the resident workgroups; the other waves cover the latency
#pragma unroll 1
            for (unsigned ktb = kt0; ktb < kt1; ktb += PU) {
                mma(ktb, wA, xA);
                if (ktb + PU < kt1) load(ktb + PU, wA, xA);
            }
        } else {
            u32x4 wB[PU][NM], xB[PU][NX];
#pragma unroll 1
            for (unsigned ktb = kt0; ktb < kt1; ktb += 2 * PU) {
                const bool n1 = ktb + PU < kt1, n2 = ktb + 2 * PU < kt1;
                if (n1) load(ktb + PU, wB, xB);
                mma(ktb, wA, xA);
                if (n2) load(ktb + 2 * PU, wA, xA);
                if (n1) mma(ktb + PU, wB, xB);
            }
        }
    }
    // ---- split-K partials -> LDS, one barrier, wave 0 finishes ----
#pragma unroll
    for (int i = 0; i < NA; ++i) red[wave][i][lane] = acc[i];
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < WPB; ++w)
#pragma unroll
        for (int i = 0; i < NA; ++i) acc[i] += red[w][i][lane];
    if constexpr (RS) {                                  // rows of the accumulator = lanes' frow: 1/rms of that row
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += rs_sh[w][frow];
        const float rs = rsqrtf(s / (float)a.K + a.eps);
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] *= rs;
    }
    if constexpr (SH) {
#pragma unroll
        for (int i = 0; i < NM; ++i) acc[i] += acc[NM + i];
    }
    if constexpr (PK) {
        // every lane of wave 0 takes part (the row sums cross the four feature quarters of a row): lanes outside the launch's rows carry zeros
        float o[4];
        const float pb[4] = {pre_b.x, pre_b.y, pre_b.z, pre_b.w};
        const float py[4] = {pre_y.x, pre_y.y, pre_y.z, pre_y.w};
        const float pg[4] = {pre_g.x, pre_g.y, pre_g.z, pre_g.w};
        const float nw4[4] = {pre_nw.x, pre_nw.y, pre_nw.z, pre_nw.w}, sc4[4] = {pre_sc.x, pre_sc.y, pre_sc.z, pre_sc.w};
        float sq = 0.f;
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (EPI == VV_EPI_RESID) o[r] = py[r] + (acc[0][r] + pb[r]);
            else o[r] = py[r] + pg[r] * acc[0][r];
            if (!epi_lane) o[r] = 0.f;
            sq += o[r] * o[r];
            pk[r] = (__bf16)(o[r] * nw4[r] * (1.f + sc4[r]));
        }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (fq == 0) a.ssq_out[tile * 16 + frow] = sq;            // all 16 rows (zeros past T): the consumer sums whole float4s
        if (n0 < a.N) {
            const int64_t otile = n0 >> 5;
            const int ol = frow + 16 * ((n0 & 31) >> 3);
            *reinterpret_cast<uint2*>(a.Yp + ((otile * 64 + ol) * 16 + (n0 & 7) * 2)) = __builtin_bit_cast(uint2, pk);
            if (epi_lane) *reinterpret_cast<float4*>(a.Y + (size_t)frow * a.ldy + n0) = float4{o[0], o[1], o[2], o[3]};
        }
        return;
    }
    if (!epi_lane) return;
    if constexpr (EPI == VV_EPI_CFG_DPM) {
        const int nc = a.n_cfg;
        const float ca = a.coef[0], cs_ = a.coef[1], csx = a.coef[2], c0 = a.coef[3], c1 = a.coef[4];
        const float cn = a.sde_noise ? a.coef[5] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float oc = acc[0][r];
            const float vu = __shfl(oc, lane + nc);
            const int n = n0 + r;
            if (frow < nc && n < a.N) {
                const float v = vu + a.cfg * (oc - vu);
                const unsigned zi = (unsigned)(frow * a.N + n);
                const float zo = a.z[zi];
                const float x0 = ca * zo - cs_ * v;
                float zn = csx * zo + c0 * x0 + c1 * (x0 - a.x0p[zi]);
                if (a.sde_noise) zn += cn * a.sde_noise[zi];
                a.x0p[zi] = x0;
                a.z[zi] = zn;
                a.z[zi + (unsigned)(nc * a.N)] = zn;
            }
        }
        return;
    }
    if constexpr (DUAL) {
        typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (__bf16)(p16_silu(acc[0][r]) * acc[1][r]);
        // element (t = frow, n) of the packed output: tile n >> 5, lane frow + 16 * ((n & 31) >> 3), slot n & 7
        const int64_t otile = n0 >> 5;
        const int ol = frow + 16 * ((n0 & 31) >> 3);
        *reinterpret_cast<uint2*>(a.Yp + ((otile * 64 + ol) * 16 + (n0 & 7) * 2)) = __builtin_bit_cast(uint2, o);
    } else {
        float o[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
        const float pb[4] = {pre_b.x, pre_b.y, pre_b.z, pre_b.w};
        const float py[4] = {pre_y.x, pre_y.y, pre_y.z, pre_y.w};
        const float pg[4] = {pre_g.x, pre_g.y, pre_g.z, pre_g.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if constexpr (EPI == VV_EPI_BIAS) o[r] += pb[r];
            else if constexpr (EPI == VV_EPI_RESID) o[r] = py[r] + (o[r] + pb[r]);
            else o[r] = py[r] + pg[r] * o[r];
        }
        *reinterpret_cast<float4*>(a.Y + (size_t)frow * a.ldy + n0) = float4{o[0], o[1], o[2], o[3]};
    }
}

}  // namespace

extern "C" {

// rows [T][K] fp32 (T <= 16) -> one packed 16-row tile.  mode 0: plain conversion; 1: RMSNorm (nw may be null: no affine);
// 2: adaLN-modulated RMSNorm (sc / sh rows)
int vv_pack16_launch(const float* x, int ldx, int mode, const float* nw, float eps, const float* sc, const float* sh, int ld_mod,
                     void* xp, int T, int K, hipStream_t s) {
    if (T < 1 || T > 16 || (K & 31) || K > 8192 || (ldx & 3) || (((uintptr_t)x) & 15) || mode < 0 || mode > 2 ||
        (mode == 2 && ((ld_mod & 3) || !sc || !sh))) return -1;
    if (mode == 2) hipLaunchKernelGGL((vv_pack16_kernel<2>), dim3(16), dim3(256), 0, s, x, ldx, nw, eps, sc, sh, ld_mod, (unsigned char*)xp, T, K);
    else if (mode == 1) hipLaunchKernelGGL((vv_pack16_kernel<1>), dim3(16), dim3(256), 0, s, x, ldx, nw, eps, sc, sh, ld_mod, (unsigned char*)xp, T, K);
    else hipLaunchKernelGGL((vv_pack16_kernel<0>), dim3(16), dim3(256), 0, s, x, ldx, nw, eps, sc, sh, ld_mod, (unsigned char*)xp, T, K);
    return vv_launch_rc(0);
}

int vv_pack16_tiles_launch(const float* x, int ldx, int64_t stride_outer, int n_inner, int64_t stride_inner, void* xp, int64_t tile_bytes,
                           int T, int K, int n_tiles, hipStream_t s) {
    if (T < 1 || T > 16 || (K & 31) || (ldx & 3) || (stride_outer & 3) || (stride_inner & 3) || (((uintptr_t)x) & 15) || n_inner < 1 || n_tiles < 1 ||
        n_tiles > 65535 || (tile_bytes & 15)) return -1;
    hipLaunchKernelGGL(vv_pack16_tiles_kernel, dim3(16, n_tiles), dim3(256), 0, s, x, ldx, stride_outer, n_inner, stride_inner, (unsigned char*)xp, tile_bytes, T, K);
    return vv_launch_rc(0);
}

// Y / Yp (op)= W . Xp for one packed 16-row activation tile.  flags: 1 = RS (operand un-normalised, row scale from ssq_in), 2 = SH (second
// operand Xs), 4 = PK (residual epilogue also packs the new rows for the next projection).  Returns -3 when the combination has no instantiation.
int vv_gemv16p_launch2(const VVGemv16p* ap, int epi, int flags, hipStream_t s) {
    const VVGemv16p& a = *ap;
    if (a.T < 1 || a.T > 16 || (a.N & 3) || a.K < 32 || (a.K & 31)) return -3;
    const int n_tiles = (a.N + 15) / 16;
    if ((flags & 1) && (!a.ssq_in || a.ssq_tiles < 1 || (((uintptr_t)a.ssq_in) & 15))) return -3;
    if ((flags & 2) && !a.Xs) return -3;
    if ((flags & 4) && (!a.Yp || !a.ssq_out || !a.Y || (a.N & 15) || (a.ldy & 3) || (a.pk_sc && (a.ld_pk & 3)))) return -3;
    // waves per workgroup as in vv_gemv_launch: 4 once there are more tiles than CUs (every workgroup resident at once), else 8
    constexpr int wide_tiles = 256;
    const bool w4 = n_tiles > wide_tiles;
#define VV_P(E_, RS_, SH_, PK_) do { if (w4) hipLaunchKernelGGL((vv_gemv16p_kernel<E_, 4, RS_, SH_, PK_>), dim3(n_tiles), dim3(256), 0, s, a); \
                                   else hipLaunchKernelGGL((vv_gemv16p_kernel<E_, 8, RS_, SH_, PK_>), dim3(n_tiles), dim3(512), 0, s, a); return vv_launch_rc(0); } while (0)
    if (epi == VV_EPI_SWIGLU) {
        if (!a.W2 || !a.Yp || (a.N & 7)) return -3;
        if (flags == 0) VV_P(VV_EPI_SWIGLU, 0, 0, 0);
        if (flags == 1) VV_P(VV_EPI_SWIGLU, 1, 0, 0);
        if (flags == 3) VV_P(VV_EPI_SWIGLU, 1, 1, 0);
    } else if (epi == VV_EPI_BIAS || epi == VV_EPI_STORE) {
        if (!a.Y || (a.ldy & 3)) return -3;
        if (flags == 0) VV_P(VV_EPI_BIAS, 0, 0, 0);
        if (flags == 1) VV_P(VV_EPI_BIAS, 1, 0, 0);
        if (flags == 3) VV_P(VV_EPI_BIAS, 1, 1, 0);
    } else if (epi == VV_EPI_RESID) {
        if (!a.Y || (a.ldy & 3)) return -3;
        if (flags == 0) VV_P(VV_EPI_RESID, 0, 0, 0);
        if (flags == 4) VV_P(VV_EPI_RESID, 0, 0, 1);
    } else if (epi == VV_EPI_GATED_RESID) {
        if (!a.Y || !a.gate || (a.ldy & 3) || (a.ld_gate & 3)) return -3;
        if (flags == 0) VV_P(VV_EPI_GATED_RESID, 0, 0, 0);
        if (flags == 4) VV_P(VV_EPI_GATED_RESID, 0, 0, 1);
    } else if (epi == VV_EPI_CFG_DPM) {
        if (!a.z || !a.x0p || !a.coef || a.n_cfg < 1 || 2 * a.n_cfg != a.T) return -3;
        if (flags == 0) VV_P(VV_EPI_CFG_DPM, 0, 0, 0);
        if (flags == 3) VV_P(VV_EPI_CFG_DPM, 1, 1, 0);
    }
#undef VV_P
    return -3;
}

int vv_gemv16p_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, const float* gate,
                      int T, int N, int K, int ldy, int ld_gate, int epi, hipStream_t s) {
    VVGemv16p a{};
    a.W = (const u32x4*)W; a.W2 = (const u32x4*)W2; a.Xp = (const u32x4*)Xp; a.Y = Y; a.Yp = (unsigned char*)Yp; a.bias = bias; a.gate = gate;
    a.T = T; a.N = N; a.K = K; a.ldy = ldy; a.ld_gate = ld_gate;
    return vv_gemv16p_launch2(&a, epi, 0, s);
}

}  // extern "C"
