// headtail.hip -- the seam between two solver steps of the diffusion head, one launch instead of two (decode rows, bf16 mode).
//
// A solver step ends with the final layer (adaLN-modulated norm -> 64 latent outputs, CFG, DPM-Solver++ update of the noisy latent z;
// modular_vibevoice_diffusion_head.py:164-188, schedule/dpm_solver.py:935-1022) and the next one begins with the in-projection of the
// new z (noisy_images_proj: 64 -> H, modular_vibevoice_diffusion_head.py:254-262).  As two launches that is 4 workgroups streaming a
// 0.46 MB matrix, a kernel boundary, then H/16 workgroups with K = 64: ~7.7 + 1.5 + 5.1 + 1.5 us of a chain that has nothing to
// stream.  Here every workgroup computes the final layer itself (W_out is 0.46 MB and L2-resident: the redundancy costs L2 reads, not
// HBM), applies the CFG + solver update in registers, and multiplies its own TPW 16-feature tiles of W_in with the new latent:
//
//   eps[t][j]  = rs_t * sum_k Wout[j][k] x[t][k] (1 + scale[t][k])  +  sum_k Wout[j][k] shift[t][k]        (t < 2n rows, j < 64)
//   v          = eps_u + cfg (eps_c - eps_u);  x0 = a z - s v;  z' = cs z + c0 x0 + c1 (x0 - x0_prev) (+ cn noise)
//   xnext[t][f] = sum_j Win[f][j] bf16(z'[t][j]) + b_in[f]                                                 (f in this workgroup's tiles)
//
// The shift rows ride as extra MFMA columns of the activation tile (columns 4..7), as in gemv.hip's folded form.  State is DOUBLE
// BUFFERED by the caller (z / x0_prev / the residual rows are read by every workgroup and written by one or by their owners: a step
// reads buffer i & 1 and writes the other one), so no workgroup can observe another's update.  Same arithmetic as the two launches it
// replaces (bf16 at the MFMA inputs, fp32 accumulation, the same K split over 8 waves and the same fixed reduction order).
#include "vv_common.h"

namespace {

__device__ __forceinline__ float ht_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

constexpr int TU = 8;      // k-steps per staging pass (256 k = one float4 per lane and row)
constexpr int WB = 4;      // k-steps per weight batch (x 4 latent tiles = 16 fragment loads), two batches in flight
constexpr int KMAX = 16;   // k-tiles per wave the staging tile holds (H <= 8 waves x 16 x 32 = 4096)

// TR = rows the launch can carry (2: one utterance's cond + uncond rows; 4: two utterances): sizes the row registers of the hoisted loads
template <int TPW, int PARTS, int TR>
__global__ __launch_bounds__(512) void vv_head_tail_kernel(const VVTail a) {
    constexpr int WPB = 8, MR = 4, MRS = 8, GSB = MRS * 16, NT = 4;
    // staging: a wave's WHOLE K range (<= KMAX k-tiles) as B fragments, written once before the weight stream starts
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];                       // dynamic: 64 KiB + 2 KiB + 128 B (> the 64 KiB static limit)
    unsigned char* const stg_all = smem;                                                       // [WPB][KMAX][4][GSB]
    static_assert(KMAX * 4 * GSB >= NT * 64 * 16, "a wave's split-K partials re-use its own (by then idle) staging tile");
    unsigned char* const zfr = smem + WPB * KMAX * 4 * GSB;                                    // bf16(z') as the in-projection's B fragments: 2 k-tiles
    float (*ssq_sh)[MR] = reinterpret_cast<float (*)[MR]>(smem + WPB * KMAX * 4 * GSB + 2 * 1024);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int frow = lane & 15, fq = lane >> 4;
    const int T = a.T, H = a.H;
    const unsigned k_tiles = (unsigned)H >> 5;
    const unsigned kper = (k_tiles + WPB - 1) / WPB;
    const unsigned kt0 = wave * kper, kt1 = min(k_tiles, kt0 + kper);
    const unsigned nk = (kt1 > kt0) ? kt1 - kt0 : 0u;
    asm volatile("" ::"s"(a.coef), "s"(a.z_in), "s"(a.x0p_in), "s"(a.sde_noise), "s"(a.z_out), "s"(a.x0p_out), "s"(a.Xout), "s"(a.cfg));
    // ---- first weight batches go out before anything else (L2-resident: 4 latent tiles per k-step) ----
    u32x4 wA[WB][NT], wB[WB][NT];
    auto w_load = [&](unsigned kb, u32x4 (&w)[WB][NT]) {
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            const unsigned kt = min(kt0 + kb + u, kt1 - 1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[u][nt] = a.Wout[((size_t)nt * k_tiles + kt) * 64 + lane];
        }
    };
    // ---- the in-projection's operands of this wave's feature tile and (waves 0..3: one latent tile each) the solver state: requested now, consumed last ----
    const int tile_in = blockIdx.x * TPW + wave;
    const bool has_in = wave < TPW && tile_in * 16 < H;
    u32x4 win[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    float4 bin4 = {0.f, 0.f, 0.f, 0.f};
    const int n_in = tile_in * 16 + fq * 4;
    if (has_in) {
        win[0] = a.Win[((size_t)tile_in * 2 + 0) * 64 + lane];
        win[1] = a.Win[((size_t)tile_in * 2 + 1) * 64 + lane];
        if (a.bin) bin4 = *reinterpret_cast<const float4*>(a.bin + n_in);
    }
    const int nc = a.n_cfg;
    float4 pz = {0.f, 0.f, 0.f, 0.f}, px = {0.f, 0.f, 0.f, 0.f}, pn = {0.f, 0.f, 0.f, 0.f};
    float cf[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (wave < NT) {
#pragma unroll
        for (int i = 0; i < 6; ++i) cf[i] = a.coef[i];
        if (frow < nc) {
            const unsigned zi = (unsigned)(frow * a.L + wave * 16 + fq * 4);
            pz = *reinterpret_cast<const float4*>(a.z_in + zi);
            px = *reinterpret_cast<const float4*>(a.x0p_in + zi);
            if (a.sde_noise) pn = *reinterpret_cast<const float4*>(a.sde_noise + zi);
        }
    }
    // ---- rows of the wave's whole K range -> B fragments (x (1 + scale) in columns 0..3, shift in 4..7), sum of squares on the way.
    // The row loads go out FIRST (loads return in order: the staging below then runs under the weight stream), then two weight batches ----
    unsigned char* stg = stg_all + (size_t)wave * (KMAX * 4 * GSB);
    const unsigned kk = lane * 4;
    const unsigned st_off = ((kk >> 5) * 4 + ((kk & 31) >> 3)) * GSB + (kk & 7) * 2;
    float ssq[MR] = {0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    constexpr int NP = KMAX / TU;
    static_assert(TR == 2 || TR == 4, "rows");
    float4 rx[NP][TR], rsc[NP][TR], rsh[NP][TR];
    float rmsk[NP];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const unsigned ktb = kt0 + pass * TU;
        unsigned k = ktb * 32 + kk;
        const bool kin = ktb < kt1 && k < min(kt1 * 32, (unsigned)H);
        if (!kin) k = 0;
        rmsk[pass] = kin ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < TR; ++r) {
            rx[pass][r] = rsc[pass][r] = rsh[pass][r] = float4{0.f, 0.f, 0.f, 0.f};
            if (r < T && ktb < kt1) {
                float4 x4 = *reinterpret_cast<const float4*>(a.X + (unsigned)(r * H) + k);
                if constexpr (PARTS) {
                    const float4 p0 = *reinterpret_cast<const float4*>(a.xa + (unsigned)(r * H) + k);
                    const float4 p1 = *reinterpret_cast<const float4*>(a.xa + (unsigned)(a.part_stride + r * H) + k);
                    x4.x = (x4.x + p0.x) + p1.x; x4.y = (x4.y + p0.y) + p1.y; x4.z = (x4.z + p0.z) + p1.z; x4.w = (x4.w + p0.w) + p1.w;
                }
                rx[pass][r] = x4;
                rsc[pass][r] = *reinterpret_cast<const float4*>(a.sc + (unsigned)(r * a.ld_mod) + k);
                rsh[pass][r] = *reinterpret_cast<const float4*>(a.sh + (unsigned)(r * a.ld_mod) + k);
            }
        }
    }
    if (nk) { w_load(0, wA); if (nk > WB) w_load(WB, wB); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        const unsigned ktb = kt0 + pass * TU;
        if (ktb < kt1) {
            const float msk = rmsk[pass];
            unsigned char* sp = stg + (size_t)pass * (TU * 4 * GSB);
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                if (r < T) {
                    const float4 x4 = rx[pass][r], s4 = rsc[pass][r], h4 = rsh[pass][r];
                    float v[4] = {x4.x * msk, x4.y * msk, x4.z * msk, x4.w * msk};
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    bf16x4 xb, sb;
                    xb[0] = (__bf16)(v[0] * (1.f + s4.x)); xb[1] = (__bf16)(v[1] * (1.f + s4.y));
                    xb[2] = (__bf16)(v[2] * (1.f + s4.z)); xb[3] = (__bf16)(v[3] * (1.f + s4.w));
                    sb[0] = (__bf16)(h4.x * msk); sb[1] = (__bf16)(h4.y * msk); sb[2] = (__bf16)(h4.z * msk); sb[3] = (__bf16)(h4.w * msk);
                    *reinterpret_cast<uint2*>(sp + st_off + r * 16) = __builtin_bit_cast(uint2, xb);
                    *reinterpret_cast<uint2*>(sp + st_off + (MR + r) * 16) = __builtin_bit_cast(uint2, sb);
                }
            }
        }
    }
    // ---- weight stream: two batches of 4 k-steps in flight, ping-pong ----
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](unsigned kb, const u32x4 (&w)[WB][NT]) {
#pragma unroll
        for (int u = 0; u < WB; ++u) {
            if (kb + u < nk) {
                u32x4 f = u32x4{0u, 0u, 0u, 0u};
                if (frow < MRS) f = *reinterpret_cast<const u32x4*>(stg + (size_t)((kb + u) * 4 + fq) * GSB + frow * 16);
                const bf16x8 xb = __builtin_bit_cast(bf16x8, f);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[u][nt]), xb, acc[nt], 0, 0, 0);
            }
        }
    };
#pragma unroll 1
    for (unsigned kb = 0; kb < nk; kb += 2 * WB) {
        mma(kb, wA);
        if (kb + 2 * WB < nk) w_load(kb + 2 * WB, wA);
        if (kb + WB < nk) mma(kb + WB, wB);
        if (kb + 3 * WB < nk) w_load(kb + 3 * WB, wB);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) reinterpret_cast<f32x4*>(stg)[nt * 64 + lane] = acc[nt];      // wave-private until the barrier
#pragma unroll
    for (int r = 0; r < MR; ++r) {
        const float s = (r < T) ? ht_wave_sum(ssq[r]) : 0.f;
        if (lane == 0) ssq_sh[wave][r] = s;
    }
    __syncthreads();
    if (wave < NT) {
        // waves 0..3 finish one latent tile each: partial sums of the 8 waves in order, 1/rms, + shift (column r + 4), CFG, solver update
        const int nt = wave;
        f32x4 e = reinterpret_cast<const f32x4*>(stg_all)[nt * 64 + lane];
#pragma unroll
        for (int w = 1; w < WPB; ++w) e += reinterpret_cast<const f32x4*>(stg_all + (size_t)w * (KMAX * 4 * GSB))[nt * 64 + lane];
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += ssq_sh[w][frow & 3];
        const float rs = rsqrtf(s / (float)H + a.eps);
        const float ca = cf[0], cs_ = cf[1], csx = cf[2], c0 = cf[3], c1 = cf[4];
        const float cn = a.sde_noise ? cf[5] : 0.f;
        bf16x4 zb;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float av = e[r];                         // scalar temporary: bit_cast of a vector element reads element 0 (clang)
            const int sv = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, av), 0x100 + MR, 0xF, 0xF, true);   // row_shl:4
            const float o = av * rs + __builtin_bit_cast(float, sv);                                                   // eps[row frow][d]
            const int oi = __builtin_bit_cast(int, o);
            // the unconditional row of utterance t sits nc columns further (nc = 1 or 2 decode utterances per launch)
            const int ui = (nc == 1) ? __builtin_amdgcn_update_dpp(0, oi, 0x101, 0xF, 0xF, true) : __builtin_amdgcn_update_dpp(0, oi, 0x102, 0xF, 0xF, true);
            const float vu = __builtin_bit_cast(float, ui);
            const int d = nt * 16 + fq * 4 + r;
            float zn = 0.f;
            if (frow < nc) {
                const float v = vu + a.cfg * (o - vu);
                const unsigned zi = (unsigned)(frow * a.L + d);
                const float zo = reinterpret_cast<const float*>(&pz)[r];
                const float x0 = ca * zo - cs_ * v;
                zn = csx * zo + c0 * x0 + c1 * (x0 - reinterpret_cast<const float*>(&px)[r]);
                if (a.sde_noise) zn += cn * reinterpret_cast<const float*>(&pn)[r];
                if (blockIdx.x == 0) {                      // one workgroup publishes the step's state (into the OTHER buffer)
                    a.x0p_out[zi] = x0;
                    a.z_out[zi] = zn;
                    a.z_out[zi + (unsigned)(nc * a.L)] = zn;
                }
            }
            zb[r] = (__bf16)zn;
        }
        if (frow < nc) {
            // B fragments of the in-projection: element (k = d, column t) at lane t + 16 ((d & 31) >> 3), slot d & 7 of k-tile d >> 5;
            // both CFG halves (columns t and t + n) carry the same latent
            const int d0 = nt * 16 + fq * 4;
            const int off = (((d0 >> 5) * 64 + frow + 16 * ((d0 & 31) >> 3)) * 8 + (d0 & 7)) * 2;
            *reinterpret_cast<uint2*>(zfr + off) = __builtin_bit_cast(uint2, zb);
            *reinterpret_cast<uint2*>(zfr + off + nc * 16) = __builtin_bit_cast(uint2, zb);
        }
    }
    __syncthreads();
    if (!has_in) return;
    f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        u32x4 f = u32x4{0u, 0u, 0u, 0u};
        if (frow < T) f = *reinterpret_cast<const u32x4*>(zfr + ((size_t)kt * 64 + lane) * 16);
        y = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, win[kt]), __builtin_bit_cast(bf16x8, f), y, 0, 0, 0);
    }
    if (frow < T && n_in < H)
        *reinterpret_cast<float4*>(a.Xout + (size_t)frow * H + n_in) = float4{y[0] + bin4.x, y[1] + bin4.y, y[2] + bin4.z, y[3] + bin4.w};
}

}  // namespace

extern "C" int vv_head_tail_ok(const VVTail* a) {
    if (a->T != 2 || a->T != 2 * a->n_cfg ||      /* one utterance's two rows: the 4-row form of this kernel spills (hoisted row loads) and is not built */ a->L != 64 || (a->H & 31) || a->H < 256 || a->H > 8 * 16 * 32) return 0;
    if (!a->Wout || !a->Win || !a->X || !a->sc || !a->sh || !a->Xout || !a->z_in || !a->x0p_in || !a->z_out || !a->x0p_out || !a->coef) return 0;
    if (a->X == a->Xout || a->z_in == a->z_out || a->x0p_in == a->x0p_out) return 0;                // double buffered by contract
    if ((a->ld_mod & 3) || ((((uintptr_t)a->X) | ((uintptr_t)a->sc) | ((uintptr_t)a->sh) | ((uintptr_t)a->Xout)) & 15)) return 0;
    if (a->bin && (((uintptr_t)a->bin) & 15)) return 0;
    if (a->n_xa != 0 && (a->n_xa != 2 || !a->xa || (((uintptr_t)a->xa) & 15) || (a->part_stride & 3))) return 0;
    if ((int64_t)a->T * a->ld_mod >= (1LL << 30) || (int64_t)a->T * a->H >= (1LL << 30)) return 0;
    return 1;
}

// tiles_per_wg: in-projection feature tiles per workgroup (1, 2, 4 or 8): fewer workgroups = fewer redundant passes over W_out
extern "C" int vv_head_tail_launch(const VVTail* ap, int tiles_per_wg, hipStream_t s) {
    if (!vv_head_tail_ok(ap)) return -3;
    const VVTail& a = *ap;
    const int n_tiles = a.H / 16;
    constexpr size_t smem = 8 * KMAX * 4 * 128 + 2 * 1024 + 8 * 4 * sizeof(float);
#define VV_T(TPW_)                                                                                                   \
    do { const dim3 grid((n_tiles + TPW_ - 1) / TPW_);                                                                \
         static bool attr = false;                                                                                    \
         if (!attr) {                                                                                                 \
             (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_head_tail_kernel<TPW_, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
             (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&vv_head_tail_kernel<TPW_, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
             attr = true;                                                                                             \
         }                                                                                                            \
         if (a.n_xa) hipLaunchKernelGGL((vv_head_tail_kernel<TPW_, 1, 2>), grid, dim3(512), smem, s, a);              \
         else hipLaunchKernelGGL((vv_head_tail_kernel<TPW_, 0, 2>), grid, dim3(512), smem, s, a);                     \
         return vv_launch_rc(0); } while (0)
    if (tiles_per_wg == 1) VV_T(1);
    if (tiles_per_wg == 2) VV_T(2);
    if (tiles_per_wg == 4) VV_T(4);
    if (tiles_per_wg == 8) VV_T(8);
#undef VV_T
    return -3;
}
