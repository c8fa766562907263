// gemv.hip -- latency-lean decode variant of the weight-streaming GEMM (T <= 4 rows).
//
// At decode every dense op of the hot path is a chain of DEPENDENT small launches
// (~450 per frame), so a launch's fixed cost matters as much as its bandwidth.  An
// s_memtime trace of the general kernel (gemm.hip) showed ~7000 of ~8500 cycles of a
// small launch outside the weight stream: 64-bit address arithmetic and guarded-load
// branches in the prologue, a ds_bpermute reduction chain for RMSNorm's sum(x^2), two
// barriers around the split-K reduction, and an epilogue that only then starts loading
// its residual / bias / gate operands.  This kernel keeps the same data path
//   packed bf16 weight tiles --global_load_dwordx4 nt--> VGPR --MFMA 16x16x32--> fp32 acc
//   fp32 activations --float4--> prologue --bf16 split--> wave-private LDS B-fragments
// and restructures everything around it:
//   * one workgroup = one 16-feature tile (two for SwiGLU), its 8 waves split K evenly;
//   * 32-bit offsets from uniform bases (no 64-bit multiplies), k-range handled with
//     clamped loads + one uniform branch per batch instead of a branch per load;
//   * the epilogue wave issues its residual / bias / gate loads at kernel entry;
//   * sum(x^2) by DPP row reductions + readlane (no LDS permutes);
//   * split-K partials go to a dedicated LDS region: a single barrier.
#include "vv_common.h"

#ifdef VV_GEMM_TIMING
#define VV_STAMP(i) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
// per-workgroup wall-clock (100 MHz, chip-wide) entry/exit stamps: the launch's occupancy timeline (tools/gemv_timeline.py)
#define VV_BSTAMP(i) do { if (a.dbg && threadIdx.x == 0 && blockIdx.y * gridDim.x + blockIdx.x < 3200) a.dbg[16 + 2 * (blockIdx.y * gridDim.x + blockIdx.x) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VV_BSTAMP(i) do { } while (0)
#define VV_STAMP(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float silu_f(float u) { return u / (1.0f + __expf(-u)); }
__device__ __forceinline__ float silu_acc(float u) { return u / (1.0f + expf(-u)); }
__device__ __forceinline__ float gelu_erf_f(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752440f)); }

// full-wave sum, result uniform (returned from SGPRs): 4 DPP steps inside each row of 16 + 4 readlanes
__device__ __forceinline__ float wave_sum_dpp(float v) {
    int x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));  // row_half_mirror
    x = __builtin_bit_cast(int, v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true));  // row_mirror
    x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

template <int XS>
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&out)[XS]) {
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
    bf16x4 h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        h[j] = (__bf16)v[j];
        if constexpr (XS > 1) {
            const float r = v[j] - (float)h[j];
            m[j] = (__bf16)r;
            if constexpr (XS > 2) l[j] = (__bf16)(r - (float)m[j]);
        }
    }
    out[0] = __builtin_bit_cast(uint2, h);
    if constexpr (XS > 1) out[1] = __builtin_bit_cast(uint2, m);
    if constexpr (XS > 2) out[2] = __builtin_bit_cast(uint2, l);
}

constexpr int U = 8;       // k-steps per batch (256 k = one float4 per lane per row)

// adaLN-modulated norm, decode rows (MR <= 4): the shift rows ride in the SAME B tile as the modulated activation rows, as columns
// MR..2MR-1 of the MFMA's 16 (a decode launch uses 2..4 of them) -- one operand, one accumulator set, half the ds_reads and MFMAs
// per k-step; the epilogue adds column r + MR to column r with one DPP row shift.  -DVV_MOD_FOLD=0 restores the two-operand form
// (A/B: profiles/r06_mod_fold_ab.json); the 16-row batch form has no spare columns and keeps two operands.
#ifndef VV_MOD_FOLD
#define VV_MOD_FOLD 1
#endif

// PRO / EPI are compile-time: a launch executes only the code of its own prologue/epilogue (the runtime-
// switched version spent a third of a small launch fetching and skipping code it never needed).
// MR = activation rows a launch can carry: 4 for decode steps, 16 for prefill chunks / batched adaLN / the T = 8 codec
// stage (same weight stream, 4x the staging work and LDS).
// WPB = waves per workgroup = K split inside the workgroup.  The launcher picks it so that EVERY workgroup of the launch
// is resident at once (a second dispatch round costs a full load-latency chain): 4 for wide outputs (> 256 tiles),
// 16 for few tiles x long K (the streaming rate of a CU is set by its waves' loads in flight), 8 otherwise.
// PARTS: the activation (prologue side) or residual (epilogue side) tensor arrives as base + 2 part tensors (the producer
// split K over 3 workgroup columns): the three loads are issued together and summed in a fixed order.
// SL: slot-batched rows (VVGemm::sl_*): the rows of one launch are gathered from / scattered to the streaming buffers of up
// to 8 utterances -- one weight pass for the tokenizer stages of a whole batch.  16-row form only.
template <int XS, int PRO, int EPI, int MR, int WPB, int PARTS = 0, int SL = 0>      // PARTS: 0 none, 1 activation side, 2 residual side
// The operands every wave needs before its first load (weight / activation bases, shape, strides) are separate leading
// scalar parameters: with -mllvm -amdgpu-kernarg-preload-count=16 the dispatcher delivers them in SGPRs at wave launch,
// so the first addresses do not wait for a scalar-cache round trip; the rest of VVGemm is fetched by one s_load batch.
__global__ __launch_bounds__(WPB * 64) void vv_gemv_kernel(const u32x4* __restrict__ pW, const u32x4* __restrict__ pW2,
                                                           const float* __restrict__ pX, float* __restrict__ pY,
                                                           const float* __restrict__ pnw, int pT, int pN, int pK, int pldx,
                                                           int pldy, const VVGemm a) {
    constexpr bool DUAL = (EPI == VV_EPI_SWIGLU);
    constexpr int NM = DUAL ? 2 : 1;
    // LDS: [wave][XS][U][4][MR] x 16 B staging tiles, then [wave][NM][64] f32x4 partials, then [wave][MR] ssq
    // adaLN-modulated norm: y = rs * W.(x*nw*(1+scale)) + W.shift -- two B operands and two accumulator sets, so the
    // 1/rms of the row is only needed in the epilogue (as for plain RMSNorm) and no pre-pass over x exists
    constexpr bool FOLD = VV_MOD_FOLD && (PRO == VV_PRO_RMS_MOD) && MR <= 4;
    constexpr int NOP = (PRO == VV_PRO_RMS_MOD && !FOLD) ? 2 : 1;
    constexpr int MRS = FOLD ? 2 * MR : MR;         // staged B columns: activation rows, then (folded form) their shift rows
    // one (k-step, k-group) plane of the staging tile = MR rows x 16 B; the 16-row form pads it by 16 B so that the planes
    // one staging store touches fall into different bank groups (unpadded: 256-B stride = the same 4 banks, 16-way conflict)
    constexpr int GSB = MRS * 16 + (MR == 16 ? 16 : 0);
    __shared__ __attribute__((aligned(16))) unsigned char stg_all[WPB * NOP * XS * U * 4 * GSB];
    __shared__ f32x4 red[WPB][NM * NOP][64];
    __shared__ float ssq_sh[WPB][MR];
    __shared__ float ssq1_sh[PRO == VV_PRO_NORMDW ? WPB : 1];
    // Pull every kernel argument into SGPRs with ONE batch of s_loads: left alone the compiler fetches
    // them lazily behind branches, i.e. 3-4 dependent ~600-cycle round trips on a launch's critical path.
    asm volatile("" ::"s"(a.mod_scale), "s"(a.mod_shift), "s"(a.addvec), "s"(a.bias), "s"(a.nscale), "s"(a.gate));
    asm volatile("" ::"s"(a.ld_mod), "s"(a.ld_gate), "s"(a.x_row_mod), "s"(a.add_rows_per_vec),
                 "s"(a.eps), "s"(a.z), "s"(a.x0p), "s"(a.coef), "s"(a.cfg), "s"(a.n_cfg));
    asm volatile("" ::"s"(a.yparts), "s"(a.xa), "s"(a.ya), "s"(a.n_xa), "s"(a.n_ya), "s"(a.part_stride));
    if constexpr (EPI == VV_EPI_CFG_DPM) asm volatile("" ::"s"(a.sde_noise));
    VV_STAMP(0);
    VV_BSTAMP(0);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // 16-row form: grid.y walks 16-row tiles of a tall activation (tokenizer stages, prefill chunks)
    const int t_base = (MR == 16) ? (int)blockIdx.y * 16 : 0;
    const int T = min(MR, pT - t_base);
    const unsigned tile = blockIdx.x;
    const unsigned k_tiles = (unsigned)(pK + 31) >> 5;
    // K range of this workgroup (grid.y splits K for few-tile x long-K shapes so that all CUs stream), then of this wave
    const unsigned KSB = (MR <= 4) ? gridDim.y : 1u;
    const unsigned ksb = (MR <= 4) ? blockIdx.y : 0u;
    const unsigned kchunk = (k_tiles + KSB - 1) / KSB;
    const unsigned kb0 = min(k_tiles, ksb * kchunk), kb1 = min(k_tiles, kb0 + kchunk);
    const unsigned kper = (kb1 - kb0 + WPB - 1) / WPB;
    const unsigned kt0 = kb0 + wave * kper;
    const unsigned kt1 = min(kb1, kt0 + kper);
    const bool has_k = kt0 < kt1;
    const int frow = lane & 15, fq = lane >> 4;
    unsigned char* stg = stg_all + (size_t)wave * (NOP * XS * U * 4 * GSB);
    const unsigned kk = lane * 4;
    const unsigned st_off = ((kk >> 5) * 4 + ((kk & 31) >> 3)) * GSB + (kk & 7) * 2;

    const u32x4* wbase = pW + (size_t)tile * k_tiles * 64 + lane;
    const u32x4* wbase2 = DUAL ? pW2 + (size_t)tile * k_tiles * 64 + lane : nullptr;

    // slot-batched rows: per-row activation offsets (wave-uniform), computed once
    unsigned xoff_sl[SL ? MR : 1];
    if constexpr (SL) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const int rg = t_base + r;
            const int j = rg / a.sl_T, tt = rg - j * a.sl_T;
            xoff_sl[r] = (r < T) ? (a.sl_x ? (unsigned)(vv_slot_id(a.sl_id, j) * a.sl_x + tt * pldx) : (unsigned)(rg * pldx)) : 0u;
        }
    }
    constexpr int MODR = (PRO == VV_PRO_RMS_MOD) ? MR : 1;
    constexpr int ADDR = (PRO == VV_PRO_ADD_SILU) ? MR : 1;
    constexpr int PR = (PARTS == 1) ? MR : 1;
    constexpr int DWH = (PRO == VV_PRO_NORMDW) ? 6 : 1, DWT = (PRO == VV_PRO_NORMDW) ? 7 : 1;
    struct XR { float4 x[MR]; float4 p0[PR]; float4 p1[PR]; float4 sc[MODR]; float4 sh[MODR]; float4 addv[ADDR]; float4 nwv;
                float4 dh[DWH]; float4 dt[DWT]; float4 db, dg, dn; };
    auto x_load = [&](unsigned ktb, XR& R) {
        unsigned k = ktb * 32 + kk;
        const bool kin = k < min(kt1 * 32, (unsigned)pK);
        if (!kin) k = 0;                                   // clamped: always a legal address, masked later
        R.nwv = pnw ? *reinterpret_cast<const float4*>(pnw + k) : float4{1.f, 1.f, 1.f, 1.f};
        if constexpr (PRO == VV_PRO_NORMDW) {          // one row: history rows, taps, bias, layer scale, first norm's weight
#pragma unroll
            for (int j = 0; j < 6; ++j) R.dh[j] = *reinterpret_cast<const float4*>(a.dw_hist + (unsigned)(j * pK) + k);
#pragma unroll
            for (int j = 0; j < 7; ++j) R.dt[j] = *reinterpret_cast<const float4*>(a.dw_w + (unsigned)(j * pK) + k);
            R.db = *reinterpret_cast<const float4*>(a.dw_b + k);
            R.dg = *reinterpret_cast<const float4*>(a.dw_gamma + k);
            R.dn = *reinterpret_cast<const float4*>(a.dw_nw + k);
        }
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < T) {
                const int rg = t_base + r;
                const int xr_idx = a.x_row_mod > 0 ? rg % a.x_row_mod : rg;
                if constexpr (SL) R.x[r] = *reinterpret_cast<const float4*>(pX + xoff_sl[r] + k);
                else R.x[r] = *reinterpret_cast<const float4*>(pX + (unsigned)(xr_idx * pldx) + k);
                if constexpr (PARTS == 1) {
                    R.p0[r] = *reinterpret_cast<const float4*>(a.xa + (unsigned)(xr_idx * pldx) + k);
                    R.p1[r] = *reinterpret_cast<const float4*>(a.xa + (unsigned)(a.part_stride + xr_idx * pldx) + k);
                }
                if constexpr (PRO == VV_PRO_ADD_SILU) {
                    const int av = a.add_rows_per_vec > 0 ? rg / a.add_rows_per_vec : 0;
                    R.addv[r] = *reinterpret_cast<const float4*>(a.addvec + (unsigned)(av * pK) + k);
                }
                if constexpr (PRO == VV_PRO_RMS_MOD) {
                    R.sc[r] = *reinterpret_cast<const float4*>(a.mod_scale + (unsigned)(rg * a.ld_mod) + k);
                    R.sh[r] = *reinterpret_cast<const float4*>(a.mod_shift + (unsigned)(rg * a.ld_mod) + k);
                }
            }
        }
    };
    auto w_load = [&](unsigned ktb, u32x4 (&dst)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned kt = min(ktb + u, kt1 - 1);      // clamped: tail k-steps re-read the last tile, MFMA skipped
            dst[u][0] = __builtin_nontemporal_load(wbase + kt * 64);
            if constexpr (DUAL) dst[u][1] = __builtin_nontemporal_load(wbase2 + kt * 64);
        }
    };
    // ---- first activation batch and first weight batch go out before anything else ----
    XR R;
    u32x4 wA[U][NM], wB[U][NM];
    if (has_k) { x_load(kt0, R); w_load(kt0, wA); }      // x first: its wait must not drag the weight stream along
    __builtin_amdgcn_sched_barrier(0);
    VV_STAMP(1);
    // PRO_NORMDW: 1/rms of the INPUT row is needed before anything can be staged (it sits inside the conv's newest tap): every
    // wave sums its own k-range of x (L2-resident, two loads per lane at C = 2048), one barrier -- under the first weight batch
    float rs1 = 1.0f;
    if constexpr (PRO == VV_PRO_NORMDW) {
        float s1 = 0.f;
        for (unsigned ktb = kt0; ktb < kt1; ktb += U) {
            const unsigned k = ktb * 32 + kk;
            if (k < min(kt1 * 32, (unsigned)pK)) {
                const float4 v = *reinterpret_cast<const float4*>(pX + k);
                s1 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            }
        }
        s1 = wave_sum_dpp(s1);
        if (lane == 0) ssq1_sh[wave] = s1;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) tot += ssq1_sh[w];
        rs1 = rsqrtf(tot / (float)pK + a.eps);
    }

    // ---- epilogue operands: requested now, consumed ~one weight stream later (wave 0 only) ----
    const int n0 = tile * 16 + fq * 4;
    const bool epi_lane = (wave == 0) && frow < T && n0 < pN;
    float4 pre_y = {0.f, 0.f, 0.f, 0.f}, pre_b = {0.f, 0.f, 0.f, 0.f}, pre_g = {1.f, 1.f, 1.f, 1.f};
    float4 pre_y0 = {0.f, 0.f, 0.f, 0.f}, pre_y1 = {0.f, 0.f, 0.f, 0.f};
    unsigned yrow_off = (unsigned)((t_base + frow) * pldy);
    if constexpr (SL) {
        const int rg = min(t_base + frow, pT - 1);
        const int j = rg / a.sl_T, tt = rg - j * a.sl_T;
        yrow_off = a.sl_y ? (unsigned)(vv_slot_id(a.sl_id, j) * a.sl_y + tt * pldy) : (unsigned)(rg * pldy);
    }
    if (epi_lane) {            // N % 4 == 0 and 16-B aligned operands are launch preconditions (vv_gemv_ok)
        if constexpr (EPI == VV_EPI_BIAS || EPI == VV_EPI_BIAS_GELU || EPI == VV_EPI_RESID) {
            if (a.bias) pre_b = *reinterpret_cast<const float4*>(a.bias + n0);
        }
        if constexpr (EPI == VV_EPI_RESID || EPI == VV_EPI_GATED_RESID) {
            pre_y = *reinterpret_cast<const float4*>(pY + (yrow_off + (unsigned)n0));
            if constexpr (PARTS == 2) {
                const float* yp0 = a.ya + (unsigned)((t_base + frow) * pldy + n0);
                pre_y0 = *reinterpret_cast<const float4*>(yp0);
                pre_y1 = *reinterpret_cast<const float4*>(yp0 + a.part_stride);
            }
            if constexpr (EPI == VV_EPI_GATED_RESID) pre_g = *reinterpret_cast<const float4*>(a.gate + (unsigned)((t_base + frow) * a.ld_gate + n0));
            else if (a.nscale) pre_g = *reinterpret_cast<const float4*>(a.nscale + n0);
        }
    }

    f32x4 acc[NM * NOP];            // [NM, 2NM): the shift operand's products (RMS_MOD)
#pragma unroll
    for (int i = 0; i < NM * NOP; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssq[MR];
#pragma unroll
    for (int r = 0; r < MR; ++r) ssq[r] = 0.f;

    auto x_stage = [&](unsigned ktb, const XR& R) {
        const unsigned k = ktb * 32 + kk;
        const float msk = (k < min(kt1 * 32, (unsigned)pK)) ? 1.f : 0.f;
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            if (r < T) {
                float v[4] = {R.x[r].x, R.x[r].y, R.x[r].z, R.x[r].w};
                if constexpr (PARTS == 1) {
                    v[0] = (v[0] + R.p0[r].x) + R.p1[r].x; v[1] = (v[1] + R.p0[r].y) + R.p1[r].y;
                    v[2] = (v[2] + R.p0[r].z) + R.p1[r].z; v[3] = (v[3] + R.p0[r].w) + R.p1[r].w;
                }
                v[0] *= msk; v[1] *= msk; v[2] *= msk; v[3] *= msk;
                if constexpr (PRO == VV_PRO_RMS) {
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] *= R.nwv.x; v[1] *= R.nwv.y; v[2] *= R.nwv.z; v[3] *= R.nwv.w;
                } else if constexpr (PRO == VV_PRO_NORMDW) {
                    if (r == 0) {
                        const float xn[4] = {R.dn.x, R.dn.y, R.dn.z, R.dn.w}, bb[4] = {R.db.x, R.db.y, R.db.z, R.db.w};
                        const float gg[4] = {R.dg.x, R.dg.y, R.dg.z, R.dg.w}, n2[4] = {R.nwv.x, R.nwv.y, R.nwv.z, R.nwv.w};
                        float hn[4], xo[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hn[e] = v[e] * rs1 * xn[e];                    // the new normed row (the conv's newest input)
                            float acc = bb[e];
#pragma unroll
                            for (int j = 0; j < 6; ++j) acc += reinterpret_cast<const float*>(&R.dt[j])[e] * reinterpret_cast<const float*>(&R.dh[j])[e];
                            acc += reinterpret_cast<const float*>(&R.dt[6])[e] * hn[e];
                            xo[e] = (v[e] + gg[e] * acc) * msk;
                        }
                        if (tile == 0 && msk != 0.f) {                     // one workgroup publishes the block's intermediate rows
                            *reinterpret_cast<float4*>(a.dw_xout + k) = float4{xo[0], xo[1], xo[2], xo[3]};
                            *reinterpret_cast<float4*>(a.dw_hnew + k) = float4{hn[0], hn[1], hn[2], hn[3]};
                        }
                        ssq[r] += xo[0] * xo[0] + xo[1] * xo[1] + xo[2] * xo[2] + xo[3] * xo[3];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = xo[e] * n2[e];
                    }
                } else if constexpr (PRO == VV_PRO_RMS_MOD) {
                    ssq[r] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] = (v[0] * R.nwv.x) * (1.f + R.sc[r].x); v[1] = (v[1] * R.nwv.y) * (1.f + R.sc[r].y);
                    v[2] = (v[2] * R.nwv.z) * (1.f + R.sc[r].z); v[3] = (v[3] * R.nwv.w) * (1.f + R.sc[r].w);
                    float sh4[4] = {R.sh[r].x * msk, R.sh[r].y * msk, R.sh[r].z * msk, R.sh[r].w * msk};
                    uint2 sparts[XS];
                    split4<XS>(sh4, sparts);
#pragma unroll
                    for (int p = 0; p < XS; ++p) {
                        if constexpr (FOLD) *reinterpret_cast<uint2*>(stg + p * (U * 4 * GSB) + st_off + (MR + r) * 16) = sparts[p];
                        else *reinterpret_cast<uint2*>(stg + (XS + p) * (U * 4 * GSB) + st_off + r * 16) = sparts[p];
                    }
                } else if constexpr (PRO == VV_PRO_ADD_SILU) {
                    v[0] = silu_acc(v[0] + R.addv[r].x) * msk; v[1] = silu_acc(v[1] + R.addv[r].y) * msk;
                    v[2] = silu_acc(v[2] + R.addv[r].z) * msk; v[3] = silu_acc(v[3] + R.addv[r].w) * msk;
                }
                uint2 parts[XS];
                split4<XS>(v, parts);
#pragma unroll
                for (int p = 0; p < XS; ++p)
                    *reinterpret_cast<uint2*>(stg + p * (U * 4 * GSB) + st_off + r * 16) = parts[p];
            }
        }
    };
    auto mma = [&](unsigned ktb, const u32x4 (&wb)[U][NM]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ktb + u < kt1) {
#pragma unroll
                for (int p = 0; p < XS; ++p) {
                    u32x4 f = u32x4{0u, 0u, 0u, 0u};
                    if (frow < MRS) f = *reinterpret_cast<const u32x4*>(stg + (size_t)((p * U + u) * 4 + fq) * GSB + frow * 16);
                    const bf16x8 xb = __builtin_bit_cast(bf16x8, f);
#pragma unroll
                    for (int i = 0; i < NM; ++i)
                        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[u][i]), xb, acc[i], 0, 0, 0);
                    if constexpr (NOP == 2) {
                        u32x4 f2 = u32x4{0u, 0u, 0u, 0u};
                        if (frow < MR) f2 = *reinterpret_cast<const u32x4*>(stg + (size_t)(((XS + p) * U + u) * 4 + fq) * GSB + frow * 16);
                        const bf16x8 sb = __builtin_bit_cast(bf16x8, f2);
#pragma unroll
                        for (int i = 0; i < NM; ++i)
                            acc[NM + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[u][i]), sb, acc[NM + i], 0, 0, 0);
                    }
                }
            }
        }
    };

    if (has_k) {
        x_stage(kt0, R);
        VV_STAMP(2);
        // two batches per trip, ping-ponging the weight buffers: no register copies, so the prefetched batch
        // stays in flight across the MFMAs of the current one
        if constexpr (DUAL) {
            // two weight streams: a second weight buffer costs 64 VGPRs and halves the resident workgroups per CU
            // (1 instead of 2); run single-buffered and let the other resident waves cover the load latency
#pragma unroll 1
            for (unsigned ktb = kt0; ktb < kt1; ktb += U) {
                const bool n1 = ktb + U < kt1;
                if (n1) x_load(ktb + U, R);
                mma(ktb, wA);
                if (n1) { w_load(ktb + U, wA); x_stage(ktb + U, R); }
            }
        } else
#pragma unroll 1
        for (unsigned ktb = kt0; ktb < kt1; ktb += 2 * U) {
            const bool n1 = ktb + U < kt1, n2 = ktb + 2 * U < kt1;
            if (n1) { x_load(ktb + U, R); w_load(ktb + U, wB); }
            mma(ktb, wA);
            if (n1) x_stage(ktb + U, R);
            if (n2) { x_load(ktb + 2 * U, R); w_load(ktb + 2 * U, wA); }
            if (n1) mma(ktb + U, wB);
            if (n2) x_stage(ktb + 2 * U, R);
        }
    }
    // rows beyond T were never staged: their fragment slots hold stale LDS -> D columns >= T are garbage, never stored.

    VV_STAMP(3);
    // ---- split-K partials -> LDS, one barrier, wave 0 finishes ----
#pragma unroll
    for (int i = 0; i < NM * NOP; ++i) red[wave][i][lane] = acc[i];
    if constexpr (PRO == VV_PRO_RMS || PRO == VV_PRO_RMS_MOD || PRO == VV_PRO_NORMDW) {
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const float s = (r < T) ? wave_sum_dpp(ssq[r]) : 0.f;
            if (lane == 0) ssq_sh[wave][r] = s;
        }
    }
    VV_STAMP(4);
    __syncthreads();
    VV_STAMP(5);
    if (wave != 0) return;
#pragma unroll
    for (int w = 1; w < WPB; ++w)
#pragma unroll
        for (int i = 0; i < NM * NOP; ++i) acc[i] += red[w][i][lane];
    f32x4 shf[FOLD ? NM : 1];                      // folded form: W.shift of row r sits in column r + MR -> row_shl:MR brings it to lane r
    if constexpr (FOLD) {
#pragma unroll
        for (int i = 0; i < NM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float av = acc[i][r];        // through a scalar temporary: bit_cast applied to a vector ELEMENT reads element 0 under this clang
                const int sv = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, av), 0x100 + MR, 0xF, 0xF, true);
                shf[i][r] = __builtin_bit_cast(float, sv);
            }
    }
    if (!epi_lane) return;
    float rs = 1.0f;
    if constexpr (PRO == VV_PRO_RMS || PRO == VV_PRO_RMS_MOD || PRO == VV_PRO_NORMDW) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) s += ssq_sh[w][frow];
        rs = rsqrtf(s / (float)pK + a.eps);
    }
    float o[4] = {acc[0][0] * rs, acc[0][1] * rs, acc[0][2] * rs, acc[0][3] * rs};
    float up[4] = {0.f, 0.f, 0.f, 0.f};            // SwiGLU: the "up" half
    if constexpr (DUAL) { up[0] = acc[1][0] * rs; up[1] = acc[1][1] * rs; up[2] = acc[1][2] * rs; up[3] = acc[1][3] * rs; }
    if constexpr (NOP == 2) {                      // + W.shift
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] += acc[NM][r]; if constexpr (DUAL) up[r] += acc[NM + 1][r]; }
    }
    if constexpr (FOLD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { o[r] += shf[0][r]; if constexpr (DUAL) up[r] += shf[1][r]; }
    }
    const float pb[4] = {pre_b.x, pre_b.y, pre_b.z, pre_b.w};
    const float py[4] = {(pre_y.x + pre_y0.x) + pre_y1.x, (pre_y.y + pre_y0.y) + pre_y1.y, (pre_y.z + pre_y0.z) + pre_y1.z, (pre_y.w + pre_y0.w) + pre_y1.w};
    const float pg[4] = {pre_g.x, pre_g.y, pre_g.z, pre_g.w};
    if constexpr (EPI == VV_EPI_BIAS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += pb[r];
    } else if constexpr (EPI == VV_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = gelu_erf_f(o[r] + pb[r]);
    } else if constexpr (EPI == VV_EPI_SWIGLU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu_acc(o[r]) * up[r];
    } else if constexpr (EPI == VV_EPI_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (ksb == 0) ? py[r] + pg[r] * (o[r] + pb[r]) : pg[r] * o[r];
    } else if constexpr (EPI == VV_EPI_GATED_RESID) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (ksb == 0) ? py[r] + pg[r] * o[r] : pg[r] * o[r];
    }
    if constexpr (EPI == VV_EPI_CFG_DPM) {
        const int nc = a.n_cfg;
        const float ca = a.coef[0], cs_ = a.coef[1], csx = a.coef[2], c0 = a.coef[3], c1 = a.coef[4];
        const float cn = a.sde_noise ? a.coef[5] : 0.f;          // sde-dpmsolver++: + cn * eps_i (dpm_solver.py:680-686, 785-793)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float vu = __shfl(o[r], lane + nc);
            const int n = n0 + r;
            if (frow < nc && n < pN) {
                const float v = vu + a.cfg * (o[r] - vu);
                const unsigned zi = (unsigned)(frow * pN + n);
                const float zo = a.z[zi];
                const float x0 = ca * zo - cs_ * v;
                float zn = csx * zo + c0 * x0 + c1 * (x0 - a.x0p[zi]);
                if (a.sde_noise) zn += cn * a.sde_noise[zi];
                a.x0p[zi] = x0;
                a.z[zi] = zn;
                a.z[zi + (unsigned)(nc * pN)] = zn;
            }
        }
        return;
    }
    float* yp = (ksb == 0 ? pY : a.yparts + (unsigned)((ksb - 1) * a.part_stride)) + (yrow_off + (unsigned)n0);
    *reinterpret_cast<float4*>(yp) = float4{o[0], o[1], o[2], o[3]};
    VV_STAMP(6);
    VV_BSTAMP(1);
}

}  // namespace

static bool gemv_combo_ok(int pro, int epi, bool wide);
// pairs with a slot-batched form: strided conv / transposed conv (bias), FFN1 (RMSNorm + bias + GELU), FFN2 (layer scale + residual)
#define VV_GEMV_SL(X) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_RESID) X(VV_PRO_RMS, VV_EPI_BIAS_GELU)
static bool gemv_parts_ok(int pro, int epi, bool xside);
// Eligibility: decode rows, aligned operands, 32-bit offsets, a specialised (prologue, epilogue) pair.
extern "C" int vv_gemv_ok(const VVGemm* a) {
    if (a->T < 1) return 0;
    if (!gemv_combo_ok(a->pro, a->epi, a->T > 4 || a->sl_n > 0)) return 0;
    if (a->sl_n > 0) {      // slot-batched rows: the three tokenizer pairs, plain operands, 32-bit offsets inside every slot buffer
        if (a->sl_n > 8 || a->sl_T < 1 || a->T != a->sl_n * a->sl_T || a->sl_x < 0 || a->sl_y < 0 || (a->sl_x & 3) || (a->sl_y & 3)) return 0;
        if (a->kgrid > 1 || a->n_xa || a->n_ya || a->x_row_mod > 0 || a->add_rows_per_vec > 0) return 0;
        bool pair = false;
#define X(P, E) if (a->pro == P && a->epi == E) pair = true;
        VV_GEMV_SL(X)
#undef X
        if (!pair) return 0;
        for (int j = 0; j < a->sl_n; ++j)
            if (a->sl_id[j] < 0 || (int64_t)a->sl_id[j] * a->sl_x + (int64_t)a->sl_T * a->ldx >= (1LL << 30) ||
                (int64_t)a->sl_id[j] * a->sl_y + (int64_t)a->sl_T * a->ldy >= (1LL << 30)) return 0;
    }
    if ((a->x_row_mod > 0 || a->add_rows_per_vec > 0) && a->pro != VV_PRO_ADD_SILU) return 0;
    if (a->pro == VV_PRO_NORMDW) {     // one row, whole K inside every workgroup, every operand present and 16-B aligned
        if (a->T != 1 || a->sl_n > 0 || a->kgrid > 1 || a->n_xa || a->n_ya || !a->dw_hist || !a->dw_w || !a->dw_b || !a->dw_gamma ||
            !a->dw_nw || !a->dw_xout || !a->dw_hnew || a->dw_xout == a->X) return 0;
        if ((((uintptr_t)a->dw_hist) | ((uintptr_t)a->dw_w) | ((uintptr_t)a->dw_b) | ((uintptr_t)a->dw_gamma) | ((uintptr_t)a->dw_nw) |
             ((uintptr_t)a->dw_xout) | ((uintptr_t)a->dw_hnew)) & 15) return 0;
        if ((int64_t)7 * a->K >= (1LL << 30)) return 0;
    }
    if ((a->K & 3) || (a->ldx & 3) || (((uintptr_t)a->X) & 15)) return 0;
    if (a->pro == VV_PRO_RMS_MOD && (a->ld_mod & 3)) return 0;
    if (a->nw && (((uintptr_t)a->nw) & 15)) return 0;
    if (a->K < 32) return 0;
    if ((int64_t)a->T * a->ldy >= (1LL << 30) || (int64_t)a->T * a->ldx >= (1LL << 30)) return 0;
    if ((a->N & 3) || (a->ldy & 3)) return 0;
    if ((((uintptr_t)a->Y) & 15) || (a->bias && (((uintptr_t)a->bias) & 15))) return 0;
    if (a->nscale && (((uintptr_t)a->nscale) & 15)) return 0;
    if (a->epi == VV_EPI_GATED_RESID && ((a->ld_gate & 3) || (((uintptr_t)a->gate) & 15))) return 0;
    if (a->pro == VV_PRO_RMS_MOD && ((((uintptr_t)a->mod_scale) & 15) || (((uintptr_t)a->mod_shift) & 15))) return 0;
    if (a->pro == VV_PRO_ADD_SILU && (((uintptr_t)a->addvec) & 15)) return 0;
    if (a->kgrid > 1 && (a->T > 4 || a->kgrid != 3 || a->pro != VV_PRO_NONE || !a->yparts || (a->part_stride & 3) ||
                         (a->epi != VV_EPI_RESID && a->epi != VV_EPI_GATED_RESID) || (((uintptr_t)a->yparts) & 15))) return 0;
    if ((a->n_xa > 0 && (!a->xa || (((uintptr_t)a->xa) & 15))) || (a->n_ya > 0 && (!a->ya || (((uintptr_t)a->ya) & 15)))) return 0;
    if ((a->n_xa != 0 && a->n_xa != 2) || (a->n_ya != 0 && a->n_ya != 2) || (a->n_xa && a->n_ya)) return 0;
    if ((a->n_xa || a->n_ya) && ((a->part_stride & 3) || a->T > 4 || !gemv_parts_ok(a->pro, a->epi, a->n_xa > 0))) return 0;
    return 1;
}

// The (prologue, epilogue) pairs the engine actually issues; anything else runs on the general kernel.
#define VV_GEMV_COMBOS(X)                                                                      \
    X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_RESID)      \
    X(VV_PRO_NONE, VV_EPI_GATED_RESID) X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_RMS, VV_EPI_BIAS_GELU) \
    X(VV_PRO_RMS, VV_EPI_SWIGLU) X(VV_PRO_RMS, VV_EPI_RESID) X(VV_PRO_RMS, VV_EPI_STORE)       \
    X(VV_PRO_RMS_MOD, VV_EPI_SWIGLU) X(VV_PRO_RMS_MOD, VV_EPI_CFG_DPM) X(VV_PRO_RMS_MOD, VV_EPI_STORE) \
    X(VV_PRO_ADD_SILU, VV_EPI_STORE) X(VV_PRO_NORMDW, VV_EPI_BIAS_GELU)
// pairs that also exist in the 16-row form (prefill chunks, batched adaLN, T = 8 codec stage, connectors)
#define VV_GEMV_WIDE(X)                                                                        \
    X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_RESID)      \
    X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_RMS, VV_EPI_BIAS_GELU) X(VV_PRO_RMS, VV_EPI_SWIGLU)    \
    X(VV_PRO_ADD_SILU, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_GATED_RESID)
// 16-row diffusion-head pairs (8 utterances x {cond, uncond} rows): two B operands -> 4-wave workgroups only (LDS)
#define VV_GEMV_WIDE_MOD(X)                                                                    \
    X(VV_PRO_RMS_MOD, VV_EPI_SWIGLU) X(VV_PRO_RMS_MOD, VV_EPI_CFG_DPM) X(VV_PRO_RMS_MOD, VV_EPI_STORE)

static bool gemv_combo_ok(int pro, int epi, bool wide) {
#define X(P, E) if (pro == P && epi == E) return true;
    if (wide) { VV_GEMV_WIDE(X) VV_GEMV_WIDE_MOD(X) } else { VV_GEMV_COMBOS(X) }
#undef X
    return false;
}

// 16-row pairs that also have a 4-wave form (tall tokenizer stages, batched adaLN)
#define VV_GEMV_WIDE4(X)                                                                       \
    X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_RESID) X(VV_PRO_RMS, VV_EPI_BIAS_GELU)
// pairs with a 4-wave form (wide outputs) and a 16-wave form (few tiles, long K); bench mode (xs == 1) only
#define VV_GEMV_W4(X)                                                                          \
    X(VV_PRO_RMS, VV_EPI_SWIGLU) X(VV_PRO_RMS_MOD, VV_EPI_SWIGLU) X(VV_PRO_RMS, VV_EPI_BIAS_GELU) \
    X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_ADD_SILU, VV_EPI_STORE) \
    X(VV_PRO_NORMDW, VV_EPI_BIAS_GELU)
#define VV_GEMV_W16(X)                                                                         \
    X(VV_PRO_NONE, VV_EPI_RESID) X(VV_PRO_NONE, VV_EPI_GATED_RESID) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_STORE)

// pairs that can consume a K-split tensor: on the activation side (x) or on the residual side (y)
#define VV_GEMV_PARTS_X(X) X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_RMS_MOD, VV_EPI_SWIGLU) X(VV_PRO_RMS_MOD, VV_EPI_CFG_DPM) X(VV_PRO_RMS_MOD, VV_EPI_STORE)
#define VV_GEMV_PARTS_Y(X) X(VV_PRO_NONE, VV_EPI_RESID) X(VV_PRO_NONE, VV_EPI_GATED_RESID)
#define VV_GEMV_PARTS(X) VV_GEMV_PARTS_X(X) VV_GEMV_PARTS_Y(X)
static bool gemv_parts_ok(int pro, int epi, bool xside) {
#define X(P, E) if (pro == P && epi == E) return true;
    if (xside) { VV_GEMV_PARTS_X(X) } else { VV_GEMV_PARTS_Y(X) }
#undef X
    return false;
}

extern "C" int vv_gemv_launch(VVGemm a, int xs, hipStream_t s) {
    const int n_tiles = (a.N + 15) / 16, k_tiles = (a.K + 31) / 32;
    if (a.epi == VV_EPI_SWIGLU && !a.W2) return -1;
    dim3 grid(n_tiles);
#define VV_GO(XS_, P, E, MR_, WP_)                                                                      \
    do { hipLaunchKernelGGL((vv_gemv_kernel<XS_, P, E, MR_, WP_>), grid, dim3(WP_ * 64), 0, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);       \
         return vv_launch_rc(0); } while (0)
    if (a.T > 4 || a.sl_n > 0) {
        if (xs > 2) return -3;       // 16-row staging tiles of the exact mode exceed the LDS: general kernel
        grid.y = (a.T + 15) / 16;
        constexpr int wide4_wgs = 128;
        // The 16-row tiles are used by the codec (T = 5..16 rows): above ~half a workgroup per CU the 4-wave form
        // (2x the resident workgroups per CU) wins; measured 3.117 -> 3.053 ms/frame on the 1.5B config for
        // thresholds 64..128 vs 512 (DESIGN.md section 8 lists the sweep).
        if (a.sl_n > 0) {
#define VV_GOSL(XS_, P, E, WP_)                                                                         \
    do { hipLaunchKernelGGL((vv_gemv_kernel<XS_, P, E, 16, WP_, 0, 1>), grid, dim3(WP_ * 64), 0, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);   \
         return vv_launch_rc(0); } while (0)
#define X(P, E) if (a.pro == P && a.epi == E) { if (xs == 2) VV_GOSL(2, P, E, 8); else if ((int64_t)n_tiles * grid.y > wide4_wgs) VV_GOSL(1, P, E, 4); else VV_GOSL(1, P, E, 8); }
            VV_GEMV_SL(X)
#undef X
#undef VV_GOSL
            return -3;
        }
        if (xs == 1 && (int64_t)n_tiles * grid.y > wide4_wgs) {
#define X(P, E) if (a.pro == P && a.epi == E) VV_GO(1, P, E, 16, 4);
            VV_GEMV_WIDE4(X)
#undef X
        }
#define X(P, E) if (a.pro == P && a.epi == E) { if (xs == 1) VV_GO(1, P, E, 16, 8); else VV_GO(2, P, E, 16, 8); }
        VV_GEMV_WIDE(X)
#undef X
#define X(P, E) if (a.pro == P && a.epi == E) { if (xs == 1) VV_GO(1, P, E, 16, 4); else return -3; }
        VV_GEMV_WIDE_MOD(X)
#undef X
        return -3;
    }
    if (a.n_xa > 0 || a.n_ya > 0) {           // consumers of a K-split tensor (decode rows only)
        if (a.kgrid > 1) grid.y = a.kgrid;
#define VV_GOP(XS_, P, E, WP_, S_)                                                                      \
    do { hipLaunchKernelGGL((vv_gemv_kernel<XS_, P, E, 4, WP_, S_>), grid, dim3(WP_ * 64), 0, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);     \
         return vv_launch_rc(0); } while (0)
#define VV_GOP2(P, E)   /* two rows, wide output: the 2-row form of the K-split consumer (7B-width A/B of the column split, round 6) */ \
    do { hipLaunchKernelGGL((vv_gemv_kernel<1, P, E, 2, 4, 1>), grid, dim3(256), 0, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);     \
         return vv_launch_rc(0); } while (0)
#define X(P, E)                                                                                         \
    if (a.pro == P && a.epi == E) {                                                                     \
        if (xs == 1 && n_tiles > 256 && E == VV_EPI_SWIGLU && a.T <= 2) VV_GOP2(P, E);                   \
        if (xs == 1 && n_tiles > 256 && E == VV_EPI_SWIGLU) VV_GOP(1, P, E, 4, 1);                       \
        if (xs == 1) VV_GOP(1, P, E, 8, 1); else if (xs == 2) VV_GOP(2, P, E, 8, 1); else VV_GOP(3, P, E, 8, 1); \
    }
        if (a.n_xa > 0) { VV_GEMV_PARTS_X(X) }
#undef X
#define X(P, E)                                                                                         \
    if (a.pro == P && a.epi == E) {                                                                     \
        if (xs == 1) VV_GOP(1, P, E, 8, 2); else if (xs == 2) VV_GOP(2, P, E, 8, 2); else VV_GOP(3, P, E, 8, 2); \
    }
        if (a.n_ya > 0) { VV_GEMV_PARTS_Y(X) }
#undef X
#undef VV_GOP
#undef VV_GOP2
        return -3;
    }
    if (a.kgrid > 1) grid.y = a.kgrid;
    if (xs == 1 && a.kgrid <= 1) {
        if (n_tiles > 256) {
            // one utterance = two rows (cond + uncond): the 2-row form halves the activation registers and the staging tile, so
            // one more workgroup fits per SIMD (RMS_MOD + SwiGLU: 160 -> <= 128 VGPRs) and a 672-tile launch is resident at once
            if (a.T <= 2) {
#define X(P, E) if (a.pro == P && a.epi == E) VV_GO(1, P, E, 2, 4);
                VV_GEMV_W4(X)
#undef X
            }
#define X(P, E) if (a.pro == P && a.epi == E) VV_GO(1, P, E, 4, 4);
            VV_GEMV_W4(X)
#undef X
        } else if (n_tiles <= 128 && k_tiles >= 96) {
#define X(P, E) if (a.pro == P && a.epi == E) VV_GO(1, P, E, 4, 16);
            VV_GEMV_W16(X)
#undef X
        }
    }
#define X(P, E)                                                                                         \
    if (a.pro == P && a.epi == E) {                                                                     \
        if (xs == 1) VV_GO(1, P, E, 4, 8); else if (xs == 2) VV_GO(2, P, E, 4, 8); else VV_GO(3, P, E, 4, 8); \
    }
    VV_GEMV_COMBOS(X)
#undef X
#undef VV_GO
    return -3;
}
