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
#include "gemv_body.h"

namespace {

template <int XS, int PRO, int EPI, int MR, int WPB, int PARTS = 0, int SL = 0>      // PARTS: 0 none, 1 activation side, 2 residual side
// The operands every wave needs before its first load (weight / activation bases, shape, strides) are separate leading
// scalar parameters: with -mllvm -amdgpu-kernarg-preload-count=16 the dispatcher delivers them in SGPRs at wave launch,
// so the first addresses do not wait for a scalar-cache round trip; the rest of VVGemm is fetched by one s_load batch.
__global__ __launch_bounds__(WPB * 64) void vv_gemv_kernel(const u32x4* __restrict__ pW, const u32x4* __restrict__ pW2,
                                                           const float* __restrict__ pX, float* __restrict__ pY,
                                                           const float* __restrict__ pnw, int pT, int pN, int pK, int pldx,
                                                           int pldy, const VVGemm a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[vv_gemv_smem_bytes<XS, PRO, EPI, MR, WPB>()];
    vv_gemv_body<XS, PRO, EPI, MR, WPB, PARTS, SL, 0>(pW, pW2, pX, pY, pnw, pT, pN, pK, pldx, pldy, a, blockIdx.x, blockIdx.y, gridDim.y, smem, VVChainSync{});
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
    X(VV_PRO_ADD_SILU, VV_EPI_STORE)
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
    X(VV_PRO_RMS, VV_EPI_BIAS) X(VV_PRO_NONE, VV_EPI_STORE) X(VV_PRO_NONE, VV_EPI_BIAS) X(VV_PRO_ADD_SILU, VV_EPI_STORE)
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
         return hipGetLastError() == hipSuccess ? 0 : -2; } while (0)
    if (a.T > 4 || a.sl_n > 0) {
        if (xs > 2) return -3;       // 16-row staging tiles of the exact mode exceed the LDS: general kernel
        grid.y = (a.T + 15) / 16;
        static const int wide4_wgs = getenv("VVHIP_WIDE4_WGS") ? atoi(getenv("VVHIP_WIDE4_WGS")) : 128;
        // The 16-row tiles are used by the codec (T = 5..16 rows): above ~half a workgroup per CU the 4-wave form
        // (2x the resident workgroups per CU) wins; measured 3.117 -> 3.053 ms/frame on the 1.5B config for
        // thresholds 64..128 vs 512 (DESIGN.md section 8 lists the sweep).
        if (a.sl_n > 0) {
#define VV_GOSL(XS_, P, E, WP_)                                                                         \
    do { hipLaunchKernelGGL((vv_gemv_kernel<XS_, P, E, 16, WP_, 0, 1>), grid, dim3(WP_ * 64), 0, s, a.W, a.W2, a.X, a.Y, a.nw, a.T, a.N, a.K, a.ldx, a.ldy, a);   \
         return hipGetLastError() == hipSuccess ? 0 : -2; } while (0)
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
         return hipGetLastError() == hipSuccess ? 0 : -2; } while (0)
#define X(P, E)                                                                                         \
    if (a.pro == P && a.epi == E) {                                                                     \
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
        return -3;
    }
    static const bool wpb8_only = getenv("VVHIP_GEMV_WPB8") != nullptr;      // A/B switch
    if (a.kgrid > 1) grid.y = a.kgrid;
    if (xs == 1 && !wpb8_only && a.kgrid <= 1) {
        if (n_tiles > 256) {
            // one utterance = two rows (cond + uncond): the 2-row form halves the activation registers and the staging tile, so
            // one more workgroup fits per SIMD (RMS_MOD + SwiGLU: 160 -> <= 128 VGPRs) and a 672-tile launch is resident at once
            static const bool no_mr2 = getenv("VVHIP_NO_MR2") != nullptr;
            if (a.T <= 2 && !no_mr2) {
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
