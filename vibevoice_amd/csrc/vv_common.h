// vv_common.h -- shared device/host declarations for libvvhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define VV_WAVE 64

// ---- packed weight tile ------------------------------------------------------
// A [N x K] matrix is stored as tiles of 16 rows x 32 k.  One tile = 1 KiB =
// 64 lanes x 16 B, in exactly the order the lanes of a wave consume it as the
// A (or B) operand of v_mfma_f32_16x16x32_bf16:
//   lane l holds W[n0 + (l & 15)][k0 + (l >> 4) * 8 + 0..7]
// tiles are ordered [n_tile][k_tile]; rows >= N and columns >= K are zero.
static inline __host__ __device__ int64_t vv_packed_elems(int N, int K) {
    return (int64_t)((N + 15) / 16) * ((K + 31) / 32) * 512;
}

// ---- row table for LM steps ---------------------------------------------------
struct VVRow {
    int cache;   // KV cache id
    int pos;     // position of this token == cache length before the append
};

// ---- workspace of the prefill GEMM's K-split partial round (prefill.hip: vv_gemm4_kernel) ----
struct VVGemmWs {
    float* partials;     // 256 slots x 256 KiB (a workgroup's 32 accumulators x 512 threads x f32x4)
    unsigned* flags;     // 256 arrival words, zero whenever no launch is in flight
    unsigned* err;       // host-mapped word, set by a wait that timed out
    // short prompts (vv_gemm3_kernel: a few dozen 128 x 128 tiles): K split over grid.y into dense fp32 partial tensors
    // [part][T][N], summed in part order (+ bias / residual) by vv_g3_reduce_kernel -- no hand-off inside a launch
    float* g3_partials;
    size_t g3_bytes;
};

// ---- generic skinny GEMM -----------------------------------------------------
// Y[t][n] (op)= sum_k f(X[t][k]) * W[n][k]
enum { VV_PRO_NONE = 0, VV_PRO_RMS = 1, VV_PRO_RMS_MOD = 2, VV_PRO_ADD_SILU = 3,
       VV_PRO_NORMDW = 4 };   // gemv.hip only, one row: Block1D's norm + causal depthwise conv + layer scale + residual, then RMSNorm (see VVGemm::dw_*)
enum { VV_EPI_STORE = 0, VV_EPI_BIAS = 1, VV_EPI_BIAS_GELU = 2, VV_EPI_SWIGLU = 3,
       VV_EPI_RESID = 4, VV_EPI_GATED_RESID = 5, VV_EPI_CFG_DPM = 6,
       VV_EPI_QKV_ROPE = 7 };   // prefill.hip only: bias + RoPE + KV-cache append in the QKV GEMM's epilogue

struct VVGemm {
    const u32x4* W;        // packed tiles
    const u32x4* W2;       // second matrix ("up") for VV_EPI_SWIGLU, else null
    const float* X;        // [T] rows, K contiguous floats each, row stride ldx
    float* Y;              // [T][N], row stride ldy
    const float* nw;       // PRO_RMS*: norm weight [K] (null = no affine)
    const float* mod_scale;// PRO_RMS_MOD: per-row [T][ld_mod]
    const float* mod_shift;
    const float* addvec;   // PRO_ADD_SILU: [K]
    const float* bias;     // [N] or null
    const float* nscale;   // EPI_RESID: per-n scale (gamma) or null
    const float* gate;     // EPI_GATED_RESID: per-row [T][ld_gate]
    int T, N, K;
    int ldx, ldy, ld_mod, ld_gate;
    int pro, epi;
    int ksplit;            // 1, 2 or 4 waves of the block split K
    int nt;                // non-temporal weight loads (streamed-once weights)
    int t_pad;             // LDS row stride of the staging tile (set by the launcher)
    // EPI_CFG_DPM (diffusion-head final layer): rows [0,n) cond, [n,2n) uncond -> CFG + DPM-Solver++ update in place
    float* z;              // [2n][N] noisy latent (both halves rewritten)
    float* x0p;            // [n][N] previous x0 prediction
    const float* coef;     // {a, s, cs, c0, c1, cn}: one 6-float row of the schedule table per solver step
    float cfg;
    int n_cfg;
    const float* sde_noise; // stochastic solver (sde-dpmsolver++): this step's variance noise [n][N], added as cn * eps; null = off
    unsigned long long* dbg;   // optional phase timestamps (VV_GEMM_TIMING builds only)
    // row t reads activation row (t % x_row_mod) and, for PRO_ADD_SILU, the add-vector (t / add_rows_per_vec)
    // (0 = off).  Used to batch the diffusion head's adaLN GEMM over all solver steps of a frame.
    int x_row_mod, add_rows_per_vec;
    float eps;
    // K split across workgroups (decode GEMV only, PRO_NONE + residual epilogues): workgroup column ks > 0 stores its
    // scaled partial to yparts + (ks-1)*part_stride instead of Y; consumers of that tensor add the parts back in a
    // fixed order (deterministic): xa/n_xa on the activation side, ya/n_ya on the residual side, same row strides.
    int kgrid;             // 0/1 = off
    float* yparts;
    const float* xa;       // n_xa part tensors laid out like X, part_stride floats apart
    const float* ya;       // n_ya part tensors laid out like Y
    int n_xa, n_ya;
    int part_stride;
    // Slot-batched rows (tokenizer stages of several utterances in ONE weight pass, 16-row GEMV form only): the launch has
    // sl_n * sl_T logical rows; row rg belongs to slot j = rg / sl_T, local row tt = rg % sl_T.  A side with a non-zero slot
    // stride (sl_x / sl_y, floats) lives in per-utterance streaming buffers: row pointer = base + sl_id[j] * stride + tt * ld;
    // a side with stride 0 is a dense [rows][ld] scratch tensor.  sl_n = 0: off.
    int sl_n, sl_T, sl_x, sl_y;
    int sl_id[8];
    // PRO_NORMDW (T = 1 tokenizer stages, modular_vibevoice_tokenizer.py:620-684 Block1D up to FFN1): X is the block's INPUT row
    // x [K]; the prologue forms  h = RMSNorm(x) * dw_nw,  xo = x + dw_gamma * (dw_b + sum_{j<6} dw_w[j] * dw_hist[j] + dw_w[6] * h)
    // (the causal k = 7 depthwise conv over the six cached normed rows and the new one), then feeds RMSNorm(xo) * nw to the
    // product like PRO_RMS.  The workgroup of tile 0 also writes xo -> dw_xout (FFN2's residual, the next block's input) and
    // h -> dw_hnew (the conv history's new row).  Replaces the separate vv_normdw_sliced launch in front of FFN1.
    const float* dw_hist;  // [6][K] normed history rows
    const float* dw_w;     // [7][K] taps
    const float* dw_b;     // [K]
    const float* dw_gamma; // [K]
    const float* dw_nw;    // [K] weight of the block's first norm
    float* dw_xout;        // [K]
    float* dw_hnew;        // [K]
};

// ---- batch decode over pre-packed activations (gemv16p.hip) ----
struct VVGemv16p {
    const u32x4* W;        // packed [N][K]
    const u32x4* W2;       // SwiGLU "up" matrix, same shape
    const u32x4* Xp;       // packed activations, ONE 16-row tile: [K/32][64][8]
    float* Y;              // fp32 [T][ldy]  (bias / residual / gated residual)
    unsigned char* Yp;     // packed bf16 [16][N]  (SwiGLU; PK: the new residual rows x pk_nw (x (1 + pk_sc)))
    const float* bias;     // [N] or null
    const float* gate;     // gated residual: per-row [T][ld_gate]
    int T, N, K, ldy, ld_gate;
    // RS: the operand is UN-normalised (x * norm weight): the accumulator rows are scaled by rsqrt(sum_tiles ssq_in[tile][row] / K + eps)
    const float* ssq_in;   // [ssq_tiles][16]
    int ssq_tiles;
    float eps;
    const u32x4* Xs;       // SH: second packed operand (adaLN shift rows), added unscaled: y = rs * W.Xp + W.Xs
    // PK (residual epilogues): besides Y, write bf16(y_new * pk_nw[n] * (1 + pk_sc[row][n])) packed to Yp and sum_n y_new^2 to ssq_out
    const float* pk_nw;    // [N] or null (= 1)
    const float* pk_sc;    // per-row [T][ld_pk] or null
    int ld_pk;
    float* ssq_out;        // [N / 16][16]
    // VV_EPI_CFG_DPM (the sampler's final layer): rows [0, n) cond, [n, 2n) uncond -> CFG + DPM-Solver++ update of z in place (gemv.hip)
    float* z; float* x0p; const float* coef; float cfg; int n_cfg; const float* sde_noise;
};

// ---- the seam between two solver steps of the diffusion head (headtail.hip): final layer + CFG + DPM-Solver++ update + in-projection ----
struct VVTail {
    const u32x4* Wout;     // packed [L][H]   final layer (adaLN-modulated norm without affine, then linear)
    const u32x4* Win;      // packed [H][L]   noisy_images_proj of the NEXT step
    const float* bin;      // [H] or null
    const float* X;        // residual rows [T][H] after the last head layer (+ n_xa part tensors: xa, part_stride)
    const float* xa; int n_xa, part_stride;
    const float* sc; const float* sh; int ld_mod;      // the final layer's scale / shift rows [T][ld_mod]
    float* Xout;           // the next step's residual rows [T][H] (a DIFFERENT buffer than X)
    const float* z_in; const float* x0p_in;            // [2n][L], [n][L]: this step's state
    float* z_out; float* x0p_out;                      // the next step's state (different buffers)
    const float* coef; float cfg; int n_cfg; const float* sde_noise;
    int T, H, L; float eps;
};

// up to 8 utterance slots of one launch (per-utterance kernels take the slot from blockIdx.y / .z)
struct VVSlotIds { int n; int id[8]; };
__device__ __forceinline__ int vv_slot_id(const int (&id)[8], int j) {     // select chain: no dynamic indexing of a by-value kernel argument
    int r = id[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = (j == i) ? id[i] : r;
    return r;
}

// Launch status of the calling host thread: hipGetLastError() after a launch (it clears the thread's error state, so the code is
// kept here for the message the API layer prints).  rc_ok is what the launcher returns on success.
inline thread_local int g_vv_launch_err = 0;
static inline int vv_launch_rc(int rc_ok) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return rc_ok;
    g_vv_launch_err = (int)e;
    return -2;
}

