// engine.hip -- host side of libvvhip.so: context, parameter registry/repacking,
// op orchestration (LM step, diffusion sampler, streaming codecs, connectors),
// hipGraph capture/replay.  Device work lives in gemm.hip / attn.hip / misc.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include <mutex>
#include <shared_mutex>
#include "../../include/vvhip.h"
#include "vv_common.h"

extern "C" {
int vv_gemm_launch(VVGemm a, int xs, hipStream_t s);
int vv_pack_launch(const void* src, int src_is_bf16, void* dst, int N, int K, int kind, int Cin, int Cout, int ksz,
                   int stride, hipStream_t s);
int vv_rope_append_launch(int D, const float* qkv, const VVRow* rows, const float* inv_freq, float* q_out, void* kc,
                          void* vc, int R, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, hipStream_t s);
int vv_rope_table_launch(const float* inv_freq, void* tab, int n_pos, int half, hipStream_t s);
int vv_attn_fused_launch(int D, int xs, const float* qkv, const VVRow* rows, const void* rope_tab, void* kc, void* vc,
                         int R, int Hq, int Hkv, int64_t cache_stride, int64_t head_stride, int S, int waves,
                         float* pm, float* pl, float* po, float* out, void* out_packed, hipStream_t s);
int vv_attn_launch(int D, int xs, const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq,
                   int Hkv, int64_t cache_stride, int64_t head_stride, int S, float* pm, float* pl, float* po,
                   float* out, hipStream_t s);
int vv_embed_launch(const void* table, const int* ids, float* out, int n, int H, hipStream_t s);
int vv_logits_full_launch(const void* table, const float* hidden, float* out, int n, int V, int H, hipStream_t s);
int vv_rmsnorm_rows_launch(const float* x, int ldx, float* y, int ldy, const float* w, int T, int C, float eps, hipStream_t s);
int vv_dwconv_res_launch(const float* nb, const float* x, float* xo, const float* w, const float* b, const float* gamma, int T, int C, hipStream_t s);
int vv_normdw_sliced_ok(int T, int C);
int vv_normdw_sliced_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                            const float* gamma, int T, int C, float eps, hipStream_t s);
int vv_normdw_sliced_slots_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                                  const float* gamma, int T, int C, float eps, const int* ids, int n, int64_t sx, int64_t snb,
                                  hipStream_t s);
int vv_normdw_rows_slots_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                                const float* gamma, int T, int C, float eps, const int* ids, int n, int64_t sx, int64_t snb,
                                hipStream_t s);
int vv_stem_conv_slots_launch(const float* in, const void* wp, const float* bias, float* out, int T, int N, const int* ids, int n,
                              int64_t s_in, int64_t s_out, hipStream_t s);
int vv_head_conv1_slots_launch(const float* x, const void* wp, const float* bias, float* out, int T, int Cin, const int* ids, int n,
                               int64_t s_in, int64_t s_out, hipStream_t s);
int vv_block1d_slots_launch(int C, int xs, const float* xin, float* xout, float* nst, const float* norm_w,
                            const float* ffn_norm_w, const float* gamma, const float* ffn_gamma,
                            const float* dw_w, const float* dw_b, const float* b1, const float* b2,
                            const void* w1, const void* w2, int T, float eps, const int* ids, int n, int64_t sx,
                            int64_t snst, hipStream_t s);
int vv_affine_slots_launch(const float* x, float* y, float mul, float add, int L, const int* ids, int n, int64_t stride, hipStream_t s);
int vv_normdw_launch(float* x, float* nb, const float* nw, const float* w, const float* b, const float* gamma, int T, int C,
                     float eps, hipStream_t s);
int vv_normdw_rows_ok(int T, int C);
int vv_normdw_rows_launch(const float* xin, float* xout, float* nb, const float* nw, const float* w, const float* b,
                          const float* gamma, int T, int C, float eps, hipStream_t s);
int vv_stem_conv_launch(const float* in, const void* wp, const float* bias, float* out, int T, int N, hipStream_t s);
int vv_head_conv1_launch(const float* x, const void* wp, const float* bias, float* out, int T, int Cin, hipStream_t s);
int vv_shift_rows_launch(const void* tab, int n_entries, int maxC, hipStream_t s);
int vv_zero_hist_launch(const void* tab, int n_entries, hipStream_t s);
int vv_cfg_dpm_launch(const float* eps, float* x, float* x0_prev, const float* coef, float cfg, int n, int L, const float* sde_noise, hipStream_t s);
int vv_affine_launch(const float* x, float* y, float mul, float add, int n, hipStream_t s);
int vv_copy_launch(void* dst, const void* src, size_t bytes, hipStream_t s);
int vv_zero_launch(void* dst, size_t bytes, hipStream_t s);
int vv_sampler_init_launch(const float* noise, float* z, float* x0p, int nL, hipStream_t s);
int vv_add_launch(const float* a, const float* b, float* y, int n, hipStream_t s);
int vv_tfreq_launch(const float* t, float* out, int n, hipStream_t s);
int vv_silu_launch(float* x, int n, hipStream_t s);
int vv_ada_in_launch(const float* cproj, const float* temb, float* out, int rows, int n_steps, int H, hipStream_t s);
int vv_add_rows_launch(const float* x, const float* v, float* y, int n, int C, hipStream_t s);
int vv_relu_launch(float* x, int n, hipStream_t s);
int vv_kv_import_launch(const void* k, const void* v, int src_bf16, void* kc, void* vc, int L, int Hkv, int D, int64_t head_stride, int pos0, hipStream_t s);
int vv_kv_move_launch(void* kc, void* vc, int layers, int Hkv, int D, int64_t layer_stride, int64_t head_stride, int src, int dst, hipStream_t s);
int vv_kv_zero_v_tail_launch(void* vc, const VVRow* rows, int R, int layers, int Hkv, int D, int64_t cache_stride, int64_t layer_stride,
                             int64_t head_stride, int max_ctx, hipStream_t s);
int vv_pcm16_launch(const float* x, short* out, int n, int samples, hipStream_t s);
int vv_cvt_launch(const void* src, void* dst, int64_t n, int to_bf16, hipStream_t s);
int vv_dw_transpose_launch(const float* src, float* dst, int C, hipStream_t s);
int vv_pack_rows_launch(const float* x, int ldx, const float* nw, float eps, void* xp, int T, int K, hipStream_t s);
int vv_unpack_rows_launch(const void* xp, float* x, int T, int K, hipStream_t s);
int vv_pack16_launch(const float* x, int ldx, int mode, const float* nw, float eps, const float* sc, const float* sh, int ld_mod,
                     void* xp, int T, int K, hipStream_t s);
int vv_gemv16p_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, const float* gate,
                      int T, int N, int K, int ldy, int ld_gate, int epi, hipStream_t s);
int vv_gemv16p_launch2(const VVGemv16p* a, int epi, int flags, hipStream_t s);
int vv_head_tail_ok(const VVTail* a);
int vv_head_tail_launch(const VVTail* a, int tiles_per_wg, hipStream_t s);
int vv_pack16_tiles_launch(const float* x, int ldx, int64_t stride_outer, int n_inner, int64_t stride_inner, void* xp, int64_t tile_bytes,
                           int T, int K, int n_tiles, hipStream_t s);
int vv_ada_pack_launch(const float* cproj, const float* temb, void* xp, int rows, int n_steps, int H, hipStream_t s);
int vv_gemm3_launch(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias, int T, int N, int K,
                    int ldy, int epi, const VVGemmWs* ws, hipStream_t s);
int vv_gemm_qkv_rope_launch(const void* W, const void* Xp, const float* bias, int T, int K, int D, int Hq, int Hkv, const VVRow* rows_dev,
                            const void* rope_tab, float* q_out, void* kc, void* vc, int64_t cache_stride, int64_t head_stride,
                            const VVGemmWs* ws, hipStream_t s);
int vv_attn_prefill4_launch(int D, const float* q, const VVRow* rows, const void* kc, const void* vc, int R, int Hq, int Hkv,
                            int64_t cache_stride, int64_t head_stride, float* out, void* out_packed, hipStream_t s);
int vv_block1d_supported(int C);
int vv_gemv_ok(const VVGemm* a);
int vv_tile_ok(const VVGemm* a, int xs);
int vv_block1d_launch(int C, int xs, const float* xin, float* xout, float* nst, const float* norm_w,
                      const float* ffn_norm_w, const float* gamma, const float* ffn_gamma, const float* dw_w,
                      const float* dw_b, const float* b1, const float* b2, const void* w1, const void* w2, int T,
                      float eps, hipStream_t s);
}

struct VVShiftH { float* buf; int T, hist, C; };

static thread_local char g_err[512] = "";
// Contexts that share weights are driven from several host threads (Engine.fork): a stream capture in one thread must not overlap
// device calls of this library in another (hipErrorStreamCaptureInvalidated was seen with three lanes: one capturing, one running
// first-sight eager launches).  Every API call that enqueues work holds this lock shared; a capture holds it exclusively.
static std::shared_mutex g_dev_mu;
#define VV_SHARED std::shared_lock<std::shared_mutex> _vv_dev_lk(g_dev_mu)
// parent / child bookkeeping of shared contexts (n_children, zombie): forks are created and closed from lane threads that hold
// g_dev_mu only SHARED, so the counters have their own mutex
static std::mutex g_family_mu;

namespace {

enum WKind { W_MAT = 0, W_VEC = 1, W_DW = 2, W_TABLE = 3, W_BIAS_REP = 4 };

struct Weight {
    std::string name;
    int kind = W_VEC;
    int64_t nelem = 0;          // source element count
    void* dev = nullptr;        // final storage
    // W_MAT packing parameters
    int N = 0, K = 0, pk = 0, Cin = 0, Cout = 0, ksz = 0, stride = 0;
    int rep = 1;                // W_BIAS_REP: repeat count
    bool loaded = false;
    bool optional = false;
};

struct Block {
    int C;
    float *norm_w, *ffn_norm_w, *gamma, *ffn_gamma, *dw_w, *dw_b, *b1, *b2;
    void *w1, *w2;
    float* nb;                  // unfused path: [6 + Tmax][C] normed buffer with history
    float* nst;                 // fused path: [12][C] normed history (rows 0..5) + next state (rows 6..11)
    int64_t nb_stride;          // floats between the nb (nst) buffers of consecutive utterance slots
};

struct ConvG {                  // conv / transposed conv as a GEMM over a time-major buffer
    void* w; float* bias;
    int K, N, ldx;              // per output row
    int rows_per_frame;         // output rows per frame
};

struct Stage {
    int C, Tpf;                 // channels, time steps per frame
    int hist;                   // history rows kept in front of xs
    float* xs;                  // [hist + Tmax][C]
    float* xs2;                 // fused stages ping-pong between xs and xs2
    float* xfinal;              // buffer holding the stage output (and its history rows)
    bool fused;                 // blocks run as one vv_block1d_kernel each
    bool pp;                    // unfused T <= 8 stage: channel-sliced norm+conv, blocks ping-pong between xs and xs2
    int64_t sl_stride;          // floats between the xs (xs2) buffers of consecutive utterance slots
    std::vector<Block> blocks;
    ConvG in;                   // produces this stage's rows from the previous buffer
};

struct CodecNet {
    bool decoder = false;
    int Fmax = 1, in_dim = 1, out_dim = 1, in_hist = 6, in_Tpf = 1;
    std::vector<float*> in_buf;            // per slot: [6 + Tin][in_dim]
    std::vector<std::vector<Stage>> st;    // per slot
    ConvG head;
    std::vector<float*> u;                 // FFN hidden scratch, one per slot (slots may run concurrently on different streams)
    std::vector<std::map<int, void*>> shift_tab;   // per slot: F -> device table
    std::vector<int> shift_n;
    std::vector<void*> zero_tab;
    int maxC = 1;
    // slot-batched stages (several utterances' rows in ONE weight pass, run_codec_batch): the leading `kd` stages of a decoder,
    // the stages from `ke` on of an encoder -- the T <= 8, C >= 1024 stages that hold ~95 % of a tokenizer's weight bytes
    int kd = 0, ke = 1 << 30;
    bool head_batch = false;                 // the head conv has a slot-batched form too
    int64_t in_stride = 0, u_stride = 0;     // floats between the in_buf / u buffers of consecutive slots
    std::map<uint64_t, std::pair<void*, int>> shift_multi; // slot bit mask -> merged history-shift table
};

struct GraphEntry { hipGraphExec_t exec; uint64_t last_use; };

}  // namespace

struct vv_ctx {
    vv_config c;
    char err[512];
    std::vector<Weight> w;
    std::map<std::string, int> widx;
    int H, D, Hq, Hkv, I, QKV;
    // LM params
    struct Layer { float *ln1, *ln2, *bqkv; void *wqkv, *wo, *wg, *wu, *wd; };
    std::vector<Layer> layers;
    float *lm_norm = nullptr, *inv_freq = nullptr;
    int ws_rows = 0;
    void* rope_tab = nullptr; bool rope_ready = false;
    float *tts_types = nullptr, *eos_b1 = nullptr, *eos_b2 = nullptr; void *eos_w1 = nullptr, *eos_w2 = nullptr;
    void *embed = nullptr, *lm_head = nullptr;
    bool lm_head_loaded = false;
    void* valid_w = nullptr; int n_valid = 0;
    // LM runtime
    void *kc = nullptr, *vc = nullptr;
    int64_t cache_stride = 0, head_stride = 0, layer_stride = 0;
    VVRow* rows_dev = nullptr; VVRow* rows_pin = nullptr; int rows_cap = 2048;
    int* ids_dev = nullptr; int* ids_pin = nullptr; int ids_cap = 64;      // token ids per vv_embed call: max(64, max_rows)
    // pinned staging is a ring (slot reuse waits on that slot's own copy event, long since complete): a step's
    // launches can be enqueued while the previous step is still running, no host-side stream sync
    static constexpr int RING = 32;
    hipEvent_t ring_ev[RING] = {}; bool ring_used[RING] = {}; int ring_i = 0;
    float *h = nullptr, *qkv = nullptr, *qrot = nullptr, *attn = nullptr, *act = nullptr;
    float *h_parts = nullptr, *xh_parts = nullptr;     // K-split partial tensors of the residual streams (2 x [rows][H] each)
    void *xp = nullptr, *actp = nullptr;               // prefill (prefill.hip): activations as packed bf16 MFMA fragments
    bool tile3_ok = false, attn2_ok = false;
    VVGemmWs gws = {nullptr, nullptr, nullptr, nullptr, 0};         // K-split workspace of the long-prompt GEMM (null: never split)
    bool fold_normdw = true;      // one-row tokenizer stages: norm + depthwise conv inside FFN1's prologue (VVHIP_FOLD_NORMDW=0: separate launch)
    // batch decode (5..16 rows, bf16 mode): activations packed once per op into one 16-row fragment tile (gemv16p.hip)
    void *p16_x = nullptr, *p16_act = nullptr; bool p16_ok = false;
    // round 6: the producer's residual epilogue packs the next projection's operand (x * norm weight, un-normalised) and leaves per-tile
    // partial sums of squares; the consumer applies 1/rms to its accumulator rows (gemv16p.hip RS / PK / SH).  VVHIP_P16_FUSE=0: the
    // separate vv_pack16 launches of round 3.
    void *p16_y = nullptr, *p16_shift = nullptr; float *ssq_a = nullptr, *ssq_b = nullptr; bool p16_fuse = false;
    bool p16_head_sh = false;       // the head's layers 1.. take the two-operand (x, shift) form; off: they keep their vv_pack16 launch
    size_t p16_shift_tile = 0;      // bytes of one packed [16][H] tile of the head's shift rows
    float *pm = nullptr, *pl = nullptr, *po = nullptr;
    // head
    int HF = 0, MODW = 0;
    struct HLayer { float* norm; void *wg, *wu, *wd; };
    std::vector<HLayer> hl;
    void *h_in = nullptr, *h_cond = nullptr, *h_t0 = nullptr, *h_t2 = nullptr, *h_ada = nullptr, *h_out = nullptr;
    int n_steps = 0;
    float *temb = nullptr, *coef = nullptr, *tvals = nullptr;
    bool sde_on = false;                   // the schedule table carries variance-noise scales (sde-dpmsolver++)
    float* mod_all = nullptr; size_t mod_all_bytes = 0;
    float* ada_in = nullptr;
    void* ada_p = nullptr;                 // the same rows as packed bf16 MFMA fragments (bf16 mode: one tile GEMM for all steps)
    float *cproj = nullptr, *mod = nullptr, *zz = nullptr, *x0p = nullptr, *xh = nullptr, *hact = nullptr, *eps = nullptr;
    // second generation of the sampler's state (headtail.hip: a solver step reads one generation and writes the other)
    float *zz2 = nullptr, *x0p2 = nullptr, *xh2 = nullptr;
    int head_tail_tpw = 0;          // in-projection tiles per workgroup of the fused seam; 0 = off (VVHIP_HEAD_TAIL=0 / exact modes)
    float *tmp1 = nullptr, *tmp2 = nullptr;
    // connectors
    struct Conn { void *fc1, *fc2; float *b1, *b2, *norm; } ac_conn, sem_conn;
    float *ct1 = nullptr;
    // codecs
    CodecNet dec, aenc, senc;
    int enc_pass = 0;                      // frames per voice-prompt encoder pass (0: aenc.Fmax)
    // vv_codec_chain_batch: the per-utterance parts of a batch's tokenizer chains fork onto these streams (graph branches)
    hipStream_t side[8] = {}; hipEvent_t ev_fork = nullptr, ev_join[8] = {}; bool side_ready = false;
    float scaling = 1.f, bias = 0.f;
    int hop = 3200;
    // staging
    void* stage = nullptr; size_t stage_bytes = 0;
    std::map<std::string, GraphEntry> graphs;      // bounded: least-recently-used entries are destroyed beyond graph_cap
    uint64_t graph_tick = 0; size_t graph_cap = 512;
    std::set<std::string> seen;
    std::set<void*> allocs;                // every dalloc() of this engine: released by vv_destroy
    // weight sharing (vv_create_shared): a child context's weight storage IS its parent's -- the k-th weight allocation of
    // vv_create returns the parent's k-th one (same model configuration -> same sequence); everything else (KV caches, activations,
    // tokenizer state, graphs, staging) is the child's own, so two contexts decode concurrently on two streams over one weight copy
    vv_ctx* parent = nullptr; int n_children = 0; bool zombie = false, creating = false;
    // VVHIP_NAN_PROBE=1 (debugging): scan kernels behind the sampler's launches, inside the captured graph as well; the first stage whose
    // output holds a non-finite value is printed after the call (nan_probe())
    unsigned* probe_rec = nullptr; std::vector<std::string> probe_names; int probe_on = -1; int probe_calls = 0;
    int64_t foreign_nodes = 0;        // nodes of captured graphs that are not kernel launches (memset / memcpy nodes: none must exist, see misc.hip's copy kernels)
    int64_t capture_fallbacks = 0; char last_capture_issue[256] = "";     // stream captures that fell back to an eager run (graphed())
    std::vector<std::pair<void*, size_t>> wallocs; size_t wshare_i = 0;
    int64_t launches = 0;
    // optional per-GEMM-launch hipEvent timing (vv_profile_begin/end)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    int prof_n = 0;
    double prof_bytes = 0.0;
    struct ProfRec { int T, N, K, pro, epi, dual; double bytes; int gemv; };
    std::vector<ProfRec> prof_rec;
    std::vector<VVGemm> prof_gemv;          // the decode-GEMV launches of the last profile window, in issue order (vv_profile_replay)
    double prof_gemv_bytes = 0.0;
    // launches of the other timed kernel families recorded in the same window (vv_profile_replay_family): 1 = vv_gemv16p_kernel
    // (batch decode projections), 2 = decode attention (vv_attn_fused_kernel + its vv_attn_merge2_kernel)
    struct ProfLaunch { int family; double bytes; std::function<int(hipStream_t)> fn; };
    std::vector<ProfLaunch> prof_other;
    int64_t prof_raw_ns = 0, prof_ev_over_ns = 0;
    hipStream_t prof_stream = nullptr;     // last vv_profile_end: uncalibrated GEMV total, one empty event pair
#ifdef VV_GEMM_TIMING
    // timing builds only (tools/step_timeline.py): every GEMM launch gets a stamp slice for its workgroups' entry/exit clocks
    unsigned long long* tl_base = nullptr; int tl_idx = 0;
    struct TlRec { int T, N, K, pro, epi; };
    std::vector<TlRec> tl_rec;
#endif
};

static int fail(vv_ctx* ctx, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    char* dst = ctx ? ctx->err : g_err;
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (ctx) snprintf(g_err, 512, "%s", ctx->err);
    return -1;
}
#define HIPCHK(ctx, e) do { hipError_t _e = (e); if (_e != hipSuccess) return fail(ctx, "%s:%d hip error %s", __FILE__, __LINE__, hipGetErrorString(_e)); } while (0)
#define VVCHK(e) do { int _r = (e); if (_r != 0) return _r < 0 ? fail(ctx, "%s:%d launch failed (%d): hip error %d (%s)", __FILE__, __LINE__, _r, g_vv_launch_err, hipGetErrorString((hipError_t)g_vv_launch_err)) : _r; } while (0)

static __global__ void vv_nan_probe_kernel(const float* __restrict__ p, int n, unsigned* __restrict__ rec) {
    unsigned cnt = 0, first = 0xffffffffu, mx = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float v = p[i];
        if (!(fabsf(v) <= 3.0e38f)) { cnt++; first = min(first, (unsigned)i); }
        else mx = max(mx, __float_as_uint(fabsf(v)));
    }
    if (cnt) { atomicAdd(rec, cnt); atomicMax(rec + 1, 0xffffffffu - first); }
    atomicMax(rec + 2, mx);
}
constexpr int PROBE_MAX = 1024;
static void nan_probe(vv_ctx* ctx, hipStream_t st, const char* name, const void* p, size_t n) {
    if (ctx->probe_on < 0) { const char* e = getenv("VVHIP_NAN_PROBE"); ctx->probe_on = (e && (e[0] == '1' || e[0] == '2')) ? 1 : 0; }
    if (!ctx->probe_on || !p || n == 0) return;
    if (!ctx->probe_rec) { if (hipMalloc(&ctx->probe_rec, PROBE_MAX * 16) != hipSuccess) { ctx->probe_on = 0; return; } hipMemset(ctx->probe_rec, 0, PROBE_MAX * 16); }
    const int id = (int)ctx->probe_names.size();
    if (id >= PROBE_MAX) return;
    ctx->probe_names.push_back(name);
    if (id == 0) {       // VVHIP_NAN_PROBE=2: reset the records with a MEMSET NODE (the form that showed the stale-pattern fills); 1: with a kernel
        const char* e = getenv("VVHIP_NAN_PROBE");
        if (e && e[0] == '2') (void)hipMemsetAsync(ctx->probe_rec, 0, PROBE_MAX * 16, st); else (void)vv_zero_launch(ctx->probe_rec, PROBE_MAX * 16, st);
    }
    hipLaunchKernelGGL(vv_nan_probe_kernel, dim3(64), dim3(256), 0, st, (const float*)p, (int)n, ctx->probe_rec + 4 * id);
}
static void nan_probe_report(vv_ctx* ctx, hipStream_t st, const char* what) {
    if (ctx->probe_on != 1 || !ctx->probe_rec) return;
    std::vector<unsigned> h(PROBE_MAX * 4);
    const hipError_t e1 = hipStreamSynchronize(st);
    const hipError_t e2 = hipMemcpy(h.data(), ctx->probe_rec, PROBE_MAX * 16, hipMemcpyDeviceToHost);
    const int call = ctx->probe_calls++;
    if (e1 != hipSuccess || e2 != hipSuccess) { fprintf(stderr, "[nan_probe] %s call %d: sync %d copy %d\n", what, call, (int)e1, (int)e2); (void)hipGetLastError(); return; }
    { const size_t ns = ctx->probe_names.size(); bool tail_dirty = false;      // the words past the last stage must still be the memset's zeros
      for (size_t i = 4 * ns; i < (size_t)PROBE_MAX * 4; ++i) if (h[i]) { tail_dirty = true; break; }
      if (tail_dirty) fprintf(stderr, "[nan_probe] %s call %d (prof %d): record buffer %p holds words nobody wrote: %08x %08x %08x %08x | %08x %08x %08x %08x (last 4 words)\n",
                              what, call, (int)ctx->prof_on, (void*)ctx->probe_rec, h[0], h[1], h[2], h[3], h[4092], h[4093], h[4094], h[4095]); }
    int bad = 0;
    for (size_t i = 0; i < ctx->probe_names.size(); ++i) if (h[4 * i]) bad++;
    if (!bad) { if (call < 6) fprintf(stderr, "[nan_probe] %s call %d: %zu stages clean\n", what, call, ctx->probe_names.size()); return; }
    fprintf(stderr, "[nan_probe] %s call %d: %d of %zu stages hold non-finite values\n", what, call, bad, ctx->probe_names.size());
    int shown = 0;
    for (size_t i = 0; i < ctx->probe_names.size() && shown < 12; ++i) {
        float mx; memcpy(&mx, &h[4 * i + 2], 4);
        if (h[4 * i] || (i + 1 < ctx->probe_names.size() && h[4 * (i + 1)] && !shown)) {
            fprintf(stderr, "[nan_probe]   stage %3zu %-28s non-finite %u (first at %u), finite absmax %.4e\n", i, ctx->probe_names[i].c_str(), h[4 * i],
                    h[4 * i] ? 0xffffffffu - h[4 * i + 1] : 0u, mx);
            if (h[4 * i]) shown++;
        }
    }
}

static int ring_acquire(vv_ctx* ctx) {
    const int slot = ctx->ring_i;
    ctx->ring_i = (ctx->ring_i + 1) % vv_ctx::RING;
    if (ctx->ring_used[slot]) hipEventSynchronize(ctx->ring_ev[slot]);
    ctx->ring_used[slot] = true;
    return slot;
}

static void* dalloc(vv_ctx* ctx, size_t bytes, bool zero = true) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) { fail(ctx, "hipMalloc(%zu) failed", bytes); return nullptr; }
    if (zero) hipMemset(p, 0, bytes);
    else { static const int poison = [] { const char* e = getenv("VVHIP_POISON"); return (e && e[0] == '1') ? 1 : 0; }();      // debugging: NaN words in
           if (poison) hipMemset(p, 0xFF, bytes); }                                                                              // every buffer handed out un-zeroed
    if (ctx) ctx->allocs.insert(p);
    return p;
}
static void dfree(vv_ctx* ctx, void* p) {
    if (!p) return;
    if (ctx) ctx->allocs.erase(p);
    hipFree(p);
}

static int add_w(vv_ctx* ctx, const std::string& name, int kind, int64_t nelem, bool optional = false) {
    Weight w; w.name = name; w.kind = kind; w.nelem = nelem; w.optional = optional;
    ctx->widx[name] = (int)ctx->w.size();
    ctx->w.push_back(w);
    return (int)ctx->w.size() - 1;
}
// registers a packed matrix stored inside `base` at n-tile offset `ntile_off` of a [Ntot x K] tile array
static void add_mat(vv_ctx* ctx, const std::string& name, int N, int K, void* base, int ntile_off,
                    int pk = 0, int Cin = 0, int Cout = 0, int ksz = 0, int stride = 0, int64_t src_nelem = -1) {
    int i = add_w(ctx, name, W_MAT, src_nelem < 0 ? (int64_t)N * K : src_nelem);
    Weight& w = ctx->w[i];
    w.N = N; w.K = K; w.pk = pk; w.Cin = Cin; w.Cout = Cout; w.ksz = ksz; w.stride = stride;
    const int k_tiles = (K + 31) / 32;
    w.dev = (char*)base + (int64_t)ntile_off * k_tiles * 1024;
}
// weight storage: during vv_create the allocation sequence is recorded (parent) or replayed from the parent (shared child)
static void* walloc(vv_ctx* ctx, size_t bytes, bool zero = true) {
    if (!ctx->creating) return dalloc(ctx, bytes, zero);
    if (ctx->parent) {
        vv_ctx* p = ctx->parent;
        if (ctx->wshare_i >= p->wallocs.size() || p->wallocs[ctx->wshare_i].second != bytes) {
            fail(ctx, "vv_create_shared: weight allocation %zu (%zu bytes) does not match the parent's -- different model configuration", ctx->wshare_i, bytes);
            return nullptr;
        }
        void* q = p->wallocs[ctx->wshare_i++].first;
        ctx->wallocs.push_back({q, bytes});
        return q;
    }
    void* q = dalloc(ctx, bytes, zero);
    ctx->wallocs.push_back({q, bytes});
    return q;
}
static void* alloc_packed(vv_ctx* ctx, int N, int K) { return walloc(ctx, (size_t)vv_packed_elems(N, K) * 2); }
static float* add_vec(vv_ctx* ctx, const std::string& name, int64_t n, float* dst = nullptr, int kind = W_VEC, int rep = 1) {
    int i = add_w(ctx, name, kind, n);
    if (!dst) dst = (float*)walloc(ctx, (size_t)n * rep * 4);
    ctx->w[i].dev = dst; ctx->w[i].rep = rep;
    return dst;
}

// ------------------------------------------------------------------ codec nets
static VVGemm mk_gemm(const void* W, const float* X, float* Y, int T, int N, int K, int ldx, int ldy);
static int build_codec(vv_ctx* ctx, CodecNet& net, const std::string& pfx, bool decoder, int vae_dim, int Fmax, int n_slots) {
    const vv_config& c = ctx->c;
    const int ns = c.n_stages;
    const int nf = c.n_filters;
    net.decoder = decoder; net.Fmax = Fmax;
    std::vector<int> depths(ns), ratios(c.n_ratios);
    if (decoder) { for (int i = 0; i < ns; ++i) depths[i] = c.enc_depths[ns - 1 - i]; for (int i = 0; i < c.n_ratios; ++i) ratios[i] = c.ratios[i]; }
    else { for (int i = 0; i < ns; ++i) depths[i] = c.enc_depths[i]; for (int i = 0; i < c.n_ratios; ++i) ratios[i] = c.ratios[c.n_ratios - 1 - i]; }
    int hop = 1; for (int i = 0; i < c.n_ratios; ++i) hop *= c.ratios[i];
    ctx->hop = hop;
    // per-stage geometry
    std::vector<int> C(ns), Tpf(ns);
    for (int i = 0; i < ns; ++i) {
        if (decoder) { C[i] = nf << (ns - 1 - i); Tpf[i] = (i == 0) ? 1 : Tpf[i - 1] * ratios[i - 1]; }
        else { C[i] = nf << i; Tpf[i] = (i == 0) ? hop : Tpf[i - 1] / ratios[i - 1]; }
    }
    net.in_dim = decoder ? vae_dim : 1;
    net.out_dim = decoder ? 1 : vae_dim;
    net.in_Tpf = decoder ? 1 : hop;
    net.in_hist = 6;
    net.maxC = 1;
    size_t umax = 0;
    // ---- weights (shared across slots) ----
    struct StageW { ConvG in; std::vector<Block> blocks; };
    std::vector<StageW> sw(ns);
    for (int i = 0; i < ns; ++i) {
        ConvG& g = sw[i].in;
        char nm[256];
        if (i == 0) {
            const int Cin = net.in_dim;
            g.K = 7 * Cin; g.N = C[0]; g.ldx = Cin; g.rows_per_frame = Tpf[0];
            g.w = alloc_packed(ctx, g.N, g.K);
            snprintf(nm, 256, "%s%s.0.0.conv.conv.", pfx.c_str(), decoder ? "upsample_layers" : "downsample_layers");
            add_mat(ctx, std::string(nm) + "weight", g.N, g.K, g.w, 0, 1, Cin, g.N, 7, 1);
            g.bias = add_vec(ctx, std::string(nm) + "bias", g.N);
        } else if (decoder) {
            const int s = ratios[i - 1], Cin = C[i - 1], Cout = C[i];
            g.K = 2 * Cin; g.N = s * Cout; g.ldx = Cin; g.rows_per_frame = Tpf[i - 1];
            g.w = alloc_packed(ctx, g.N, g.K);
            snprintf(nm, 256, "%supsample_layers.%d.0.convtr.convtr.", pfx.c_str(), i);
            add_mat(ctx, std::string(nm) + "weight", g.N, g.K, g.w, 0, 2, Cin, Cout, 2 * s, s, (int64_t)Cin * Cout * 2 * s);
            g.bias = add_vec(ctx, std::string(nm) + "bias", Cout, nullptr, W_BIAS_REP, s);
        } else {
            const int s = ratios[i - 1], Cin = C[i - 1], Cout = C[i];
            g.K = 2 * s * Cin; g.N = Cout; g.ldx = s * Cin; g.rows_per_frame = Tpf[i];
            g.w = alloc_packed(ctx, g.N, g.K);
            snprintf(nm, 256, "%sdownsample_layers.%d.0.conv.conv.", pfx.c_str(), i);
            add_mat(ctx, std::string(nm) + "weight", g.N, g.K, g.w, 0, 1, Cin, Cout, 2 * s, s);
            g.bias = add_vec(ctx, std::string(nm) + "bias", Cout);
        }
        if (C[i] > net.maxC) net.maxC = C[i];
        for (int j = 0; j < depths[i]; ++j) {
            Block b; b.C = C[i]; b.nb = nullptr;
            snprintf(nm, 256, "%sstages.%d.%d.", pfx.c_str(), i, j);
            std::string p(nm);
            b.gamma = add_vec(ctx, p + "gamma", C[i]);
            b.ffn_gamma = add_vec(ctx, p + "ffn_gamma", C[i]);
            b.norm_w = add_vec(ctx, p + "norm.weight", C[i]);
            b.ffn_norm_w = add_vec(ctx, p + "ffn_norm.weight", C[i]);
            b.dw_w = add_vec(ctx, p + "mixer.conv.conv.conv.weight", (int64_t)C[i] * 7, nullptr, W_DW);
            b.dw_b = add_vec(ctx, p + "mixer.conv.conv.conv.bias", C[i]);
            b.w1 = alloc_packed(ctx, 4 * C[i], C[i]);
            add_mat(ctx, p + "ffn.linear1.weight", 4 * C[i], C[i], b.w1, 0);
            b.b1 = add_vec(ctx, p + "ffn.linear1.bias", 4 * C[i]);
            b.w2 = alloc_packed(ctx, C[i], 4 * C[i]);
            add_mat(ctx, p + "ffn.linear2.weight", C[i], 4 * C[i], b.w2, 0);
            b.b2 = add_vec(ctx, p + "ffn.linear2.bias", C[i]);
            sw[i].blocks.push_back(b);
            size_t ub = (size_t)Tpf[i] * Fmax * 4 * C[i] * 4;
            if (ub > umax) umax = ub;
        }
    }
    {   // head conv k7
        ConvG& g = net.head;
        const int Cin = C[ns - 1];
        g.K = 7 * Cin; g.N = net.out_dim; g.ldx = Cin; g.rows_per_frame = Tpf[ns - 1];
        g.w = alloc_packed(ctx, g.N, g.K);
        add_mat(ctx, pfx + "head.conv.conv.weight", g.N, g.K, g.w, 0, 1, Cin, g.N, 7, 1);
        g.bias = add_vec(ctx, pfx + "head.conv.conv.bias", g.N);
    }
    // ---- per-slot buffers + shift tables.  Every kind of buffer is ONE allocation with a uniform slot stride (a multiple of
    // 64 floats), so a slot-batched launch reaches utterance k's copy at base + k * stride ----
    auto pad64 = [](size_t n) { return (n + 63) / 64 * 64; };
    net.u.resize(n_slots);
    net.u_stride = (int64_t)pad64(umax / 4 + 1);
    float* u_all = (float*)dalloc(ctx, (size_t)n_slots * net.u_stride * 4, false);
    net.in_buf.resize(n_slots); net.st.resize(n_slots); net.shift_tab.resize(n_slots); net.zero_tab.resize(n_slots);
    net.shift_n.resize(n_slots);
    net.in_stride = (int64_t)pad64((size_t)(6 + net.in_Tpf * Fmax) * net.in_dim);
    float* in_all = (float*)dalloc(ctx, (size_t)n_slots * net.in_stride * 4);
    if (!u_all || !in_all) return -1;
    for (int sl = 0; sl < n_slots; ++sl) {
        net.u[sl] = u_all + (size_t)sl * net.u_stride;
        net.in_buf[sl] = in_all + (size_t)sl * net.in_stride;
        net.st[sl].resize(ns);
        net.zero_tab[sl] = nullptr;
    }
    for (int i = 0; i < ns; ++i) {
        const int hist = (i == ns - 1) ? 6 : (decoder ? 1 : ratios[i]);
        const bool fused = vv_block1d_supported(C[i]) && Tpf[i] >= 8 && !sw[i].blocks.empty();
        // unfused stages ping-pong between xs and xs2 when a one-launch norm + depthwise-conv kernel exists for them:
        // channel-sliced (T <= 8, C = 1024 / 2048) or row-tiled (middle stages, any T)
        const bool pp = !fused && !sw[i].blocks.empty() && (vv_normdw_sliced_ok(Tpf[i], C[i]) || vv_normdw_rows_ok(Tpf[i], C[i]));
        const int64_t xstride = (int64_t)pad64((size_t)(hist + (size_t)Tpf[i] * Fmax) * C[i]);
        float* xs_all = (float*)dalloc(ctx, (size_t)n_slots * xstride * 4);
        float* xs2_all = (fused || pp) ? (float*)dalloc(ctx, (size_t)n_slots * xstride * 4) : nullptr;
        if (!xs_all || ((fused || pp) && !xs2_all)) return -1;
        const int64_t nbstride = (int64_t)pad64(fused ? (size_t)12 * C[i] : (size_t)(6 + (size_t)Tpf[i] * Fmax) * C[i]);
        std::vector<float*> nb_all(sw[i].blocks.size());
        for (auto& p : nb_all) { p = (float*)dalloc(ctx, (size_t)n_slots * nbstride * 4); if (!p) return -1; }
        for (int sl = 0; sl < n_slots; ++sl) {
            Stage& s = net.st[sl][i];
            s.C = C[i]; s.Tpf = Tpf[i]; s.in = sw[i].in; s.hist = hist; s.fused = fused; s.pp = pp; s.sl_stride = xstride;
            s.xs = xs_all + (size_t)sl * xstride;
            s.xs2 = xs2_all ? xs2_all + (size_t)sl * xstride : nullptr;
            s.blocks = sw[i].blocks;
            s.xfinal = ((s.fused || s.pp) && (s.blocks.size() & 1)) ? s.xs2 : s.xs;
            for (size_t j = 0; j < s.blocks.size(); ++j) {
                Block& b = s.blocks[j];
                b.nb = nullptr; b.nst = nullptr; b.nb_stride = nbstride;
                if (s.fused) b.nst = nb_all[j] + (size_t)sl * nbstride;
                else b.nb = nb_all[j] + (size_t)sl * nbstride;
            }
        }
    }
    // which stages can run slot-batched (bf16 modes): the incoming conv as a slot-batched GEMV (or the stem kernel), the blocks
    // as fused block kernels, or channel-sliced / row-tiled norm+conv + slot-batched FFN GEMVs.  VVHIP_BATCH_CODEC=heavy keeps
    // only the weight-heavy T <= 8 stages batched (the rest per utterance on forked streams), =0 turns batching off.
    {
        const char* mode = getenv("VVHIP_BATCH_CODEC");
        const bool off = (mode && !strcmp(mode, "0")) || ctx->c.xsplit > 2 || n_slots < 2 || Fmax != 1;
        const bool heavy_only = mode && !strcmp(mode, "heavy");
        auto gemm_ok = [&](const ConvG& cg, int64_t sx, int64_t sy) {
            VVGemm g = mk_gemm(cg.w, net.in_buf[0], net.u[0], 2 * cg.rows_per_frame, cg.N, cg.K, cg.ldx, cg.N);
            g.epi = VV_EPI_BIAS; g.bias = cg.bias;
            g.sl_n = 2; g.sl_T = cg.rows_per_frame; g.sl_x = (int)sx; g.sl_y = (int)sy; g.sl_id[0] = 0; g.sl_id[1] = n_slots - 1;
            return vv_gemv_ok(&g) != 0;
        };
        auto ok = [&](int i) {
            const Stage& s = net.st[0][i];
            if (off || s.blocks.empty() || (s.C & 31)) return false;
            const bool stem = (i == 0 && s.in.K == 7 && s.in.ldx == 1);
            if (!stem && !gemm_ok(s.in, i == 0 ? net.in_stride : net.st[0][i - 1].sl_stride, s.sl_stride)) return false;
            if (s.pp && vv_normdw_sliced_ok(s.Tpf, s.C)) return true;
            if (heavy_only) return false;
            return s.fused || (s.pp && vv_normdw_rows_ok(s.Tpf, s.C));
        };
        net.kd = 0; net.ke = ns;
        if (decoder) { while (net.kd < ns && ok(net.kd)) net.kd++; }
        else { while (net.ke > 0 && ok(net.ke - 1)) net.ke--; }
        const ConvG& h = net.head;
        const bool conv1 = h.N == 1 && h.K == 7 * h.ldx && (h.ldx & 3) == 0 && h.ldx <= 1024;
        net.head_batch = !off && !heavy_only && (conv1 || gemm_ok(h, net.st[0][ns - 1].sl_stride, 0));
        if (!decoder && net.ke < ns && !(conv1 || gemm_ok(h, net.st[0][ns - 1].sl_stride, 0))) net.ke = ns;   // encoder tail needs its head batched
    }
    return 0;
}

static void codec_table_entries(CodecNet& net, int sl, int F, std::vector<VVShiftH>& t) {
    t.push_back({net.in_buf[sl], net.in_Tpf * F, 6, net.in_dim});
    for (auto& s : net.st[sl]) {
        t.push_back({s.xfinal, s.Tpf * F, s.hist, s.C});
        for (auto& b : s.blocks) {
            if (s.fused) t.push_back({b.nst, 6, 6, s.C});
            else t.push_back({b.nb, s.Tpf * F, 6, s.C});
        }
    }
}
// one history-shift table for a set of slots (one launch after a slot-batched pass), F = 1
static int codec_tables_multi(vv_ctx* ctx, CodecNet& net, const int* ids, int n, void** tab_out, int* n_out) {
    uint64_t mask = 0;
    for (int j = 0; j < n; ++j) mask |= 1ull << ids[j];
    auto it = net.shift_multi.find(mask);
    if (it != net.shift_multi.end()) { *tab_out = it->second.first; *n_out = it->second.second; return 0; }
    std::vector<VVShiftH> t;
    for (int j = 0; j < n; ++j) codec_table_entries(net, ids[j], 1, t);
    void* d = dalloc(ctx, t.size() * sizeof(VVShiftH), false);
    if (!d) return -1;
    HIPCHK(ctx, hipMemcpy(d, t.data(), t.size() * sizeof(VVShiftH), hipMemcpyHostToDevice));
    net.shift_multi[mask] = {d, (int)t.size()};
    *tab_out = d; *n_out = (int)t.size();
    return 0;
}
static int codec_tables(vv_ctx* ctx, CodecNet& net, int sl, int F, void** tab_out, int* n_out) {
    auto it = net.shift_tab[sl].find(F);
    if (it != net.shift_tab[sl].end()) { *tab_out = it->second; *n_out = net.shift_n[sl]; return 0; }
    std::vector<VVShiftH> t;
    codec_table_entries(net, sl, F, t);
    void* d = dalloc(ctx, t.size() * sizeof(VVShiftH), false);
    if (!d) return -1;
    HIPCHK(ctx, hipMemcpy(d, t.data(), t.size() * sizeof(VVShiftH), hipMemcpyHostToDevice));
    net.shift_tab[sl][F] = d; net.shift_n[sl] = (int)t.size();
    *tab_out = d; *n_out = (int)t.size();
    return 0;
}

static VVGemm mk_gemm(const void* W, const float* X, float* Y, int T, int N, int K, int ldx, int ldy) {
    VVGemm g; memset(&g, 0, sizeof(g));
    g.W = (const u32x4*)W; g.X = X; g.Y = Y; g.T = T; g.N = N; g.K = K; g.ldx = ldx; g.ldy = ldy;
    g.pro = VV_PRO_NONE; g.epi = VV_EPI_STORE; g.ksplit = 0; g.nt = 0; g.eps = 1e-6f;
    return g;
}
// Few output tiles x long K (the down projections of small models): split K over 2-3 workgroup columns so every CU
// streams; the partial tensors are added back by the consumers (VVGemm::xa / ya).  Returns the number of EXTRA parts.
static int ksplit_parts(const vv_ctx* ctx, VVGemm& g, float* parts, int part_stride) {
    const int n_tiles = (g.N + 15) / 16, k_tiles = (g.K + 31) / 32;
    // few tiles x long K only: at 7B widths (224 tiles for 256 CUs) three K columns put 672 workgroups on the chip, i.e. the SAME 87.5 %
    // balance (2.625 per CU against 3) as 224 workgroups on 256 CUs, and the consumer reads two more part tensors -- measured, not shipped
    // (profiles/r06_down_ksplit_7b_ab.json; VVHIP_KSPLIT_MAX_TILES raises the limit for that A/B)
    static int max_tiles = -1;
    if (max_tiles < 0) { const char* e = getenv("VVHIP_KSPLIT_MAX_TILES"); max_tiles = e ? atoi(e) : 128; }
    if (g.T > 4 || n_tiles > max_tiles || k_tiles < 96) return 0;
    const int ks = 3;
    g.kgrid = ks; g.yparts = parts; g.part_stride = part_stride;
    if (!vv_gemv_ok(&g)) { g.kgrid = 0; g.yparts = nullptr; g.part_stride = 0; return 0; }
    return ks - 1;
}
static double gemm_bytes(const VVGemm& g) {
    // algorithmic bytes of one launch: packed weights once (+ second matrix), activations in, result out (RMW epilogues twice)
    double w = (double)vv_packed_elems(g.N, g.K) * 2.0 * (g.W2 ? 2.0 : 1.0);
    double x = (double)g.T * g.K * 4.0;
    double y = (double)g.T * g.N * 4.0 * ((g.epi == VV_EPI_RESID || g.epi == VV_EPI_GATED_RESID) ? 2.0 : 1.0);
    return w + x + y;
}
static int gemm_prof(vv_ctx* ctx, const VVGemm& g, hipStream_t st) {
    if ((size_t)(2 * ctx->prof_n + 2) > ctx->prof_ev.size()) {
        size_t old = ctx->prof_ev.size();
        ctx->prof_ev.resize(old + 2048);
        for (size_t i = old; i < ctx->prof_ev.size(); ++i) hipEventCreate(&ctx->prof_ev[i]);
    }
    ctx->prof_stream = st;
    hipEventRecord(ctx->prof_ev[2 * ctx->prof_n], st);
    int r = vv_gemm_launch(g, ctx->c.xsplit, st);
    hipEventRecord(ctx->prof_ev[2 * ctx->prof_n + 1], st);
    ctx->prof_n++;
    ctx->prof_bytes += gemm_bytes(g);
    // which kernel vv_gemm_launch picks (gemm.hip): MFMA tile GEMM, decode GEMV, or the general kernel
    const bool is_gemv = !vv_tile_ok(&g, ctx->c.xsplit) && g.ksplit <= 0 && vv_gemv_ok(&g) && (g.T <= 4 || ctx->c.xsplit <= 2);
    ctx->prof_rec.push_back({g.T, g.N, g.K, g.pro, g.epi, g.W2 ? 1 : 0, gemm_bytes(g), is_gemv});
    if (is_gemv) { ctx->prof_gemv.push_back(g); ctx->prof_gemv_bytes += gemm_bytes(g); }
    return r;
}
#ifdef VV_GEMM_TIMING
constexpr int TL_MAX = 4096, TL_STRIDE = 16 + 2 * 3200;
static int gemm_tl(vv_ctx* ctx, VVGemm g, hipStream_t st) {
    if (!ctx->tl_base && getenv("VVHIP_TIMELINE")) {
        if (hipMalloc(&ctx->tl_base, (size_t)TL_MAX * TL_STRIDE * 8) != hipSuccess) return -9;
        hipMemset(ctx->tl_base, 0, (size_t)TL_MAX * TL_STRIDE * 8);
    }
    const bool gv = vv_gemv_ok(&g) && (g.T <= 4 || ctx->c.xsplit <= 2);
    const bool want = gv ? ((g.T <= 4 || g.T != 16 || g.N > 16384) && (g.N + 15) / 16 <= 3200) : (g.T > 16);
    if (ctx->tl_base && ctx->tl_idx < TL_MAX && want) {
        g.dbg = ctx->tl_base + (size_t)ctx->tl_idx * TL_STRIDE;
        ctx->tl_rec.push_back({g.T, g.N, g.K, g.pro, gv ? g.epi : g.epi + 100});
        ctx->tl_idx++;
    }
    return vv_gemm_launch(g, ctx->c.xsplit, st);
}
extern "C" int vv_timeline_dump(vv_ctx* ctx, unsigned long long* out_host, int* meta_host, int max_launches) {
    VV_SHARED;                   // a device-wide synchronize: never while another context's capture is open
    hipDeviceSynchronize();
    const int n = std::min(max_launches, ctx->tl_idx);
    if (n > 0) hipMemcpy(out_host, ctx->tl_base, (size_t)n * TL_STRIDE * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) { const auto& r = ctx->tl_rec[i]; int* m = meta_host + 5 * i; m[0] = r.T; m[1] = r.N; m[2] = r.K; m[3] = r.pro; m[4] = r.epi; }
    return n;
}
#define GEMM(g) do { ctx->launches++; VVCHK(gemm_tl(ctx, g, st)); } while (0)
#else
#define GEMM(g) do { ctx->launches++; if (ctx->prof_on) VVCHK(gemm_prof(ctx, g, st)); else VVCHK(vv_gemm_launch(g, ctx->c.xsplit, st)); } while (0)
#endif

// Runs one codec net over F frames for slot `sl`.  The caller has already written the
// input rows into net.in_buf[sl] + 6*in_dim.
// Stages [i0, i1) only (i1 < 0: to the end); `head` / `shift`: run the head conv / the history shift at the end.  The split
// forms serve vv_codec_chain_batch, where part of the net runs slot-batched (run_codec_batch) and the rest per utterance.
// tail_valid >= 0 (encoder, last pass of a ragged input): only the first tail_valid rows of stage 0's output are real signal.
// The reference's non-streaming encoder right-pads with zeros PER strided conv (SConv1d: get_extra_padding_for_conv1d), i.e. the
// rows past the end of the signal are ZERO at the input of every strided conv -- not conv(0) + bias, which is what the rows past the
// end hold here when the waveform is zero-padded to whole frames.  Every layer is causal, so zeroing those rows of stage i-1's
// output right before stage i's incoming conv reproduces the reference exactly; only the last, partial frame's latent changes.
static int run_codec(vv_ctx* ctx, CodecNet& net, int sl, int F, float* out, hipStream_t st, int i0 = 0, int i1 = -1,
                     bool head = true, bool shift = true, int tail_valid = -1) {
    const float eps = ctx->c.codec_eps;
    const bool stream_w = (F == 1);      // T=1 stages stream their weights exactly once
    auto& stages = net.st[sl];
    const int ns = (int)stages.size();
    if (i1 < 0) i1 = ns;
    for (int i = i0; i < i1; ++i) {
        Stage& s = stages[i];
        const int T = s.Tpf * F;
        float* x = s.xs + (size_t)s.hist * s.C;
        if (tail_valid >= 0 && i > 0) {
            Stage& pv = stages[i - 1];
            const int Tp = pv.Tpf * F;
            if (tail_valid < Tp)
                VVCHK(vv_zero_launch(pv.xfinal + ((size_t)pv.hist + tail_valid) * pv.C, (size_t)(Tp - tail_valid) * pv.C * 4, st));
            const int r = pv.Tpf / s.Tpf;                       // this stage's incoming stride
            tail_valid = (tail_valid + r - 1) / r;
        }
        {   // incoming conv
            const ConvG& cg = s.in;
            const float* X = (i == 0) ? net.in_buf[sl] : stages[i - 1].xfinal;
            const int Trows = cg.rows_per_frame * F;
            if (i == 0 && cg.K == 7 && cg.ldx == 1) {
                ctx->launches++;                 // encoder stem: mono input, k = 7 (not an MFMA shape)
                VVCHK(vv_stem_conv_launch(X, cg.w, cg.bias, x, Trows, cg.N, st));
            } else {
                VVGemm g = mk_gemm(cg.w, X, x, Trows, cg.N, cg.K, cg.ldx, cg.N);
                g.epi = VV_EPI_BIAS; g.bias = cg.bias; g.nt = stream_w && Trows <= 16;
                GEMM(g);
            }
        }
        if (s.fused) {
            float* cur = s.xs;
            float* oth = s.xs2;
            for (auto& b : s.blocks) {
                ctx->launches++;
                VVCHK(vv_block1d_launch(s.C, ctx->c.xsplit, cur + (size_t)s.hist * s.C, oth + (size_t)s.hist * s.C, b.nst,
                                        b.norm_w, b.ffn_norm_w, b.gamma, b.ffn_gamma, b.dw_w, b.dw_b, b.b1, b.b2, b.w1, b.w2,
                                        T, eps, st));
                std::swap(cur, oth);
            }
            continue;
        }
        float* xo = s.pp ? s.xs2 + (size_t)s.hist * s.C : x;       // pp stages: each block's norm+conv writes the other buffer
        for (auto& b : s.blocks) {
            if (s.pp && T == 1 && ctx->fold_normdw && vv_normdw_sliced_ok(T, s.C)) {
                // one-row stages (C = 2048: 8 blocks per net): the block's norm + depthwise conv + layer scale + residual run in
                // FFN1's prologue (VV_PRO_NORMDW) -- one launch less per block on a chain where every launch is a latency link
                VVGemm g1 = mk_gemm(b.w1, x, net.u[sl], T, 4 * s.C, s.C, s.C, 4 * s.C);
                g1.pro = VV_PRO_NORMDW; g1.nw = b.ffn_norm_w; g1.eps = eps; g1.epi = VV_EPI_BIAS_GELU; g1.bias = b.b1; g1.nt = stream_w;
                g1.dw_hist = b.nb; g1.dw_w = b.dw_w; g1.dw_b = b.dw_b; g1.dw_gamma = b.gamma; g1.dw_nw = b.norm_w;
                g1.dw_xout = xo; g1.dw_hnew = b.nb + 6 * (size_t)s.C;
                if (vv_gemv_ok(&g1)) {
                    GEMM(g1);
                    VVGemm g2 = mk_gemm(b.w2, net.u[sl], xo, T, s.C, 4 * s.C, 4 * s.C, s.C);
                    g2.epi = VV_EPI_RESID; g2.bias = b.b2; g2.nscale = b.ffn_gamma; g2.nt = stream_w;
                    GEMM(g2);
                    std::swap(x, xo);
                    continue;
                }
            }
            if (s.pp && vv_normdw_sliced_ok(T, s.C)) {
                ctx->launches += 1;
                VVCHK(vv_normdw_sliced_launch(x, xo, b.nb, b.norm_w, b.dw_w, b.dw_b, b.gamma, T, s.C, eps, st));
            } else if (s.pp && vv_normdw_rows_ok(T, s.C)) {
                ctx->launches += 1;
                VVCHK(vv_normdw_rows_launch(x, xo, b.nb, b.norm_w, b.dw_w, b.dw_b, b.gamma, T, s.C, eps, st));
            } else if (!s.pp && (size_t)T * s.C <= 8192 && (s.C & 3) == 0) {     // one workgroup is only faster for tiny row sets
                ctx->launches += 1;
                VVCHK(vv_normdw_launch(x, b.nb, b.norm_w, b.dw_w, b.dw_b, b.gamma, T, s.C, eps, st));
            } else {
                ctx->launches += 2;
                VVCHK(vv_rmsnorm_rows_launch(x, s.C, b.nb + 6 * (size_t)s.C, s.C, b.norm_w, T, s.C, eps, st));
                VVCHK(vv_dwconv_res_launch(b.nb, x, xo, b.dw_w, b.dw_b, b.gamma, T, s.C, st));
            }
            VVGemm g1 = mk_gemm(b.w1, xo, net.u[sl], T, 4 * s.C, s.C, s.C, 4 * s.C);
            g1.pro = VV_PRO_RMS; g1.nw = b.ffn_norm_w; g1.eps = eps; g1.epi = VV_EPI_BIAS_GELU; g1.bias = b.b1;
            g1.nt = stream_w && T <= 16;
            GEMM(g1);
            VVGemm g2 = mk_gemm(b.w2, net.u[sl], xo, T, s.C, 4 * s.C, 4 * s.C, s.C);
            g2.epi = VV_EPI_RESID; g2.bias = b.b2; g2.nscale = b.ffn_gamma; g2.nt = stream_w && T <= 16;
            GEMM(g2);
            if (s.pp) std::swap(x, xo);
        }
    }
    if (head) {   // head conv
        const ConvG& cg = net.head;
        Stage& s = stages[ns - 1];
        if (cg.N == 1 && cg.K == 7 * cg.ldx && (cg.ldx & 3) == 0 && cg.ldx <= 1024) {
            ctx->launches++;                     // decoder head: k = 7 conv to one channel
            VVCHK(vv_head_conv1_launch(s.xfinal, cg.w, cg.bias, out, cg.rows_per_frame * F, cg.ldx, st));
        } else {
            VVGemm g = mk_gemm(cg.w, s.xfinal, out, cg.rows_per_frame * F, cg.N, cg.K, cg.ldx, cg.N);
            g.epi = VV_EPI_BIAS; g.bias = cg.bias;
            GEMM(g);
        }
    }
    if (!shift) return 0;
    void* tab; int nt;
    if (codec_tables(ctx, net, sl, F, &tab, &nt)) return -1;
    ctx->launches++;
    VVCHK(vv_shift_rows_launch(tab, nt, net.maxC, st));
    return 0;
}

// Stages [i0, i1) of `n` utterance slots (ids ascending, one frame each) in ONE pass over the weights: every GEMM carries the
// rows of all n slots (VVGemm::sl_*: gathered from / scattered to the per-slot streaming buffers, which sit at uniform
// strides), the norm + depthwise-conv kernel takes the slot from blockIdx.y.  Only stages net.kd / net.ke admit (channel-
// sliced norm+conv stages, T <= 8 rows per frame).  `out`: dense [n][out_dim] rows of the head conv (head = true).
static int run_codec_batch(vv_ctx* ctx, CodecNet& net, const int* ids, int n, int i0, int i1, float* out, bool head, hipStream_t st) {
    const float eps = ctx->c.codec_eps;
    auto& st0 = net.st[0];                       // slot 0's descriptors: base pointers of every buffer kind
    const int ns = (int)st0.size();
    auto slots = [&](VVGemm& g, int T, int64_t sx, int64_t sy) {
        g.sl_n = n; g.sl_T = T; g.sl_x = (int)sx; g.sl_y = (int)sy; g.T = n * T;
        for (int j = 0; j < 8; ++j) g.sl_id[j] = j < n ? ids[j] : 0;
    };
    for (int i = i0; i < i1; ++i) {
        Stage& s = st0[i];
        const int T = s.Tpf;
        float* x = s.xs + (size_t)s.hist * s.C;
        {   // incoming conv: per-slot window rows in, per-slot stage rows out
            const ConvG& cg = s.in;
            const float* X = (i == 0) ? net.in_buf[0] : st0[i - 1].xfinal;
            const int64_t sx = (i == 0) ? net.in_stride : st0[i - 1].sl_stride;
            if (i == 0 && cg.K == 7 && cg.ldx == 1) {
                ctx->launches++;
                VVCHK(vv_stem_conv_slots_launch(X, cg.w, cg.bias, x, cg.rows_per_frame, cg.N, ids, n, sx, s.sl_stride, st));
            } else {
                VVGemm g = mk_gemm(cg.w, X, x, cg.rows_per_frame, cg.N, cg.K, cg.ldx, cg.N);
                g.epi = VV_EPI_BIAS; g.bias = cg.bias; g.nt = 1;
                slots(g, cg.rows_per_frame, sx, s.sl_stride);
                GEMM(g);
            }
        }
        float* xo = s.xs2 + (size_t)s.hist * s.C;
        if (s.fused) {
            for (auto& b : s.blocks) {
                ctx->launches++;
                VVCHK(vv_block1d_slots_launch(s.C, ctx->c.xsplit, x, xo, b.nst, b.norm_w, b.ffn_norm_w, b.gamma, b.ffn_gamma, b.dw_w, b.dw_b,
                                              b.b1, b.b2, b.w1, b.w2, T, eps, ids, n, s.sl_stride, b.nb_stride, st));
                std::swap(x, xo);
            }
            continue;
        }
        for (auto& b : s.blocks) {
            ctx->launches += 1;
            if (vv_normdw_sliced_ok(T, s.C))
                VVCHK(vv_normdw_sliced_slots_launch(x, xo, b.nb, b.norm_w, b.dw_w, b.dw_b, b.gamma, T, s.C, eps, ids, n, s.sl_stride, b.nb_stride, st));
            else
                VVCHK(vv_normdw_rows_slots_launch(x, xo, b.nb, b.norm_w, b.dw_w, b.dw_b, b.gamma, T, s.C, eps, ids, n, s.sl_stride, b.nb_stride, st));
            VVGemm g1 = mk_gemm(b.w1, xo, net.u[0], T, 4 * s.C, s.C, s.C, 4 * s.C);          // u: dense [n * T][4C] scratch
            g1.pro = VV_PRO_RMS; g1.nw = b.ffn_norm_w; g1.eps = eps; g1.epi = VV_EPI_BIAS_GELU; g1.bias = b.b1; g1.nt = 1;
            slots(g1, T, s.sl_stride, 0);
            GEMM(g1);
            VVGemm g2 = mk_gemm(b.w2, net.u[0], xo, T, s.C, 4 * s.C, 4 * s.C, s.C);
            g2.epi = VV_EPI_RESID; g2.bias = b.b2; g2.nscale = b.ffn_gamma; g2.nt = 1;
            slots(g2, T, 0, s.sl_stride);
            GEMM(g2);
            std::swap(x, xo);
        }
    }
    if (head) {
        const ConvG& cg = net.head;
        Stage& s = st0[ns - 1];
        if (cg.N == 1 && cg.K == 7 * cg.ldx && (cg.ldx & 3) == 0 && cg.ldx <= 1024) {
            ctx->launches++;                     // decoder head: k = 7 conv to one channel; dense [n][rows] output
            VVCHK(vv_head_conv1_slots_launch(s.xfinal, cg.w, cg.bias, out, cg.rows_per_frame, cg.ldx, ids, n, s.sl_stride, cg.rows_per_frame, st));
        } else {
            VVGemm g = mk_gemm(cg.w, s.xfinal, out, cg.rows_per_frame, cg.N, cg.K, cg.ldx, cg.N);
            g.epi = VV_EPI_BIAS; g.bias = cg.bias;
            slots(g, cg.rows_per_frame, s.sl_stride, 0);
            GEMM(g);
        }
    }
    return 0;
}

static int zero_codec(vv_ctx* ctx, CodecNet& net, int sl, hipStream_t st) {
    void* tab; int nt;
    if (codec_tables(ctx, net, sl, 1, &tab, &nt)) return -1;
    ctx->launches++;
    VVCHK(vv_zero_hist_launch(tab, nt, st));
    return 0;
}

// ------------------------------------------------------------------ graphs
template <class F>
static int graphed(vv_ctx* ctx, const std::string& key, hipStream_t st, F&& body) {
    if (!ctx->c.use_graph || ctx->prof_on) { VV_SHARED; return body(); }
    auto it = ctx->graphs.find(key);
    if (it == ctx->graphs.end()) {
        // first sight of a key: run eagerly (lazy allocations, shift tables) and remember it; a key that comes back is
        // captured then.  One-off launch shapes (prompt prefill chunks: unique pointers) never pay for a capture.
        if (ctx->seen.size() > 8192) ctx->seen.clear();
        if (ctx->seen.insert(key).second) { VV_SHARED; return body(); }
        hipGraph_t graph = nullptr;
        GraphEntry ge; ge.last_use = 0;
        bool captured = false;
        {   // one capture at a time in the process: contexts sharing weights are driven from several host threads (Engine.fork)
            std::unique_lock<std::shared_mutex> lk(g_dev_mu);
            HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
            const int r = body();
            const hipError_t e = hipStreamEndCapture(st, &graph);
            if (r == 0 && e == hipSuccess) {
                size_t nn = 0;
                if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess && nn) {
                    std::vector<hipGraphNode_t> nodes(nn);
                    if (hipGraphGetNodes(graph, nodes.data(), &nn) == hipSuccess)
                        for (size_t i = 0; i < nn; ++i) { hipGraphNodeType t; if (hipGraphNodeGetType(nodes[i], &t) == hipSuccess && t != hipGraphNodeTypeKernel) ctx->foreign_nodes++; }
                }
                HIPCHK(ctx, hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0));
                hipGraphDestroy(graph);
                captured = true;
            } else {
                // A capture that did not close cleanly executed nothing.  Known cause: ANOTHER host thread called a device-wide
                // synchronize (hipDeviceSynchronize / torch.cuda.synchronize()) while this capture was open -- ROCm 7.0.2 refuses
                // that call in its thread and marks this capture invalid although its mode is Relaxed.  The work is run the way the
                // first sight of a key runs (eagerly).  On that ROCm the stream itself does not recover from an invalidated capture
                // (every later call on it answers hipErrorStreamCaptureInvalidated; ending the capture a second time or destroying
                // the returned graph handle crashes inside the runtime -- both tried): then the eager run fails too and the message
                // below says why.  Callers keep device-wide synchronizes out of processes that generate.
                (void)hipGetLastError();
                ctx->capture_fallbacks++;
                snprintf(ctx->last_capture_issue, sizeof(ctx->last_capture_issue), "capture of '%s': body rc %d, hipStreamEndCapture: %s", key.c_str(), r, hipGetErrorString(e));
            }
        }
        if (!captured) {
            VV_SHARED;
            const int r2 = body();
            if (r2 != 0) {
                char prev[200]; snprintf(prev, sizeof(prev), "%s", ctx->err);
                return fail(ctx, "%s; the eager re-run failed as well (%s).  If another host thread called a device-wide synchronize "
                                 "(torch.cuda.synchronize()) during the capture, the stream is lost: synchronize streams or events instead", ctx->last_capture_issue, prev);
            }
            return 0;
        }
        if (ctx->graphs.size() >= ctx->graph_cap) {
            // a long-running process with varied launch shapes (prefill remainders over temporary buffers) must not
            // accumulate executables: drop the least-recently-used quarter.  Rare (a cache miss at the cap), so it may wait
            // for every stream: replays of the victims may still be running on this or a side stream.
            std::vector<std::pair<uint64_t, std::string>> order;
            for (auto& g : ctx->graphs) order.push_back({g.second.last_use, g.first});
            std::sort(order.begin(), order.end());
            std::unique_lock<std::shared_mutex> lk(g_dev_mu);      // a device-wide synchronize: never while another context's capture is open
            HIPCHK(ctx, hipDeviceSynchronize());
            for (size_t i = 0; i < order.size() / 4 + 1; ++i) {
                auto v = ctx->graphs.find(order[i].second);
                hipGraphExecDestroy(v->second.exec);
                ctx->graphs.erase(v);
            }
        }
        it = ctx->graphs.emplace(key, ge).first;
    }
    it->second.last_use = ++ctx->graph_tick;
    VV_SHARED;
    HIPCHK(ctx, hipGraphLaunch(it->second.exec, st));
    return 0;
}

// ------------------------------------------------------------------ API
extern "C" const char* vv_last_error(vv_ctx* ctx) { return ctx ? ctx->err : g_err; }

// Content hash of the sources this library was compiled from (vibevoice_amd/build.py passes -DVV_BUILD_ID): the loader
// compares it with the hash of the sources next to it, so a stale in-tree binary is rebuilt or refused instead of silently run.
#ifndef VV_BUILD_ID
#define VV_BUILD_ID "unknown"
#endif
extern "C" const char* vv_build_id() { return "VVHIP_BUILD_ID=" VV_BUILD_ID; }

static int create_impl(const vv_config* cfg, vv_ctx* parent, vv_ctx** out) {
    VV_SHARED;
    vv_ctx* ctx = new vv_ctx();
    ctx->c = *cfg; ctx->err[0] = 0;
    ctx->parent = parent; ctx->creating = true;
    { const char* e = getenv("VVHIP_FOLD_NORMDW"); if (e && e[0] == '0') ctx->fold_normdw = false; }
    vv_config& c = ctx->c;
    if (c.max_rows < 1 || c.max_rows > 16384) { delete ctx; return fail(nullptr, "max_rows must be in [1,16384]"); }
    if (c.lm_head_dim != 64 && c.lm_head_dim != 128) { delete ctx; return fail(nullptr, "head_dim must be 64 or 128"); }
    // the attention kernels put the query heads of a GQA group on the 16 MFMA columns (attn.hip)
    if (c.lm_kv_heads < 1 || c.lm_heads % c.lm_kv_heads != 0 || c.lm_heads / c.lm_kv_heads > 16) {
        const int hq = c.lm_heads, hkv = c.lm_kv_heads;
        delete ctx;
        return fail(nullptr, "lm_heads=%d / lm_kv_heads=%d: the GQA group size must be an integer <= 16", hq, hkv);
    }
    if (c.xsplit < 1 || c.xsplit > 3) c.xsplit = 2;
    if (c.attn_splits < 1) c.attn_splits = 128;
    if (c.enc_frames < 1) c.enc_frames = 1;
    if (getenv("VVHIP_GRAPH_CAP")) ctx->graph_cap = (size_t)std::max(4, atoi(getenv("VVHIP_GRAPH_CAP")));
    c.max_ctx = (c.max_ctx + 127) & ~127;
    const int H = ctx->H = c.lm_hidden, D = ctx->D = c.lm_head_dim, Hq = ctx->Hq = c.lm_heads, Hkv = ctx->Hkv = c.lm_kv_heads;
    const int I = ctx->I = c.lm_inter;
    const int QKV = ctx->QKV = (Hq + 2 * Hkv) * D;
    const int R = c.max_rows;
    // ---- LM ----
    ctx->embed = walloc(ctx, (size_t)c.lm_vocab * H * 2);
    { int i = add_w(ctx, "lm.embed_tokens.weight", W_TABLE, (int64_t)c.lm_vocab * H); ctx->w[i].dev = ctx->embed; }
    { int i = add_w(ctx, "lm_head.weight", W_TABLE, (int64_t)c.lm_vocab * H, true); ctx->w[i].dev = nullptr; }
    ctx->inv_freq = add_vec(ctx, "lm.rope.inv_freq", D / 2);
    ctx->lm_norm = add_vec(ctx, "lm.norm.weight", H);
    if (c.tts_layers > 0) {
        if (c.tts_layers >= c.lm_layers) { delete ctx; return fail(nullptr, "tts_layers must be < lm_layers"); }
        ctx->tts_types = add_vec(ctx, "tts_input_types.weight", 2 * (int64_t)H);
        ctx->eos_w1 = alloc_packed(ctx, H, H); add_mat(ctx, "eos.fc1.weight", H, H, ctx->eos_w1, 0);
        ctx->eos_b1 = add_vec(ctx, "eos.fc1.bias", H);
        ctx->eos_w2 = alloc_packed(ctx, 1, H); add_mat(ctx, "eos.fc2.weight", 1, H, ctx->eos_w2, 0);
        ctx->eos_b2 = add_vec(ctx, "eos.fc2.bias", 1);
    }
    ctx->layers.resize(c.lm_layers);
    for (int l = 0; l < c.lm_layers; ++l) {
        auto& L = ctx->layers[l];
        char nm[128]; snprintf(nm, 128, "lm.layers.%d.", l);
        std::string p(nm);
        L.ln1 = add_vec(ctx, p + "input_layernorm.weight", H);
        L.ln2 = add_vec(ctx, p + "post_attention_layernorm.weight", H);
        L.wqkv = alloc_packed(ctx, QKV, H);
        L.bqkv = (float*)walloc(ctx, (size_t)QKV * 4);
        add_mat(ctx, p + "self_attn.q_proj.weight", Hq * D, H, L.wqkv, 0);
        add_mat(ctx, p + "self_attn.k_proj.weight", Hkv * D, H, L.wqkv, Hq * D / 16);
        add_mat(ctx, p + "self_attn.v_proj.weight", Hkv * D, H, L.wqkv, (Hq + Hkv) * D / 16);
        add_vec(ctx, p + "self_attn.q_proj.bias", Hq * D, L.bqkv);
        add_vec(ctx, p + "self_attn.k_proj.bias", Hkv * D, L.bqkv + Hq * D);
        add_vec(ctx, p + "self_attn.v_proj.bias", Hkv * D, L.bqkv + (Hq + Hkv) * D);
        L.wo = alloc_packed(ctx, H, Hq * D); add_mat(ctx, p + "self_attn.o_proj.weight", H, Hq * D, L.wo, 0);
        L.wg = alloc_packed(ctx, I, H); add_mat(ctx, p + "mlp.gate_proj.weight", I, H, L.wg, 0);
        L.wu = alloc_packed(ctx, I, H); add_mat(ctx, p + "mlp.up_proj.weight", I, H, L.wu, 0);
        L.wd = alloc_packed(ctx, H, I); add_mat(ctx, p + "mlp.down_proj.weight", H, I, L.wd, 0);
    }
    const int n_caches = 2 * c.n_slots;
    ctx->head_stride = (int64_t)c.max_ctx * D;
    ctx->layer_stride = ctx->head_stride * Hkv;
    ctx->cache_stride = ctx->layer_stride * c.lm_layers;
    ctx->kc = dalloc(ctx, (size_t)ctx->cache_stride * n_caches * 2);
    ctx->vc = dalloc(ctx, (size_t)ctx->cache_stride * n_caches * 2);
    ctx->rows_cap = std::max(2048, c.max_rows);
    ctx->rows_dev = (VVRow*)dalloc(ctx, sizeof(VVRow) * ctx->rows_cap);
    hipHostMalloc((void**)&ctx->rows_pin, sizeof(VVRow) * (size_t)ctx->rows_cap * vv_ctx::RING);
    for (int i = 0; i < vv_ctx::RING; ++i) hipEventCreateWithFlags(&ctx->ring_ev[i], hipEventDisableTiming);
    ctx->ids_cap = std::max(64, c.max_rows);
    ctx->ids_dev = (int*)dalloc(ctx, sizeof(int) * ctx->ids_cap);
    hipHostMalloc((void**)&ctx->ids_pin, sizeof(int) * (size_t)ctx->ids_cap * vv_ctx::RING);
    ctx->h = (float*)dalloc(ctx, (size_t)R * H * 4);
    ctx->h_parts = (float*)dalloc(ctx, (size_t)2 * R * H * 4);
    ctx->qkv = (float*)dalloc(ctx, (size_t)R * QKV * 4);
    ctx->qrot = (float*)dalloc(ctx, (size_t)R * Hq * D * 4);
    ctx->attn = (float*)dalloc(ctx, (size_t)R * Hq * D * 4);
    ctx->act = (float*)dalloc(ctx, (size_t)R * I * 4);
    if (R >= 64 && (H % 8) == 0 && ((Hq * D) % 8) == 0 && (I % 8) == 0) {
        // prompt prefill in bf16-activation mode: LDS-staged 128 x 128 MFMA GEMM over packed activations (prefill.hip)
        ctx->xp = dalloc(ctx, (size_t)vv_packed_elems(R, std::max(H, Hq * D)) * 2);
        ctx->actp = dalloc(ctx, (size_t)vv_packed_elems(R, I) * 2);
        ctx->tile3_ok = c.xsplit == 1;
        if (ctx->tile3_ok) {      // short prompts: K parts of the 128 x 128 GEMM (up to 8 dense [rows][features] fp32 tensors, rows <= 1024)
            ctx->gws.g3_bytes = (size_t)8 * std::min(R, 1024) * std::max(ctx->QKV, H) * 4;
            ctx->gws.g3_partials = (float*)dalloc(ctx, ctx->gws.g3_bytes, false);
            if (!ctx->gws.g3_partials) ctx->gws.g3_bytes = 0;
        }
        const char* no_ks = getenv("VVHIP_NO_KSPLIT");  // opt-out: every tile of the partial round computed whole (no inter-workgroup hand-off)
        if (ctx->tile3_ok && R >= 1024 && !(no_ks && no_ks[0] == '1')) {   // prompts long enough for the 256 x 256 GEMM: its partial round is split along K
            ctx->gws.partials = (float*)dalloc(ctx, (size_t)256 * 32 * 512 * 16, false);
            ctx->gws.flags = (unsigned*)dalloc(ctx, 256 * sizeof(unsigned));
            if (hipHostMalloc((void**)&ctx->gws.err, sizeof(unsigned), hipHostMallocMapped) == hipSuccess && ctx->gws.err) *ctx->gws.err = 0u;
            else ctx->gws.err = nullptr;
        }
    }
    ctx->attn2_ok = c.xsplit == 1;
    if (c.xsplit == 1 && R > 4 && (H % 32) == 0 && ((Hq * D) % 32) == 0 && (I % 32) == 0) {
        const int kx = std::max(H, Hq * D), ka = std::max(I, c.head_ffn);
        ctx->p16_x = dalloc(ctx, (size_t)vv_packed_elems(16, kx) * 2);
        ctx->p16_act = dalloc(ctx, (size_t)vv_packed_elems(16, ka + 32) * 2);
        ctx->p16_ok = ctx->p16_x && ctx->p16_act;
        const char* fz = getenv("VVHIP_P16_FUSE");
        if (ctx->p16_ok && !(fz && fz[0] == '0') && (H % 16) == 0) {
            ctx->p16_y = dalloc(ctx, (size_t)vv_packed_elems(16, kx) * 2);
            ctx->ssq_a = (float*)dalloc(ctx, (size_t)(H / 16) * 16 * 4);
            ctx->ssq_b = (float*)dalloc(ctx, (size_t)(H / 16) * 16 * 4);
            ctx->p16_fuse = ctx->p16_y && ctx->ssq_a && ctx->ssq_b;
            const char* hs = getenv("VVHIP_P16_HEAD_SH");
            ctx->p16_head_sh = hs && hs[0] == '1';
        }
    }
    ctx->rope_tab = dalloc(ctx, (size_t)c.max_ctx * (D / 2) * 8, false);
    // split-attention partials exist for decode rows and short ragged launches only (prompt chunks use the prefill kernel)
    ctx->ws_rows = std::min(R, 64);
    const size_t np = (size_t)ctx->ws_rows * Hkv * c.attn_splits * 16;
    ctx->pm = (float*)dalloc(ctx, np * 4); ctx->pl = (float*)dalloc(ctx, np * 4); ctx->po = (float*)dalloc(ctx, np * D * 4);
    // ---- diffusion head ----
    const int L = c.latent_dim, HL = c.head_layers, HF = ctx->HF = c.head_ffn;
    const int MODW = ctx->MODW = HL * 3 * H + 2 * H;
    ctx->h_in = alloc_packed(ctx, H, L); add_mat(ctx, "head.noisy_images_proj.weight", H, L, ctx->h_in, 0);
    ctx->h_cond = alloc_packed(ctx, H, H); add_mat(ctx, "head.cond_proj.weight", H, H, ctx->h_cond, 0);
    ctx->h_t0 = alloc_packed(ctx, H, 256); add_mat(ctx, "head.t_embedder.mlp.0.weight", H, 256, ctx->h_t0, 0);
    ctx->h_t2 = alloc_packed(ctx, H, H); add_mat(ctx, "head.t_embedder.mlp.2.weight", H, H, ctx->h_t2, 0);
    ctx->h_ada = alloc_packed(ctx, MODW, H);
    ctx->hl.resize(HL);
    for (int l = 0; l < HL; ++l) {
        char nm[128]; snprintf(nm, 128, "head.layers.%d.", l);
        std::string p(nm);
        ctx->hl[l].norm = add_vec(ctx, p + "norm.weight", H);
        add_mat(ctx, p + "adaLN_modulation.1.weight", 3 * H, H, ctx->h_ada, l * 3 * H / 16);
        ctx->hl[l].wg = alloc_packed(ctx, HF, H); add_mat(ctx, p + "ffn.gate_proj.weight", HF, H, ctx->hl[l].wg, 0);
        ctx->hl[l].wu = alloc_packed(ctx, HF, H); add_mat(ctx, p + "ffn.up_proj.weight", HF, H, ctx->hl[l].wu, 0);
        ctx->hl[l].wd = alloc_packed(ctx, H, HF); add_mat(ctx, p + "ffn.down_proj.weight", H, HF, ctx->hl[l].wd, 0);
    }
    add_mat(ctx, "head.final_layer.adaLN_modulation.1.weight", 2 * H, H, ctx->h_ada, HL * 3 * H / 16);
    ctx->h_out = alloc_packed(ctx, L, H); add_mat(ctx, "head.final_layer.linear.weight", L, H, ctx->h_out, 0);
    const int R2 = 16;
    ctx->cproj = (float*)dalloc(ctx, (size_t)R2 * H * 4);
    ctx->mod = (float*)dalloc(ctx, (size_t)R2 * MODW * 4);
    ctx->zz = (float*)dalloc(ctx, (size_t)R2 * L * 4);
    ctx->x0p = (float*)dalloc(ctx, (size_t)R2 * L * 4);
    ctx->xh = (float*)dalloc(ctx, (size_t)R2 * H * 4);
    ctx->zz2 = (float*)dalloc(ctx, (size_t)R2 * L * 4);
    ctx->x0p2 = (float*)dalloc(ctx, (size_t)R2 * L * 4);
    ctx->xh2 = (float*)dalloc(ctx, (size_t)R2 * H * 4);
    {   // decode rows of the bf16 mode: final layer + solver update + next in-projection as one launch (headtail.hip)
        const char* e = getenv("VVHIP_HEAD_TAIL");
        const int tpw = e ? atoi(e) : 4;
        ctx->head_tail_tpw = (c.xsplit == 1 && L == 64 && (H % 32) == 0 && (tpw == 1 || tpw == 2 || tpw == 4 || tpw == 8)) ? tpw : 0;
    }
    ctx->xh_parts = (float*)dalloc(ctx, (size_t)4 * R2 * H * 4);      // two generations: a layer reads one while writing the other
    ctx->hact = (float*)dalloc(ctx, (size_t)R2 * HF * 4);
    ctx->eps = (float*)dalloc(ctx, (size_t)R2 * L * 4);
    ctx->tmp1 = (float*)dalloc(ctx, (size_t)64 * H * 4);
    ctx->tmp2 = (float*)dalloc(ctx, (size_t)64 * 256 * 4);
    // ---- connectors ----
    auto mk_conn = [&](vv_ctx::Conn& cn, const std::string& p, int din) {
        cn.fc1 = alloc_packed(ctx, H, din); add_mat(ctx, p + "fc1.weight", H, din, cn.fc1, 0);
        cn.b1 = add_vec(ctx, p + "fc1.bias", H);
        cn.norm = add_vec(ctx, p + "norm.weight", H);
        cn.fc2 = alloc_packed(ctx, H, H); add_mat(ctx, p + "fc2.weight", H, H, cn.fc2, 0);
        cn.b2 = add_vec(ctx, p + "fc2.bias", H);
    };
    mk_conn(ctx->ac_conn, "ac_conn.", L);
    if (c.sem_dim > 0) mk_conn(ctx->sem_conn, "sem_conn.", c.sem_dim);
    ctx->ct1 = (float*)dalloc(ctx, (size_t)256 * H * 4);
    // ---- codecs ----
    if (build_codec(ctx, ctx->dec, "dec.", true, L, 1, c.n_slots)) { *out = ctx; return -1; }
    if (c.sem_dim > 0 && build_codec(ctx, ctx->senc, "senc.", false, c.sem_dim, 1, c.n_slots)) { *out = ctx; return -1; }
    if (c.has_acoustic_encoder && build_codec(ctx, ctx->aenc, "aenc.", false, L, c.enc_frames, 1)) { *out = ctx; return -1; }
    if (hipDeviceSynchronize() != hipSuccess || ctx->err[0]) { *out = ctx; return fail(ctx, "vv_create: allocation failed: %s", ctx->err); }
    ctx->creating = false;
    if (parent) {
        if (ctx->wshare_i != parent->wallocs.size() || ctx->w.size() != parent->w.size()) {
            *out = ctx;
            return fail(ctx, "vv_create_shared: the child registered %zu weight buffers / %zu parameters, the parent %zu / %zu -- different model configuration",
                        ctx->wshare_i, ctx->w.size(), parent->wallocs.size(), parent->w.size());
        }
        for (size_t i = 0; i < ctx->w.size(); ++i) {
            if (ctx->w[i].name != parent->w[i].name || ctx->w[i].nelem != parent->w[i].nelem) { *out = ctx; return fail(ctx, "vv_create_shared: parameter table differs at '%s'", ctx->w[i].name.c_str()); }
            ctx->w[i].loaded = parent->w[i].loaded;
            if (ctx->w[i].name == "lm_head.weight") ctx->w[i].dev = parent->w[i].dev;
        }
        ctx->lm_head = parent->lm_head; ctx->lm_head_loaded = parent->lm_head_loaded;
        ctx->scaling = parent->scaling; ctx->bias = parent->bias;
        { std::lock_guard<std::mutex> fl(g_family_mu); parent->n_children++; }
    }
    *out = ctx;
    return 0;
}

extern "C" int vv_create(const vv_config* cfg, vv_ctx** out) { return create_impl(cfg, nullptr, out); }

// A second context over the SAME weight storage as `parent` (which must be fully uploaded and is not a shared child itself): own KV
// caches, activations, tokenizer state, graphs and staging, sized by cfg's runtime fields (n_slots, max_ctx, max_rows, attn_splits,
// use_graph); the model fields must equal the parent's.  Two such contexts, each driven on its own stream, interleave two independent
// decode chains on one GPU over one copy of the weights: one chain's launch boundaries are filled by the other's kernels.
extern "C" int vv_create_shared(const vv_config* cfg, vv_ctx* parent, vv_ctx** out) {
    if (!parent) return fail(nullptr, "vv_create_shared: no parent context");
    if (parent->parent) return fail(nullptr, "vv_create_shared: the parent is itself a shared context; share from the owner of the weights");
    for (auto& w : parent->w)
        if (!w.loaded && !w.optional) return fail(nullptr, "vv_create_shared: parent parameter '%s' is not uploaded yet", w.name.c_str());
    vv_config a = *cfg, b = parent->c;
    a.n_slots = b.n_slots; a.max_ctx = b.max_ctx; a.max_rows = b.max_rows; a.attn_splits = b.attn_splits; a.use_graph = b.use_graph;
    if (a.xsplit < 1 || a.xsplit > 3) a.xsplit = 2;
    if (a.enc_frames < 1) a.enc_frames = 1;
    if (memcmp(&a, &b, sizeof(vv_config)) != 0) return fail(nullptr, "vv_create_shared: the model fields of the configuration differ from the parent's");
    int r = create_impl(cfg, parent, out);
    if (r != 0 && *out) { vv_ctx* c = *out; c->parent = nullptr; }      // a failed child holds no reference
    return r;
}

static void destroy_impl(vv_ctx* ctx) {
    if (!ctx) return;
    hipDeviceSynchronize();
    {   // shared children still read these weights: the last of them frees
        std::lock_guard<std::mutex> fl(g_family_mu);
        if (ctx->n_children > 0) { ctx->zombie = true; return; }
    }
    vv_ctx* par = ctx->parent;
    for (auto& g : ctx->graphs) hipGraphExecDestroy(g.second.exec);
    if (ctx->side_ready) {
        for (int j = 0; j < 8; ++j) { hipStreamDestroy(ctx->side[j]); hipEventDestroy(ctx->ev_join[j]); }
        hipEventDestroy(ctx->ev_fork);
    }
    for (void* p : ctx->allocs) hipFree(p);          // weights, KV caches, state and scratch buffers
    ctx->allocs.clear();
    if (ctx->stage) hipFree(ctx->stage);
    hipHostFree(ctx->rows_pin); hipHostFree(ctx->ids_pin);
    if (ctx->gws.err) hipHostFree(ctx->gws.err);
    for (int i = 0; i < vv_ctx::RING; ++i) if (ctx->ring_ev[i]) hipEventDestroy(ctx->ring_ev[i]);
    delete ctx;
    bool last_child = false;
    if (par) { std::lock_guard<std::mutex> fl(g_family_mu); last_child = (--par->n_children == 0 && par->zombie); }
    if (last_child) destroy_impl(par);
}
extern "C" void vv_destroy(vv_ctx* ctx) {
    VV_SHARED;                   // device-wide synchronisation + frees: never while another context's capture is open
    destroy_impl(ctx);
}

extern "C" int vv_num_weights(vv_ctx* ctx) { return (int)ctx->w.size(); }
extern "C" int vv_weight_info(vv_ctx* ctx, int idx, char* name, int cap, int64_t* nelem, int* loaded) {
    if (idx < 0 || idx >= (int)ctx->w.size()) return fail(ctx, "weight index out of range");
    const Weight& w = ctx->w[idx];
    snprintf(name, cap, "%s", w.name.c_str());
    if (nelem) *nelem = w.nelem;
    if (loaded) *loaded = (w.loaded || w.optional) ? 1 : 0;
    return 0;
}

extern "C" int vv_upload(vv_ctx* ctx, const char* name, const void* src, int src_dtype, int64_t nelem) {
    VV_SHARED;
    auto it = ctx->widx.find(name);
    if (it == ctx->widx.end()) return fail(ctx, "unknown parameter '%s'", name);
    Weight& w = ctx->w[it->second];
    if (ctx->parent) return fail(ctx, "parameter '%s': this context shares its parent's weights -- upload through the parent", name);
    {   // a child snapshots what it derives from the parameters when it is created (lm_head / tied embedding table, the valid-token
        // rows, the RoPE table from inv_freq, the speech factors): an upload behind its back would leave those stale
        std::lock_guard<std::mutex> fl(g_family_mu);
        if (ctx->n_children > 0)
            return fail(ctx, "parameter '%s': %d shared context(s) were created from this one and hold snapshots derived from its parameters -- "
                             "destroy them (model.close_lanes()), upload / merge, then fork again", name, ctx->n_children);
    }
    if (nelem != w.nelem) return fail(ctx, "parameter '%s': expected %lld elements, got %lld", name, (long long)w.nelem, (long long)nelem);
    const size_t esz = src_dtype ? 2 : 4;
    hipPointerAttribute_t attr;
    bool on_dev = (hipPointerGetAttributes(&attr, src) == hipSuccess) && (attr.type == hipMemoryTypeDevice);
    (void)hipGetLastError();
    const void* dsrc = src;
    if (!on_dev) {
        const size_t need = (size_t)nelem * esz;
        if (need > ctx->stage_bytes) {
            if (ctx->stage) hipFree(ctx->stage);
            ctx->stage = nullptr; ctx->stage_bytes = 0;
            HIPCHK(ctx, hipMalloc(&ctx->stage, need));
            ctx->stage_bytes = need;
        }
        HIPCHK(ctx, hipMemcpy(ctx->stage, src, need, hipMemcpyHostToDevice));
        dsrc = ctx->stage;
    }
    hipStream_t st = 0;
    if (w.kind == W_TABLE) {
        if (w.name == "lm_head.weight" && !ctx->lm_head_loaded) {
            ctx->lm_head = dalloc(ctx, (size_t)nelem * 2, false);
            if (!ctx->lm_head) return -1;
            ctx->lm_head_loaded = true; w.dev = ctx->lm_head;
        }
        if (src_dtype) HIPCHK(ctx, hipMemcpy(w.dev, dsrc, (size_t)nelem * 2, hipMemcpyDeviceToDevice));
        else VVCHK(vv_cvt_launch(dsrc, w.dev, nelem, 1, st));
    } else if (w.kind == W_MAT) {
        VVCHK(vv_pack_launch(dsrc, src_dtype, w.dev, w.N, w.K, w.pk, w.Cin, w.Cout, w.ksz, w.stride, st));
    } else {
        // fp32 vectors
        const float* f32 = (const float*)dsrc;
        float* tmp = nullptr;
        if (src_dtype) {
            tmp = (float*)dalloc(ctx, (size_t)nelem * 4, false);
            if (!tmp) return -1;
            VVCHK(vv_cvt_launch(dsrc, tmp, nelem, 0, st));
            f32 = tmp;
        }
        if (w.kind == W_DW) {
            VVCHK(vv_dw_transpose_launch(f32, (float*)w.dev, (int)(nelem / 7), st));
        } else if (w.kind == W_BIAS_REP) {
            for (int r = 0; r < w.rep; ++r)
                HIPCHK(ctx, hipMemcpyAsync((float*)w.dev + (size_t)r * nelem, f32, (size_t)nelem * 4, hipMemcpyDeviceToDevice, st));
        } else {
            HIPCHK(ctx, hipMemcpyAsync(w.dev, f32, (size_t)nelem * 4, hipMemcpyDeviceToDevice, st));
        }
        HIPCHK(ctx, hipStreamSynchronize(st));
        dfree(ctx, tmp);
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    w.loaded = true;
    ctx->rope_ready = false;      // (cos, sin) table is rebuilt from the current inv_freq on the next decode step
    return 0;
}

extern "C" int vv_set_speech_factors(vv_ctx* ctx, float scaling, float bias) {
    VV_SHARED;
    ctx->scaling = scaling; ctx->bias = bias;
    for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();) {
        if (it->first.rfind("dec", 0) == 0) { hipGraphExecDestroy(it->second.exec); it = ctx->graphs.erase(it); } else ++it;
    }
    return 0;
}

extern "C" int vv_set_valid_tokens(vv_ctx* ctx, const int* ids, int n) {
    VV_SHARED;
    if (n < 1 || n > 16) return fail(ctx, "n_valid must be in [1,16]");
    const int H = ctx->H;
    const void* table = ctx->lm_head_loaded ? ctx->lm_head : ctx->embed;
    void* rows = dalloc(ctx, (size_t)n * H * 2, false);
    if (!rows) return -1;
    for (int i = 0; i < n; ++i) {
        if (ids[i] < 0 || ids[i] >= ctx->c.lm_vocab) return fail(ctx, "token id %d out of range", ids[i]);
        HIPCHK(ctx, hipMemcpy((char*)rows + (size_t)i * H * 2, (const char*)table + (size_t)ids[i] * H * 2, (size_t)H * 2, hipMemcpyDeviceToDevice));
    }
    if (!ctx->valid_w) ctx->valid_w = alloc_packed(ctx, 16, H);
    VVCHK(vv_pack_launch(rows, 1, ctx->valid_w, n, H, 0, 0, 0, 0, 0, 0));
    HIPCHK(ctx, hipDeviceSynchronize());
    dfree(ctx, rows);
    ctx->n_valid = n;
    return 0;
}

static int set_schedule(vv_ctx* ctx, int n_steps, const float* t, const float* coef, int width, void* stream);
// coef: n_steps rows {a, s, cs, c0, c1} (the deterministic DPM-Solver++(2M) the model classes build)
extern "C" int vv_set_schedule(vv_ctx* ctx, int n_steps, const float* t, const float* coef, void* stream) {
    return set_schedule(ctx, n_steps, t, coef, 5, stream);
}
// coef: n_steps rows {a, s, cs, c0, c1, cn} -- sde-dpmsolver++ (demo/gradio_demo.py:142-146); sampling then needs the per-step
// variance noise: vv_diffusion_sample_sde
extern "C" int vv_set_schedule_sde(vv_ctx* ctx, int n_steps, const float* t, const float* coef6, void* stream) {
    return set_schedule(ctx, n_steps, t, coef6, 6, stream);
}
static int set_schedule(vv_ctx* ctx, int n_steps, const float* t, const float* coef, int width, void* stream) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (n_steps < 1 || n_steps > 64) return fail(ctx, "n_steps must be in [1,64]");
    const int H = ctx->H;
    if (!ctx->temb) {
        ctx->temb = (float*)dalloc(ctx, (size_t)64 * H * 4);
        ctx->coef = (float*)dalloc(ctx, 64 * 6 * 4);
        ctx->tvals = (float*)dalloc(ctx, 64 * 4);
    }
    float rows6[64 * 6];
    for (int i = 0; i < n_steps; ++i)
        for (int j = 0; j < 6; ++j) rows6[i * 6 + j] = (j < width) ? coef[i * width + j] : 0.f;
    // by the API used, not by the values: a one-step stochastic schedule has sigma_t = 0 on its only step (all noise scales
    // zero) and must still be sampled through vv_diffusion_sample_sde, as the reference runs it (a noise-free first-order step)
    ctx->sde_on = (width == 6);
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipMemcpy(ctx->coef, rows6, (size_t)n_steps * 6 * 4, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->tvals, t, (size_t)n_steps * 4, hipMemcpyHostToDevice));
    // t_emb[i] = W2 . silu(W1 . sinusoid(t_i))   (TimestepEmbedder, modular_vibevoice_diffusion_head.py:66-93)
    VVCHK(vv_tfreq_launch(ctx->tvals, ctx->tmp2, n_steps, st));
    for (int i0 = 0; i0 < n_steps; i0 += 16) {
        const int nn = std::min(16, n_steps - i0);
        VVGemm g = mk_gemm(ctx->h_t0, ctx->tmp2 + (size_t)i0 * 256, ctx->tmp1 + (size_t)i0 * H, nn, H, 256, 256, H);
        GEMM(g);
    }
    VVCHK(vv_silu_launch(ctx->tmp1, n_steps * H, st));
    for (int i0 = 0; i0 < n_steps; i0 += 16) {
        const int nn = std::min(16, n_steps - i0);
        VVGemm g = mk_gemm(ctx->h_t2, ctx->tmp1 + (size_t)i0 * H, ctx->temb + (size_t)i0 * H, nn, H, H, H, H);
        GEMM(g);
    }
    HIPCHK(ctx, hipStreamSynchronize(st));
    ctx->n_steps = n_steps;
    {   // room for the batched adaLN modulations of up to 8 sampled utterances (16 rows) per step
        const size_t need = (size_t)n_steps * 16 * ctx->MODW * 4;
        if (need > ctx->mod_all_bytes) {
            dfree(ctx, ctx->mod_all);
            dfree(ctx, ctx->ada_in);
            ctx->mod_all = (float*)dalloc(ctx, need, false);
            ctx->ada_in = (float*)dalloc(ctx, (size_t)n_steps * 16 * ctx->H * 4, false);
            dfree(ctx, ctx->ada_p);
            ctx->ada_p = (ctx->c.xsplit == 1 && (ctx->H & 7) == 0) ? dalloc(ctx, (size_t)vv_packed_elems(n_steps * 16, ctx->H) * 2, false) : nullptr;
            ctx->mod_all_bytes = (ctx->mod_all && ctx->ada_in) ? need : 0;
            dfree(ctx, ctx->p16_shift); ctx->p16_shift = nullptr;
            if (ctx->p16_fuse) {          // the adaLN shift rows of every (solver step, layer) as packed bf16 operand tiles
                ctx->p16_shift_tile = (size_t)vv_packed_elems(16, ctx->H) * 2;
                ctx->p16_shift = dalloc(ctx, (size_t)n_steps * (ctx->c.head_layers + 1) * ctx->p16_shift_tile);
            }
        }
    }
    for (auto it = ctx->graphs.begin(); it != ctx->graphs.end();) {
        if (it->first.rfind("samp", 0) == 0 || it->first.rfind("sde:", 0) == 0) { hipGraphExecDestroy(it->second.exec); it = ctx->graphs.erase(it); } else ++it;
    }
    return 0;
}

// batch-decode projection (gemv16p.hip); inside a profile window the launch is also recorded for the family replay
static int p16_gemv(vv_ctx* ctx, hipStream_t st, const void* W, const void* W2, const void* Xp, float* Y, void* Yp, const float* bias,
                    const float* gate, int T, int N, int K, int ldy, int ld_gate, int epi) {
    if (ctx->prof_on) {
        const double by = (double)vv_packed_elems(N, K) * 2.0 * (W2 ? 2.0 : 1.0) + (double)vv_packed_elems(16, K) * 2.0 +
                          (Yp ? (double)T * N * 2.0 : (double)T * N * 4.0 * (epi == VV_EPI_RESID || epi == VV_EPI_GATED_RESID ? 2.0 : 1.0));
        ctx->prof_other.push_back({1, by, [=](hipStream_t s) { return vv_gemv16p_launch(W, W2, Xp, Y, Yp, bias, gate, T, N, K, ldy, ld_gate, epi, s); }});
    }
    return vv_gemv16p_launch(W, W2, Xp, Y, Yp, bias, gate, T, N, K, ldy, ld_gate, epi, st);
}

// the struct form (round 6: RS / SH / PK operands); recorded for the family replay like p16_gemv
static int p16_go(vv_ctx* ctx, hipStream_t st, const VVGemv16p& a, int epi, int flags) {
    if (ctx->prof_on) {
        const double by = (double)vv_packed_elems(a.N, a.K) * 2.0 * (a.W2 ? 2.0 : 1.0) + (double)vv_packed_elems(16, a.K) * 2.0 * ((flags & 2) ? 2.0 : 1.0) +
                          ((epi == VV_EPI_SWIGLU) ? (double)a.T * a.N * 2.0 : (double)a.T * a.N * 4.0 * (epi == VV_EPI_RESID || epi == VV_EPI_GATED_RESID ? 2.0 : 1.0)) +
                          ((flags & 4) ? (double)a.T * a.N * 2.0 : 0.0);
        const VVGemv16p ac = a;
        ctx->prof_other.push_back({1, by, [=](hipStream_t s) { return vv_gemv16p_launch2(&ac, epi, flags, s); }});
    }
    return vv_gemv16p_launch2(&a, epi, flags, st);
}
static VVGemv16p p16_args(const void* W, const void* W2, const void* Xp, float* Y, void* Yp, int T, int N, int K, int ldy) {
    VVGemv16p a{};
    a.W = (const u32x4*)W; a.W2 = (const u32x4*)W2; a.Xp = (const u32x4*)Xp; a.Y = Y; a.Yp = (unsigned char*)Yp;
    a.T = T; a.N = N; a.K = K; a.ldy = ldy;
    return a;
}

static int lm_body(vv_ctx* ctx, hipStream_t st, int R, const float* x_in, float* hidden_out, int l0, int l1, int final_norm, bool fused_attn, bool contiguous,
                   int attn_S, int64_t kv_positions = 0, int attn_W = 4) {
    const vv_config& c = ctx->c;
    const int H = ctx->H, D = ctx->D, Hq = ctx->Hq, Hkv = ctx->Hkv, I = ctx->I, QKV = ctx->QKV;
    VVCHK(vv_copy_launch(ctx->h, x_in, (size_t)R * H * 4, st));        // copies / fills inside captured sequences are kernels, never memcpy / memset nodes (misc.hip)
    int hp = 0;                                    // extra parts the residual stream h currently consists of
    const int hps = ctx->c.max_rows * H;
    if (contiguous && ctx->tile3_ok && R >= 64) {
        // ---- prompt prefill, bf16-activation mode: packed activations + LDS-staged MFMA GEMMs + 64-row prefill attention ----
        // cache slots past the chunk inside its last 64-position stage: V zeroed once for every layer of this pass (0 x NaN, see misc.hip)
        ctx->launches++;
        VVCHK(vv_kv_zero_v_tail_launch((char*)ctx->vc + (size_t)l0 * ctx->layer_stride * 2, ctx->rows_dev, R, l1 - l0, Hkv, D, ctx->cache_stride,
                                       ctx->layer_stride, ctx->head_stride, ctx->c.max_ctx, st));
        for (int l = l0; l < l1; ++l) {
            auto& L = ctx->layers[l];
            char* kl = (char*)ctx->kc + (size_t)l * ctx->layer_stride * 2;
            char* vl = (char*)ctx->vc + (size_t)l * ctx->layer_stride * 2;
            ctx->launches += 9;
            VVCHK(vv_pack_rows_launch(ctx->h, H, L.ln1, c.lm_eps, ctx->xp, R, H, st));
            // long prompts at head_dim 128: bias + RoPE + cache append in the QKV GEMM's epilogue; otherwise GEMM, then vv_rope_append
            const int fq = vv_gemm_qkv_rope_launch(L.wqkv, ctx->xp, L.bqkv, R, H, D, Hq, Hkv, ctx->rows_dev, ctx->rope_tab, ctx->qrot, kl, vl,
                                                   ctx->cache_stride, ctx->head_stride, &ctx->gws, st);
            if (fq < 0) return fail(ctx, "vv_gemm_qkv_rope_launch failed (%d)", fq);
            if (fq == 0) {
                VVCHK(vv_gemm3_launch(L.wqkv, nullptr, ctx->xp, ctx->qkv, nullptr, L.bqkv, R, QKV, H, QKV, VV_EPI_BIAS, &ctx->gws, st));
                VVCHK(vv_rope_append_launch(D, ctx->qkv, ctx->rows_dev, ctx->inv_freq, ctx->qrot, kl, vl,
                                            R, Hq, Hkv, ctx->cache_stride, ctx->head_stride, st));
            }
            // the attention writes the o-projection's packed operand itself (K = Hq * D: whole 32-wide k-tiles)
            const bool apk = ((Hq * D) & 31) == 0;
            VVCHK(vv_attn_prefill4_launch(D, ctx->qrot, ctx->rows_dev, kl, vl, R, Hq, Hkv, ctx->cache_stride, ctx->head_stride, ctx->attn,
                                          apk ? ctx->xp : nullptr, st));
            if (!apk) VVCHK(vv_pack_rows_launch(ctx->attn, Hq * D, nullptr, 0.f, ctx->xp, R, Hq * D, st));
            VVCHK(vv_gemm3_launch(L.wo, nullptr, ctx->xp, ctx->h, nullptr, nullptr, R, H, Hq * D, H, VV_EPI_RESID, &ctx->gws, st));
            VVCHK(vv_pack_rows_launch(ctx->h, H, L.ln2, c.lm_eps, ctx->xp, R, H, st));
            VVCHK(vv_gemm3_launch(L.wg, L.wu, ctx->xp, nullptr, ctx->actp, nullptr, R, I, H, 0, VV_EPI_SWIGLU, &ctx->gws, st));
            VVCHK(vv_gemm3_launch(L.wd, nullptr, ctx->actp, ctx->h, nullptr, nullptr, R, H, I, H, VV_EPI_RESID, &ctx->gws, st));
        }
        ctx->launches++;
        if (final_norm) VVCHK(vv_rmsnorm_rows_launch(ctx->h, H, hidden_out, H, ctx->lm_norm, R, H, c.lm_eps, st));
        else VVCHK(vv_copy_launch(hidden_out, ctx->h, (size_t)R * H * 4, st));
        return 0;
    }
    const bool p16 = R > 4 && R <= 16 && ctx->p16_ok && fused_attn;      // batch decode rows: packed-activation projections
    for (int l = l0; l < l1; ++l) {
        auto& L = ctx->layers[l];
        if (p16 && ctx->p16_fuse && l > l0) {
            // the previous layer's down projection left x * ln1 packed in p16_x and the rows' partial sums of squares in ssq_b
            ctx->launches += 1;
            VVGemv16p a = p16_args(L.wqkv, nullptr, ctx->p16_x, ctx->qkv, nullptr, R, QKV, H, QKV);
            a.bias = L.bqkv; a.ssq_in = ctx->ssq_b; a.ssq_tiles = H / 16; a.eps = c.lm_eps;
            VVCHK(p16_go(ctx, st, a, VV_EPI_BIAS, 1));
        } else if (p16) {
            ctx->launches += 2;
            VVCHK(vv_pack16_launch(ctx->h, H, 1, L.ln1, c.lm_eps, nullptr, nullptr, 0, ctx->p16_x, R, H, st));
            VVCHK(p16_gemv(ctx, st, L.wqkv, nullptr, ctx->p16_x, ctx->qkv, nullptr, L.bqkv, nullptr, R, QKV, H, QKV, 0, VV_EPI_BIAS));
        } else {
        VVGemm g = mk_gemm(L.wqkv, ctx->h, ctx->qkv, R, QKV, H, H, QKV);
        g.pro = VV_PRO_RMS; g.nw = L.ln1; g.eps = c.lm_eps; g.epi = VV_EPI_BIAS; g.bias = L.bqkv; g.nt = 1;
        g.xa = ctx->h_parts; g.n_xa = hp; g.part_stride = hps;
        GEMM(g);
        }
        char* kl = (char*)ctx->kc + (size_t)l * ctx->layer_stride * 2;
        char* vl = (char*)ctx->vc + (size_t)l * ctx->layer_stride * 2;
        if (fused_attn) {
            // decode rows (one cache each): RoPE + KV append + split attention in ONE launch (+ the merge launch when split)
            ctx->launches += (attn_S > 1) ? 2 : 1;
            if (ctx->prof_on) {
                // algorithmic bytes: every cached position of every row once, K and V (bf16) + the row's q / new k, v / output
                const double by = (double)kv_positions * Hkv * D * 2.0 * 2.0 + (double)R * (QKV + Hq * D) * 4.0;
                const int xs = c.xsplit; vv_ctx* cx = ctx;
                void* opk = (p16 && ctx->p16_fuse) ? ctx->p16_y : nullptr;
                ctx->prof_other.push_back({2, by, [=](hipStream_t s) {
                    return vv_attn_fused_launch(D, xs, cx->qkv, cx->rows_dev, cx->rope_tab, kl, vl, R, Hq, Hkv, cx->cache_stride,
                                                cx->head_stride, attn_S, attn_W, cx->pm, cx->pl, cx->po, cx->attn, opk, s); }});
            }
            // batch decode: the attention (or its merge) writes the o-projection's packed bf16 operand itself
            VVCHK(vv_attn_fused_launch(D, c.xsplit, ctx->qkv, ctx->rows_dev, ctx->rope_tab, kl, vl, R, Hq, Hkv, ctx->cache_stride,
                                       ctx->head_stride, attn_S, attn_W, ctx->pm, ctx->pl, ctx->po, ctx->attn, (p16 && ctx->p16_fuse) ? ctx->p16_y : nullptr, st));
        } else {
            // rows of one launch share caches (prefill chunks): every append must land before any row attends
            ctx->launches += 3;
            VVCHK(vv_rope_append_launch(D, ctx->qkv, ctx->rows_dev, ctx->inv_freq, ctx->qrot, kl, vl,
                                        R, Hq, Hkv, ctx->cache_stride, ctx->head_stride, st));
            if (contiguous && ctx->attn2_ok)      // prompt chunk, bf16 mode: 64 query rows x all heads of the group share every K/V block
                VVCHK(vv_attn_prefill4_launch(D, ctx->qrot, ctx->rows_dev, kl, vl, R, Hq, Hkv, ctx->cache_stride, ctx->head_stride, ctx->attn, nullptr, st));
            else {
                // ragged row sets (the streaming model's text windows) and the prompt chunks of the exact modes (xsplit 2, 3): the
                // split + merge pair, at most ws_rows rows per launch (its partial buffers); every row attends its own causal prefix
                for (int g0 = 0; g0 < R; g0 += ctx->ws_rows) {
                    const int ng = std::min(ctx->ws_rows, R - g0);
                    if (g0) ctx->launches += 2;
                    VVCHK(vv_attn_launch(D, c.xsplit, ctx->qrot + (size_t)g0 * Hq * D, ctx->rows_dev + g0, kl, vl, ng, Hq, Hkv, ctx->cache_stride,
                                         ctx->head_stride, attn_S, ctx->pm, ctx->pl, ctx->po, ctx->attn + (size_t)g0 * Hq * D, st));
                }
            }
        }
        if (p16 && ctx->p16_fuse) {
            ctx->launches += 3;
            // o-projection: h += Wo . attn; its epilogue packs h * ln2 (-> p16_x) and the rows' partial sums of squares (-> ssq_a)
            VVGemv16p ao = p16_args(L.wo, nullptr, ctx->p16_y, ctx->h, ctx->p16_x, R, H, Hq * D, H);
            ao.pk_nw = L.ln2; ao.ssq_out = ctx->ssq_a;
            VVCHK(p16_go(ctx, st, ao, VV_EPI_RESID, 4));
            VVGemv16p ag = p16_args(L.wg, L.wu, ctx->p16_x, nullptr, ctx->p16_act, R, I, H, 0);
            ag.ssq_in = ctx->ssq_a; ag.ssq_tiles = H / 16; ag.eps = c.lm_eps;
            VVCHK(p16_go(ctx, st, ag, VV_EPI_SWIGLU, 1));
            // down projection: h += Wd . act; the next layer's QKV operand (h * its ln1 -> p16_x, ssq_b) unless this is the last layer
            VVGemv16p ad = p16_args(L.wd, nullptr, ctx->p16_act, ctx->h, nullptr, R, H, I, H);
            if (l + 1 < l1) {
                ad.Yp = (unsigned char*)ctx->p16_x; ad.pk_nw = ctx->layers[l + 1].ln1; ad.ssq_out = ctx->ssq_b;
                VVCHK(p16_go(ctx, st, ad, VV_EPI_RESID, 4));
            } else VVCHK(p16_go(ctx, st, ad, VV_EPI_RESID, 0));
            continue;
        }
        if (p16) {
            ctx->launches += 2;
            VVCHK(vv_pack16_launch(ctx->attn, Hq * D, 0, nullptr, 0.f, nullptr, nullptr, 0, ctx->p16_x, R, Hq * D, st));
            VVCHK(p16_gemv(ctx, st, L.wo, nullptr, ctx->p16_x, ctx->h, nullptr, nullptr, nullptr, R, H, Hq * D, H, 0, VV_EPI_RESID));
            ctx->launches += 3;
            VVCHK(vv_pack16_launch(ctx->h, H, 1, L.ln2, c.lm_eps, nullptr, nullptr, 0, ctx->p16_x, R, H, st));
            VVCHK(p16_gemv(ctx, st, L.wg, L.wu, ctx->p16_x, nullptr, ctx->p16_act, nullptr, nullptr, R, I, H, 0, 0, VV_EPI_SWIGLU));
            VVCHK(p16_gemv(ctx, st, L.wd, nullptr, ctx->p16_act, ctx->h, nullptr, nullptr, nullptr, R, H, I, H, 0, VV_EPI_RESID));
            continue;
        }
        VVGemm go = mk_gemm(L.wo, ctx->attn, ctx->h, R, H, Hq * D, Hq * D, H);
        go.epi = VV_EPI_RESID; go.nt = 1;
        go.ya = ctx->h_parts; go.n_ya = hp; go.part_stride = hps;     // o_proj folds the parts back: h is whole again
        GEMM(go);
        hp = 0;
        VVGemm gm = mk_gemm(L.wg, ctx->h, ctx->act, R, I, H, H, I);
        gm.W2 = (const u32x4*)L.wu; gm.pro = VV_PRO_RMS; gm.nw = L.ln2; gm.eps = c.lm_eps; gm.epi = VV_EPI_SWIGLU; gm.nt = 1;
        GEMM(gm);
        VVGemm gd = mk_gemm(L.wd, ctx->act, ctx->h, R, H, I, I, H);
        gd.epi = VV_EPI_RESID; gd.nt = 1;
        if (l + 1 < l1) hp = ksplit_parts(ctx, gd, ctx->h_parts, hps);     // the last layer leaves h whole for the final norm
        GEMM(gd);
    }
    ctx->launches++;
    if (final_norm) VVCHK(vv_rmsnorm_rows_launch(ctx->h, H, hidden_out, H, ctx->lm_norm, R, H, c.lm_eps, st));
    else VVCHK(vv_copy_launch(hidden_out, ctx->h, (size_t)R * H * 4, st));
    return 0;
}

extern "C" int vv_lm_forward_range(vv_ctx* ctx, void* stream, int n_rows, const vv_row* rows, const float* x_in_dev,
                                   float* hidden_out_dev, int l0, int l1, int final_norm);
// The prefill GEMM's K-split hand-off (prefill.hip g4_finish) reports a lost producer through a host-mapped word instead of
// hanging the GPU; the affected tile is wrong (summed from incomplete partials) and the arrival words are left untouched.  Recovery, done here at the next
// enqueue / vv_check: wait for the stream (nothing of that launch is in flight any more), re-zero the arrival words, clear the
// word and fail THIS call -- the caller knows the output of the prompt pass in flight is invalid and can retry; the context
// stays usable.
static int ksplit_check(vv_ctx* ctx, hipStream_t st) {
    if (!(ctx->gws.err && *ctx->gws.err)) return 0;
    {   // stream-level waits only, under the lock captures take exclusively: a device-wide synchronize (or a null-stream memset) here
        // would invalidate a capture ANOTHER context of the process has open (lanes: vv_create_shared) -- exactly when one lane
        // recovers from a timed-out hand-off while the other keeps decoding.  The arrival words belong to this context; the launches
        // that touch them run on st (the only stream a prompt pass is enqueued on).
        VV_SHARED;
        hipStreamSynchronize(st);
        if (ctx->gws.flags) { hipMemsetAsync(ctx->gws.flags, 0, 256 * sizeof(unsigned), st); hipStreamSynchronize(st); }
    }
    *ctx->gws.err = 0u;
    return fail(ctx, "prefill GEMM: a K-split hand-off timed out (lost producer workgroup); the prompt pass that was in flight is invalid -- "
                     "the arrival words were re-armed, retry the pass (VVHIP_NO_KSPLIT=1 disables the split)");
}
extern "C" int vv_check(vv_ctx* ctx, void* stream) {
    if (!ctx) return -1;
    return ksplit_check(ctx, (hipStream_t)stream);
}
extern "C" int vv_lm_forward(vv_ctx* ctx, void* stream, int n_rows, const vv_row* rows, const float* x_in_dev, float* hidden_out_dev) {
    return vv_lm_forward_range(ctx, stream, n_rows, rows, x_in_dev, hidden_out_dev, 0, ctx->c.lm_layers, 1);
}
extern "C" int vv_lm_forward_range(vv_ctx* ctx, void* stream, int n_rows, const vv_row* rows, const float* x_in_dev,
                                   float* hidden_out_dev, int l0, int l1, int final_norm) {
    hipStream_t st = (hipStream_t)stream;
    if (l0 < 0 || l1 > ctx->c.lm_layers || l0 >= l1) return fail(ctx, "layer range [%d,%d) invalid", l0, l1);
    if (n_rows < 1 || n_rows > ctx->c.max_rows) return fail(ctx, "n_rows %d out of range [1,%d]", n_rows, ctx->c.max_rows);
    if (ksplit_check(ctx, st)) return -1;
    (void)hipGetLastError();            // a stale error of this host thread (another library's query) is not a launch failure of ours
    for (int i = 0; i < n_rows; ++i) {
        if (rows[i].cache < 0 || rows[i].cache >= 2 * ctx->c.n_slots) return fail(ctx, "row %d: cache id %d out of range", i, rows[i].cache);
        if (rows[i].pos < 0 || rows[i].pos >= ctx->c.max_ctx) return fail(ctx, "row %d: position %d exceeds max_ctx %d", i, rows[i].pos, ctx->c.max_ctx);
    }
    const int slot = ring_acquire(ctx);
    VVRow* pin = ctx->rows_pin + (size_t)slot * ctx->rows_cap;
    for (int i = 0; i < n_rows; ++i) { pin[i].cache = rows[i].cache; pin[i].pos = rows[i].pos; }
    HIPCHK(ctx, hipMemcpyAsync(ctx->rows_dev, pin, sizeof(VVRow) * n_rows, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipEventRecord(ctx->ring_ev[slot], st));
    ctx->launches = 0;
    bool fused = true;
    for (int i = 0; i < n_rows && fused; ++i)
        for (int j = 0; j < i; ++j) if (rows[i].cache == rows[j].cache) { fused = false; break; }
    if (!ctx->rope_ready) {                   // (cos, sin) table of every position, once the inv_freq parameter is in place
        VVCHK(vv_rope_table_launch(ctx->inv_freq, ctx->rope_tab, ctx->c.max_ctx, ctx->D / 2, st));
        ctx->rope_ready = true;
    }
    bool contiguous = !fused && n_rows >= 8;      // one cache, consecutive positions
    for (int i = 1; i < n_rows && contiguous; ++i)
        if (rows[i].cache != rows[0].cache || rows[i].pos != rows[0].pos + i) contiguous = false;
    if (!contiguous && n_rows > ctx->ws_rows)
        return fail(ctx, "a launch of %d rows must be consecutive positions of one cache (decode / ragged launches take <= %d rows)", n_rows, ctx->ws_rows);
    // decode attention geometry: one split (workgroup column) per 1024 positions of the longest row, at most attn_splits; the
    // 8-wave form once the KV stream dominates.  Both are grid / template choices, so they are part of the graph key: a
    // growing context re-captures the step graph every 512 positions.
    int max_len = 1;
    for (int i = 0; i < n_rows; ++i) max_len = std::max(max_len, rows[i].pos + 1);
    // ... and no more splits than it takes to put ~256 workgroups on the chip: with eight 32K-context utterances in flight the
    // rows themselves are the parallelism (8 splits of 4096 positions: 122 us per layer against 162 us with 32 splits).
    // Measured and left alone: 512 / 256 positions per split (no gain once the merge is its own launch), 8-wave workgroups.
    constexpr int split_pos = 1024;
    static int target_wgs = -1;            // workgroups a launch of long rows aims for (VVHIP_ATTN_TARGET_WGS: A/B of round 6)
    if (target_wgs < 0) { const char* e = getenv("VVHIP_ATTN_TARGET_WGS"); target_wgs = e ? std::max(64, atoi(e)) : 256; }
    int n_long = 0;
    for (int i = 0; i < n_rows; ++i) if (rows[i].pos + 1 > split_pos) ++n_long;
    const int by_wgs = std::max(1, (target_wgs + std::max(1, n_long) * ctx->Hkv - 1) / (std::max(1, n_long) * ctx->Hkv));
    const int attn_S = std::min(std::min(ctx->c.attn_splits, by_wgs), std::max(1, (max_len + split_pos - 1) / split_pos));
    // one split, but several 32-position blocks per wave: the 8-wave form (all K/V requests of a <= 512-position context in
    // flight at once).  Measured three times now, the third with every wave combining its share of the output tiles: slower than
    // 4 waves (round 4: 8.90 vs 8.28 us per unit at 400 positions, 1.5B; 5.84 vs 5.78 at 250, 0.5B) -- off unless
    // VVHIP_ATTN_W8_MIN = positions from which to use it
    static int w8_min = -1;
    if (w8_min < 0) { const char* e = getenv("VVHIP_ATTN_W8_MIN"); w8_min = e ? atoi(e) : 0; }
    const int attn_W = (fused && attn_S == 1 && w8_min > 0 && max_len >= w8_min) ? 8 : 4;
    char key[160]; snprintf(key, 160, "lm:%d:%p:%p:%d:%d:%d:%d:%d", n_rows, (const void*)x_in_dev, (void*)hidden_out_dev, l0, l1, final_norm,
                            fused ? 1 : (contiguous ? 2 : 0), (contiguous && ctx->attn2_ok) ? 0 : attn_S * 16 + attn_W);
    int64_t kv_positions = 0;
    for (int i = 0; i < n_rows; ++i) kv_positions += rows[i].pos + 1;
    return graphed(ctx, key, st, [&]() { return lm_body(ctx, st, n_rows, x_in_dev, hidden_out_dev, l0, l1, final_norm, fused, contiguous, attn_S, kv_positions, attn_W); });
}

extern "C" int vv_kv_import_at(vv_ctx* ctx, void* stream, int cache, int layer, int pos0, int n_pos, const void* k_dev, const void* v_dev, int src_dtype) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (cache < 0 || cache >= 2 * ctx->c.n_slots) return fail(ctx, "cache id %d out of range", cache);
    if (layer < 0 || layer >= ctx->c.lm_layers) return fail(ctx, "layer %d out of range", layer);
    if (pos0 < 0 || n_pos < 0 || (int64_t)pos0 + n_pos > ctx->c.max_ctx) return fail(ctx, "positions [%d, %d) exceed max_ctx %d", pos0, pos0 + n_pos, ctx->c.max_ctx);
    if (n_pos == 0) return 0;
    const size_t off = ((size_t)cache * ctx->cache_stride + (size_t)layer * ctx->layer_stride) * 2;
    VVCHK(vv_kv_import_launch(k_dev, v_dev, src_dtype, (char*)ctx->kc + off, (char*)ctx->vc + off, n_pos, ctx->Hkv, ctx->D, ctx->head_stride, pos0, st));
    return 0;
}
extern "C" int vv_kv_move(vv_ctx* ctx, void* stream, int cache, int src_pos, int dst_pos) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (cache < 0 || cache >= 2 * ctx->c.n_slots) return fail(ctx, "cache id %d out of range", cache);
    if (src_pos < 0 || dst_pos < 0 || src_pos >= ctx->c.max_ctx || dst_pos >= ctx->c.max_ctx)
        return fail(ctx, "vv_kv_move: positions %d -> %d outside [0, %d)", src_pos, dst_pos, ctx->c.max_ctx);
    if (src_pos == dst_pos) return 0;
    const size_t off = (size_t)cache * ctx->cache_stride * 2;
    ctx->launches++;
    VVCHK(vv_kv_move_launch((char*)ctx->kc + off, (char*)ctx->vc + off, ctx->c.lm_layers, ctx->Hkv, ctx->D, ctx->layer_stride, ctx->head_stride,
                            src_pos, dst_pos, st));
    return 0;
}
extern "C" int vv_kv_import(vv_ctx* ctx, void* stream, int cache, int layer, int n_pos, const void* k_dev, const void* v_dev, int src_dtype) {
    return vv_kv_import_at(ctx, stream, cache, layer, 0, n_pos, k_dev, v_dev, src_dtype);
}

extern "C" int vv_audio_to_pcm16(vv_ctx* ctx, void* stream, int n, int samples, const float* audio_dev, int16_t* pcm_out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (n < 1 || samples < 1) return fail(ctx, "vv_audio_to_pcm16: n and samples must be positive");
    VVCHK(vv_pcm16_launch(audio_dev, (short*)pcm_out_dev, n, samples, st));
    return 0;
}

extern "C" int vv_add_type_embedding(vv_ctx* ctx, void* stream, int n, const float* x_dev, int type, float* out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->tts_types) return fail(ctx, "engine was not configured with tts_layers");
    if (type < 0 || type > 1) return fail(ctx, "type must be 0 (speech) or 1 (text)");
    VVCHK(vv_add_rows_launch(x_dev, ctx->tts_types + (size_t)type * ctx->H, out_dev, n, ctx->H, st));
    return 0;
}

extern "C" int vv_eos_logit(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->eos_w1) return fail(ctx, "engine was not configured with tts_layers");
    if (n < 1 || n > 16) return fail(ctx, "vv_eos_logit: n must be in [1,16]");
    const int H = ctx->H;
    VVGemm g1 = mk_gemm(ctx->eos_w1, hidden_dev, ctx->ct1, n, H, H, H, H);
    g1.epi = VV_EPI_BIAS; g1.bias = ctx->eos_b1; GEMM(g1);
    VVCHK(vv_relu_launch(ctx->ct1, n * H, st));
    VVGemm g2 = mk_gemm(ctx->eos_w2, ctx->ct1, out_dev, n, 1, H, H, 1);
    g2.epi = VV_EPI_BIAS; g2.bias = ctx->eos_b2; GEMM(g2);
    return 0;
}

extern "C" int vv_embed(vv_ctx* ctx, void* stream, int n, const int* ids, float* out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (n < 1 || n > ctx->ids_cap) return fail(ctx, "vv_embed: n must be in [1,%d] (max(64, max_rows))", ctx->ids_cap);
    for (int i = 0; i < n; ++i) if (ids[i] < 0 || ids[i] >= ctx->c.lm_vocab) return fail(ctx, "token id %d out of range", ids[i]);
    const int slot = ring_acquire(ctx);
    int* pin = ctx->ids_pin + (size_t)slot * ctx->ids_cap;
    memcpy(pin, ids, sizeof(int) * n);
    HIPCHK(ctx, hipMemcpyAsync(ctx->ids_dev, pin, sizeof(int) * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipEventRecord(ctx->ring_ev[slot], st));
    VVCHK(vv_embed_launch(ctx->embed, ctx->ids_dev, out_dev, n, ctx->H, st));
    return 0;
}

extern "C" int vv_lm_logits_full(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* logits_out_dev) {
    VV_SHARED;
    const void* table = ctx->lm_head_loaded ? ctx->lm_head : ctx->embed;
    if (!table) return fail(ctx, "vv_lm_logits_full: no lm_head / embedding table has been uploaded");
    if (n < 1 || n > 16) return fail(ctx, "vv_lm_logits_full: n must be in [1,16]");
    if (ctx->H & 7) return fail(ctx, "vv_lm_logits_full: hidden size %d is not a multiple of 8", ctx->H);
    VVCHK(vv_logits_full_launch(table, hidden_dev, logits_out_dev, n, ctx->c.lm_vocab, ctx->H, (hipStream_t)stream));
    return 0;
}
extern "C" int vv_lm_logits(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* logits_out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->valid_w) return fail(ctx, "vv_set_valid_tokens has not been called");
    if (n < 1 || n > 16) return fail(ctx, "vv_lm_logits: n must be in [1,16]");
    VVGemm g = mk_gemm(ctx->valid_w, hidden_dev, logits_out_dev, n, ctx->n_valid, ctx->H, ctx->H, ctx->n_valid);
    GEMM(g);
    return 0;
}

// one head evaluation on 2n rows; mod/xh/hact/eps are ctx scratch. temb = t-embedding row for this step.
static int head_eval(vv_ctx* ctx, hipStream_t st, int rows, const float* zrows, const float* temb_row, float* eps_out,
                     const float* coef = nullptr, float cfg = 0.f, const float* mod_ready = nullptr, const float* sde_noise = nullptr,
                     const unsigned char* sh_tiles = nullptr, int gen = 0, bool have_x = false, bool seam = false) {
    // gen / have_x / seam (sampler, decode rows, bf16 mode): the step's state is generation `gen` (xh / zz / x0p or their second copies);
    // have_x: the previous step's seam launch already produced this step's in-projection; seam: end this step with the fused launch
    // (final layer + CFG + solver update + the NEXT step's in-projection, written to the other generation) instead of the final layer
    const vv_config& c = ctx->c;
    const int H = ctx->H, L = c.latent_dim, HL = c.head_layers, HF = ctx->HF, MODW = ctx->MODW;
    const float* mod = mod_ready ? mod_ready : ctx->mod;
    float* const xh = gen ? ctx->xh2 : ctx->xh;
    float* const zcur = gen ? ctx->zz2 : ctx->zz;
    float* const x0cur = gen ? ctx->x0p2 : ctx->x0p;
    if (!mod_ready) {
        VVGemm ga = mk_gemm(ctx->h_ada, ctx->cproj, ctx->mod, rows, MODW, H, H, MODW);
        ga.pro = VV_PRO_ADD_SILU; ga.addvec = temb_row; ga.nt = 1;
        GEMM(ga);
    }
    if (!have_x) {
        VVGemm gi = mk_gemm(ctx->h_in, zrows, xh, rows, H, L, L, H);
        GEMM(gi);
        nan_probe(ctx, st, "in-proj xh", xh, (size_t)rows * H);
    }
    int xp = 0;                                    // extra parts xh currently consists of
    const int xps = 16 * H;
    for (int l = 0; l < HL; ++l) {
        const float* base = mod + (size_t)l * 3 * H;
        if (rows > 4 && rows <= 16 && ctx->p16_ok && (HF % 32) == 0 && ctx->p16_fuse && sh_tiles && coef && ctx->p16_head_sh) {
            // batch rows, round 6: layer 0 packs its operand (the in-projection is not a packed-activation launch); every later layer
            // finds x * w * (1 + scale) packed by the previous down projection's epilogue, the rows' sums of squares beside it, and the
            // shift rows of this (step, layer) packed once per frame: y = rs * W.xm + W.shift
            if (l == 0) {
                ctx->launches += 1;
                VVCHK(vv_pack16_launch(xh, H, 2, ctx->hl[l].norm, c.head_eps, base + H, base, MODW, ctx->p16_x, rows, H, st));
                VVCHK(p16_gemv(ctx, st, ctx->hl[l].wg, ctx->hl[l].wu, ctx->p16_x, nullptr, ctx->p16_act, nullptr, nullptr, rows, HF, H, 0, 0, VV_EPI_SWIGLU));
            } else {
                VVGemv16p ag = p16_args(ctx->hl[l].wg, ctx->hl[l].wu, ctx->p16_x, nullptr, ctx->p16_act, rows, HF, H, 0);
                ag.ssq_in = ctx->ssq_a; ag.ssq_tiles = H / 16; ag.eps = c.head_eps;
                ag.Xs = (const u32x4*)(sh_tiles + (size_t)l * ctx->p16_shift_tile);
                VVCHK(p16_go(ctx, st, ag, VV_EPI_SWIGLU, 3));
            }
            ctx->launches += 2;
            VVGemv16p ad = p16_args(ctx->hl[l].wd, nullptr, ctx->p16_act, xh, ctx->p16_x, rows, H, HF, H);
            ad.gate = base + 2 * H; ad.ld_gate = MODW; ad.ssq_out = ctx->ssq_a;
            ad.pk_nw = (l + 1 < HL) ? ctx->hl[l + 1].norm : nullptr;
            ad.pk_sc = mod + (size_t)(l + 1) * 3 * H + H; ad.ld_pk = MODW;          // layer l + 1's scale rows (l + 1 == HL: the final layer's)
            VVCHK(p16_go(ctx, st, ad, VV_EPI_GATED_RESID, 4));
            continue;
        }
        if (rows > 4 && rows <= 16 && ctx->p16_ok && (HF % 32) == 0) {
            // batch rows: normalise + modulate + pack ONCE, then both projections stream weights against packed fragments
            ctx->launches += 3;
            VVCHK(vv_pack16_launch(xh, H, 2, ctx->hl[l].norm, c.head_eps, base + H, base, MODW, ctx->p16_x, rows, H, st));
            VVCHK(p16_gemv(ctx, st, ctx->hl[l].wg, ctx->hl[l].wu, ctx->p16_x, nullptr, ctx->p16_act, nullptr, nullptr, rows, HF, H, 0, 0, VV_EPI_SWIGLU));
            if (l + 1 == HL && ctx->p16_fuse && sh_tiles && coef) {
                // the last layer's down projection leaves the FINAL layer's operand (x * (1 + scale), un-normalised) packed and the rows' sums of squares
                VVGemv16p ad = p16_args(ctx->hl[l].wd, nullptr, ctx->p16_act, xh, ctx->p16_x, rows, H, HF, H);
                ad.gate = base + 2 * H; ad.ld_gate = MODW; ad.ssq_out = ctx->ssq_a;
                ad.pk_nw = nullptr; ad.pk_sc = mod + (size_t)HL * 3 * H + H; ad.ld_pk = MODW;
                VVCHK(p16_go(ctx, st, ad, VV_EPI_GATED_RESID, 4));
            } else
            VVCHK(p16_gemv(ctx, st, ctx->hl[l].wd, nullptr, ctx->p16_act, xh, nullptr, nullptr, base + 2 * H, rows, H, HF, H, MODW, VV_EPI_GATED_RESID));
            continue;
        }
        VVGemm g1 = mk_gemm(ctx->hl[l].wg, xh, ctx->hact, rows, HF, H, H, HF);
        g1.W2 = (const u32x4*)ctx->hl[l].wu; g1.pro = VV_PRO_RMS_MOD; g1.nw = ctx->hl[l].norm; g1.eps = c.head_eps;
        g1.mod_shift = base; g1.mod_scale = base + H; g1.ld_mod = MODW; g1.epi = VV_EPI_SWIGLU; g1.nt = 1;
        float* cur = ctx->xh_parts + (size_t)(l & 1) * 2 * xps;          // parts written by layer l-1
        float* nxt = ctx->xh_parts + (size_t)((l + 1) & 1) * 2 * xps;    // parts layer l writes
        g1.xa = cur; g1.n_xa = xp; g1.part_stride = xps;
        GEMM(g1);
        VVGemm g2 = mk_gemm(ctx->hl[l].wd, ctx->hact, xh, rows, H, HF, HF, H);
        g2.epi = VV_EPI_GATED_RESID; g2.gate = base + 2 * H; g2.ld_gate = MODW; g2.nt = 1;
        g2.ya = cur; g2.n_ya = xp; g2.part_stride = xps;
        xp = ksplit_parts(ctx, g2, nxt, xps);
        if (ctx->probe_on == 1) { char nm[64]; snprintf(nm, 64, "layer %d hact (parts in %d)", l, g1.n_xa); nan_probe(ctx, st, nm, ctx->hact, (size_t)rows * HF); }
        GEMM(g2);
        if (ctx->probe_on == 1) {
            char nm[64]; snprintf(nm, 64, "layer %d xh", l); nan_probe(ctx, st, nm, xh, (size_t)rows * H);
            for (int q = 0; q < xp; ++q) { snprintf(nm, 64, "layer %d part %d", l, q); nan_probe(ctx, st, nm, nxt + (size_t)q * xps, (size_t)rows * H); }
        }
    }
    const float* fb = mod + (size_t)HL * 3 * H;
    if (rows > 4 && rows <= 16 && ctx->p16_ok && (HF % 32) == 0 && ctx->p16_fuse && sh_tiles && coef && HL > 0) {
        // the sampler's final layer over the packed operand the last down projection left (4 workgroups that only stream: the 16-row
        // vv_gemv form staged 16 x H modulated rows in each of its 4 workgroups, 21 us), CFG + DPM-Solver++ update in the epilogue
        ctx->launches += 1;
        VVGemv16p af = p16_args(ctx->h_out, nullptr, ctx->p16_x, nullptr, nullptr, rows, L, H, L);
        af.ssq_in = ctx->ssq_a; af.ssq_tiles = H / 16; af.eps = c.head_eps;
        af.Xs = (const u32x4*)(sh_tiles + (size_t)HL * ctx->p16_shift_tile);
        af.z = zcur; af.x0p = x0cur; af.coef = coef; af.cfg = cfg; af.n_cfg = rows / 2; af.sde_noise = sde_noise;
        VVCHK(p16_go(ctx, st, af, VV_EPI_CFG_DPM, 3));
        return 0;
    }
    if (seam && coef && rows == 2 && ctx->head_tail_tpw > 0) {
        VVTail t{};
        t.Wout = (const u32x4*)ctx->h_out; t.Win = (const u32x4*)ctx->h_in; t.bin = nullptr;
        t.X = xh; t.xa = ctx->xh_parts + (size_t)(HL & 1) * 2 * xps; t.n_xa = xp; t.part_stride = xps;
        t.sc = fb + H; t.sh = fb; t.ld_mod = MODW;
        t.Xout = gen ? ctx->xh : ctx->xh2;
        t.z_in = zcur; t.x0p_in = x0cur; t.z_out = gen ? ctx->zz : ctx->zz2; t.x0p_out = gen ? ctx->x0p : ctx->x0p2;
        t.coef = coef; t.cfg = cfg; t.n_cfg = rows / 2; t.sde_noise = sde_noise;
        t.T = rows; t.H = H; t.L = L; t.eps = c.head_eps;
        if (vv_head_tail_ok(&t)) {
            ctx->launches++;
            if (ctx->prof_on) {
                const VVTail tc = t; const int tpw = ctx->head_tail_tpw;
                ctx->prof_other.push_back({3, (double)vv_packed_elems(L, H) * 2.0 + (double)vv_packed_elems(H, L) * 2.0 + (double)rows * H * 8.0,
                                           [=](hipStream_t s2) { return vv_head_tail_launch(&tc, tpw, s2); }});
            }
            VVCHK(vv_head_tail_launch(&t, ctx->head_tail_tpw, st));
            return 1;          // the next step's in-projection is done (generation gen ^ 1)
        }
    }
    VVGemm gf = mk_gemm(ctx->h_out, xh, eps_out, rows, L, H, H, L);
    gf.pro = VV_PRO_RMS_MOD; gf.nw = nullptr; gf.eps = c.head_eps; gf.mod_shift = fb; gf.mod_scale = fb + H; gf.ld_mod = MODW;
    gf.xa = ctx->xh_parts + (size_t)(HL & 1) * 2 * xps; gf.n_xa = xp; gf.part_stride = xps;
    if (coef) {   // CFG + DPM-Solver++ update fused into the epilogue: the noisy latent is rewritten in place
        gf.epi = VV_EPI_CFG_DPM; gf.z = zcur; gf.x0p = x0cur; gf.coef = coef; gf.cfg = cfg; gf.n_cfg = rows / 2;
        gf.sde_noise = sde_noise;
    }
    GEMM(gf);
    return 0;
}

static int sample_body(vv_ctx* ctx, hipStream_t st, int n, const float* cond, const float* noise, float cfg, float* latent_out,
                       const float* step_noise = nullptr) {
    const vv_config& c = ctx->c;
    const int H = ctx->H, L = c.latent_dim;
    const int rows = 2 * n;
    ctx->probe_names.clear();
    nan_probe(ctx, st, "cond (input)", cond, (size_t)rows * H);
    nan_probe(ctx, st, "noise (input)", noise, (size_t)n * L);
    // both CFG halves see the same noisy latent (modeling_vibevoice_inference.py:703-704)
    VVCHK(vv_sampler_init_launch(noise, ctx->zz, ctx->x0p, n * L, st));
    VVGemm gc = mk_gemm(ctx->h_cond, cond, ctx->cproj, rows, H, H, H, H);
    gc.nt = 1;
    GEMM(gc);
    // adaLN modulations depend on (cond, t) only, not on the evolving latent: evaluate them for ALL solver steps
    // up front, <=16 rows per GEMM, so the (3*layers+2)*H x H modulation matrix is streamed ceil(2nN/16) times per
    // frame instead of N times (the reference recomputes it inside every head call)
    const int MODW = ctx->MODW;
    const bool batch_ada = ctx->mod_all_bytes != 0;
    if (batch_ada) {
        // SiLU(cond + t) for all (step, row) pairs in one small launch: the GEMM workgroups (one per 16 output features,
        // > 1000 of them) then stage plain rows instead of each re-evaluating 16 x H SiLUs
        const int total = rows * ctx->n_steps;
        // bf16 mode, three or more 16-row passes: ONE MFMA tile GEMM over all (step, row) pairs instead -- the modulation
        // matrix (360 MB for the 7B head) is streamed once, not once per 16 rows (8 utterances x 20 steps: 20 passes)
        if (ctx->ada_p && total > 32 && (MODW & 3) == 0) {
            ctx->launches += 2;
            VVCHK(vv_ada_pack_launch(ctx->cproj, ctx->temb, ctx->ada_p, rows, ctx->n_steps, H, st));
            VVCHK(vv_gemm3_launch(ctx->h_ada, nullptr, ctx->ada_p, ctx->mod_all, nullptr, nullptr, total, MODW, H, MODW, VV_EPI_STORE, nullptr, st));
        } else {
        VVCHK(vv_ada_in_launch(ctx->cproj, ctx->temb, ctx->ada_in, rows, ctx->n_steps, H, st));
        ctx->launches++;
        for (int t0 = 0; t0 < total; t0 += 16) {
            const int T = std::min(16, total - t0);
            VVGemm ga = mk_gemm(ctx->h_ada, ctx->ada_in + (size_t)t0 * H, ctx->mod_all + (size_t)t0 * MODW, T, MODW, H, H, MODW);
            GEMM(ga);
        }
        }
    }
    nan_probe(ctx, st, "cproj", ctx->cproj, (size_t)rows * H);
    if (batch_ada) nan_probe(ctx, st, "mod_all", ctx->mod_all, (size_t)rows * ctx->n_steps * MODW);
    const bool sh_ok = batch_ada && rows > 4 && rows <= 16 && ctx->p16_fuse && ctx->p16_shift;
    if (sh_ok) {
        // the shift rows of every (solver step, layer) -- and the final layer's -- as packed bf16 tiles, one launch per frame:
        // tile (i, l) = rows [i * rows, (i + 1) * rows) of mod_all, columns [l * 3H, l * 3H + H)
        ctx->launches++;
        VVCHK(vv_pack16_tiles_launch(ctx->mod_all, MODW, (int64_t)rows * MODW, ctx->c.head_layers + 1, (int64_t)3 * H, ctx->p16_shift,
                                     (int64_t)ctx->p16_shift_tile, rows, H, ctx->n_steps * (ctx->c.head_layers + 1), st));
    }
    int gen = 0; bool have_x = false;
    for (int i = 0; i < ctx->n_steps; ++i) {
        const float* mod_i = batch_ada ? ctx->mod_all + (size_t)i * rows * MODW : nullptr;
        const float* sn = step_noise ? step_noise + (size_t)i * n * L : nullptr;
        const unsigned char* sht = sh_ok ? (const unsigned char*)ctx->p16_shift + (size_t)i * (ctx->c.head_layers + 1) * ctx->p16_shift_tile : nullptr;
        // decode rows, bf16 mode: every step but the last ends with the fused seam (final layer + CFG + solver update + the next step's
        // in-projection, headtail.hip), which leaves the next step's state in the other generation of (xh, zz, x0p)
        const bool seam = (i + 1 < ctx->n_steps) && rows == 2 && ctx->head_tail_tpw > 0;
        const int hr = head_eval(ctx, st, rows, gen ? ctx->zz2 : ctx->zz, ctx->temb + (size_t)i * H, ctx->eps, ctx->coef + i * 6, cfg, mod_i, sn, sht,
                                 gen, have_x, seam);
        if (hr < 0) return -1;
        if (ctx->probe_on == 1) {
            char nm[64];
            snprintf(nm, 64, "step %d z%s", i, hr == 1 ? " (seam, next gen)" : ""); nan_probe(ctx, st, nm, (gen ^ (hr == 1)) ? ctx->zz2 : ctx->zz, (size_t)rows * L);
            snprintf(nm, 64, "step %d x0p", i); nan_probe(ctx, st, nm, (gen ^ (hr == 1)) ? ctx->x0p2 : ctx->x0p, (size_t)n * L);
            if (hr == 1) { snprintf(nm, 64, "step %d next xh", i); nan_probe(ctx, st, nm, (gen ^ 1) ? ctx->xh2 : ctx->xh, (size_t)rows * H); }
        }
        have_x = (hr == 1);
        if (have_x) gen ^= 1;
    }
    VVCHK(vv_copy_launch(latent_out, gen ? ctx->zz2 : ctx->zz, (size_t)n * L * 4, st));
    return 0;
}

extern "C" int vv_diffusion_sample(vv_ctx* ctx, void* stream, int n, const float* cond_dev, const float* noise_dev, float cfg_scale, float* latent_out_dev) {
    hipStream_t st = (hipStream_t)stream;
    if (ctx->n_steps < 1) return fail(ctx, "vv_set_schedule has not been called");
    if (n < 1 || n > 8) return fail(ctx, "vv_diffusion_sample: n must be in [1,8]");
    if (ctx->sde_on) return fail(ctx, "the schedule is stochastic (vv_set_schedule_sde): sample with vv_diffusion_sample_sde and its per-step noise");
    ctx->launches = 0;
    char key[128]; snprintf(key, 128, "samp:%d:%p:%p:%p:%a", n, (const void*)cond_dev, (const void*)noise_dev, (void*)latent_out_dev, cfg_scale);
    const int rc = graphed(ctx, key, st, [&]() { return sample_body(ctx, st, n, cond_dev, noise_dev, cfg_scale, latent_out_dev); });
    nan_probe_report(ctx, st, key);
    return rc;
}

// The stochastic solver: step_noise_dev = [n_steps][n][latent_dim] fp32, the variance noise scheduler.step() draws per solver step
// (dpm_solver.py:994-997; the reference draws [2n][latent] and only the first n rows reach the next step, :703-704).
extern "C" int vv_diffusion_sample_sde(vv_ctx* ctx, void* stream, int n, const float* cond_dev, const float* noise_dev,
                                       const float* step_noise_dev, float cfg_scale, float* latent_out_dev) {
    hipStream_t st = (hipStream_t)stream;
    if (ctx->n_steps < 1) return fail(ctx, "vv_set_schedule_sde has not been called");
    if (n < 1 || n > 8) return fail(ctx, "vv_diffusion_sample_sde: n must be in [1,8]");
    if (!ctx->sde_on) return fail(ctx, "the schedule is deterministic (vv_set_schedule): sample with vv_diffusion_sample");
    if (!step_noise_dev) return fail(ctx, "vv_diffusion_sample_sde: step_noise is null");
    ctx->launches = 0;
    char key[160]; snprintf(key, 160, "sde:%d:%p:%p:%p:%p:%a", n, (const void*)cond_dev, (const void*)noise_dev, (const void*)step_noise_dev,
                            (void*)latent_out_dev, cfg_scale);
    return graphed(ctx, key, st, [&]() { return sample_body(ctx, st, n, cond_dev, noise_dev, cfg_scale, latent_out_dev, step_noise_dev); });
}

extern "C" int vv_head_forward(vv_ctx* ctx, void* stream, int n, const float* noisy_dev, const float* t_host, const float* cond_dev, float* out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (n < 1 || n > 16) return fail(ctx, "vv_head_forward: n must be in [1,16]");
    const int H = ctx->H;
    for (int i = 1; i < n; ++i) if (t_host[i] != t_host[0]) return fail(ctx, "vv_head_forward: all rows must share one timestep");
    float* tdev = ctx->tmp2 + 63 * 256;     // scratch
    HIPCHK(ctx, hipStreamSynchronize(st));
    HIPCHK(ctx, hipMemcpy(tdev + 128, t_host, 4, hipMemcpyHostToDevice));
    VVCHK(vv_tfreq_launch(tdev + 128, ctx->tmp2, 1, st));
    VVGemm g = mk_gemm(ctx->h_t0, ctx->tmp2, ctx->tmp1, 1, H, 256, 256, H); GEMM(g);
    VVCHK(vv_silu_launch(ctx->tmp1, H, st));
    VVGemm g2 = mk_gemm(ctx->h_t2, ctx->tmp1, ctx->tmp1 + H, 1, H, H, H, H); GEMM(g2);
    VVGemm gc = mk_gemm(ctx->h_cond, cond_dev, ctx->cproj, n, H, H, H, H); GEMM(gc);
    if (head_eval(ctx, st, n, noisy_dev, ctx->tmp1 + H, out_dev)) return -1;
    HIPCHK(ctx, hipStreamSynchronize(st));
    return 0;
}

extern "C" int vv_codec_decode(vv_ctx* ctx, void* stream, int slot, int frames, const float* latent_dev, float* audio_out_dev, int apply) {
    hipStream_t st = (hipStream_t)stream;
    if (slot < 0 || slot >= ctx->c.n_slots) return fail(ctx, "slot %d out of range", slot);
    if (frames != 1) return fail(ctx, "vv_codec_decode: streaming decode takes one frame per call");
    CodecNet& net = ctx->dec;
    ctx->launches = 0;
    char key[128]; snprintf(key, 128, "dec:%d:%d:%p:%p:%d", slot, frames, (const void*)latent_dev, (void*)audio_out_dev, apply);
    return graphed(ctx, key, st, [&]() {
        const int L = ctx->c.latent_dim;
        const float mul = apply ? 1.0f / ctx->scaling : 1.0f, add = apply ? -ctx->bias : 0.0f;
        ctx->launches++;
        VVCHK(vv_affine_launch(latent_dev, net.in_buf[slot] + 6 * L, mul, add, frames * L, st));
        return run_codec(ctx, net, slot, frames, audio_out_dev, st);
    });
}

extern "C" int vv_semantic_encode(vv_ctx* ctx, void* stream, int slot, int frames, const float* audio_dev, float* sem_out_dev) {
    hipStream_t st = (hipStream_t)stream;
    if (ctx->c.sem_dim <= 0) return fail(ctx, "no semantic tokenizer configured");
    if (slot < 0 || slot >= ctx->c.n_slots) return fail(ctx, "slot %d out of range", slot);
    if (frames != 1) return fail(ctx, "vv_semantic_encode: streaming encode takes one frame per call");
    CodecNet& net = ctx->senc;
    ctx->launches = 0;
    char key[128]; snprintf(key, 128, "senc:%d:%d:%p:%p", slot, frames, (const void*)audio_dev, (void*)sem_out_dev);
    return graphed(ctx, key, st, [&]() {
        VVCHK(vv_copy_launch(net.in_buf[slot] + 6, audio_dev, (size_t)frames * ctx->hop * 4, st));
        return run_codec(ctx, net, slot, frames, sem_out_dev, st);
    });
}

// One frame of n utterances through the acoustic decoder and (sem_out_dev != null) the semantic encoder -- the batched
// `acoustic_tokenizer.decode(..., sample_indices=diffusion_indices)` + `semantic_tokenizer.encode(...)` pair of the reference's
// loop (modeling_vibevoice_inference.py:636-672).  Row j of latent / audio / sem belongs to slot slots[j].  The stages that
// hold the weight bytes (decoder stages [0, kd), encoder stages [ke, end) + head) run slot-batched: one pass over the weights
// for the whole batch; the many-row, few-channel stages in between run per utterance on forked streams.
extern "C" int vv_codec_chain_batch(vv_ctx* ctx, void* stream, int n, const int* slots, const float* latent_dev,
                                    float* audio_out_dev, float* sem_out_dev, int apply) {
    hipStream_t st = (hipStream_t)stream;
    if (n < 1 || n > 8) return fail(ctx, "vv_codec_chain_batch: n = %d, must be 1..8", n);
    uint64_t mask = 0;
    for (int j = 0; j < n; ++j) {
        if (slots[j] < 0 || slots[j] >= ctx->c.n_slots || slots[j] >= 64) return fail(ctx, "slot %d out of range", slots[j]);
        if (mask & (1ull << slots[j])) return fail(ctx, "vv_codec_chain_batch: slot %d listed twice", slots[j]);
        mask |= 1ull << slots[j];
    }
    const bool sem = sem_out_dev != nullptr;
    if (sem && ctx->c.sem_dim <= 0) return fail(ctx, "no semantic tokenizer configured");
    if (!ctx->side_ready) {
        for (int j = 0; j < 8; ++j) {
            HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->side[j], hipStreamNonBlocking));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_join[j], hipEventDisableTiming));
        }
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        ctx->side_ready = true;
    }
    ctx->launches = 0;
    std::string key = "chain:";
    for (int j = 0; j < n; ++j) key += std::to_string(slots[j]) + ",";
    char kp[128]; snprintf(kp, 128, ":%p:%p:%p:%d", (const void*)latent_dev, (void*)audio_out_dev, (void*)sem_out_dev, apply);
    key += kp;
    std::vector<int> ids(slots, slots + n);
    return graphed(ctx, key, st, [&]() {
        CodecNet& dec = ctx->dec; CodecNet& senc = ctx->senc;
        const int L = ctx->c.latent_dim, S = ctx->c.sem_dim, hop = ctx->hop;
        const float mul = apply ? 1.0f / ctx->scaling : 1.0f, add = apply ? -ctx->bias : 0.0f;
        const int ns_d = (int)dec.st[0].size(), ns_e = sem ? (int)senc.st[0].size() : 0;
        const bool bd = n > 1 && dec.kd > 0, be = sem && n > 1 && senc.ke < ns_e;
        const bool dec_full = bd && dec.kd == ns_d && dec.head_batch;      // the whole decoder runs slot-batched
        const bool enc_full = be && senc.ke == 0;
        const bool fork = n > 1 && !(dec_full && (!sem || enc_full));      // some part still runs per utterance
        if (bd) {
            ctx->launches++;
            VVCHK(vv_affine_slots_launch(latent_dev, dec.in_buf[0] + 6 * L, mul, add, L, ids.data(), n, dec.in_stride, st));
            if (run_codec_batch(ctx, dec, ids.data(), n, 0, dec.kd, audio_out_dev, dec_full, st)) return -1;
            if (dec_full) {
                void* tab; int nt;
                if (codec_tables_multi(ctx, dec, ids.data(), n, &tab, &nt)) return -1;
                ctx->launches++;
                VVCHK(vv_shift_rows_launch(tab, nt, dec.maxC, st));
            }
        }
        if (fork || n == 1) {
            if (fork) HIPCHK(ctx, hipEventRecord(ctx->ev_fork, st));
            for (int j = 0; j < n; ++j) {
                hipStream_t ss = fork ? ctx->side[j] : st;
                const int sl = ids[j];
                if (fork) HIPCHK(ctx, hipStreamWaitEvent(ss, ctx->ev_fork, 0));
                float* audio = audio_out_dev + (size_t)j * hop;
                if (!dec_full) {
                    if (!bd) {
                        ctx->launches++;
                        VVCHK(vv_affine_launch(latent_dev + (size_t)j * L, dec.in_buf[sl] + 6 * L, mul, add, L, ss));
                    }
                    if (run_codec(ctx, dec, sl, 1, audio, ss, bd ? dec.kd : 0, ns_d, true, true)) return -1;
                }
                if (sem) {
                    VVCHK(vv_copy_launch(senc.in_buf[sl] + 6, audio, (size_t)hop * 4, ss));
                    if (run_codec(ctx, senc, sl, 1, sem_out_dev + (size_t)j * S, ss, 0, be ? senc.ke : ns_e, !be, !be)) return -1;
                }
                if (fork) HIPCHK(ctx, hipEventRecord(ctx->ev_join[j], ss));
            }
            if (fork) for (int j = 0; j < n; ++j) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_join[j], 0));
        } else if (sem) {
            ctx->launches++;       // the batch's audio rows into the encoder's per-utterance input buffers
            VVCHK(vv_affine_slots_launch(audio_out_dev, senc.in_buf[0] + 6, 1.0f, 0.0f, hop, ids.data(), n, senc.in_stride, st));
        }
        if (be) {
            if (run_codec_batch(ctx, senc, ids.data(), n, senc.ke, ns_e, sem_out_dev, true, st)) return -1;
            void* tab; int nt;
            if (codec_tables_multi(ctx, senc, ids.data(), n, &tab, &nt)) return -1;
            ctx->launches++;
            VVCHK(vv_shift_rows_launch(tab, nt, senc.maxC, st));
        }
        return 0;
    });
}

// valid_samples: samples of real signal in wav_dev [frames * hop] (the rest must be zeros); frames = ceil(valid_samples / hop).
extern "C" int vv_acoustic_encode_ragged(vv_ctx* ctx, void* stream, int frames, long long valid_samples, const float* wav_dev, float* mean_out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (!ctx->c.has_acoustic_encoder) return fail(ctx, "no acoustic encoder configured");
    if (valid_samples <= (int64_t)(frames - 1) * ctx->hop || valid_samples > (int64_t)frames * ctx->hop)
        return fail(ctx, "vv_acoustic_encode_ragged: %lld valid samples do not end in frame %d of %d", (long long)valid_samples, frames - 1, frames);
    CodecNet& net = ctx->aenc;
    if (zero_codec(ctx, net, 0, st)) return -1;
    const int L = ctx->c.latent_dim;
    const int pass = ctx->enc_pass > 0 ? std::min(ctx->enc_pass, net.Fmax) : net.Fmax;
    for (int f0 = 0; f0 < frames; f0 += pass) {
        const int F = std::min(pass, frames - f0);
        VVCHK(vv_copy_launch(net.in_buf[0] + 6, wav_dev + (size_t)f0 * ctx->hop, (size_t)F * ctx->hop * 4, st));
        const int64_t v = valid_samples - (int64_t)f0 * ctx->hop;            // real samples inside this pass
        const int tail = (f0 + F == frames && v < (int64_t)F * ctx->hop) ? (int)v : -1;
        if (run_codec(ctx, net, 0, F, mean_out_dev + (size_t)f0 * L, st, 0, -1, true, true, tail)) return -1;
    }
    return 0;
}

extern "C" int vv_acoustic_encode(vv_ctx* ctx, void* stream, int frames, const float* wav_dev, float* mean_out_dev) {
    return vv_acoustic_encode_ragged(ctx, stream, frames, (long long)frames * ctx->hop, wav_dev, mean_out_dev);
}

extern "C" int vv_set_enc_pass_frames(vv_ctx* ctx, int frames_per_pass) {
    if (!ctx->c.has_acoustic_encoder) return fail(ctx, "no acoustic encoder configured");
    if (frames_per_pass < 1 || frames_per_pass > ctx->aenc.Fmax) return fail(ctx, "frames_per_pass %d out of range [1,%d]", frames_per_pass, ctx->aenc.Fmax);
    ctx->enc_pass = frames_per_pass;
    return 0;
}

extern "C" int vv_codec_reset(vv_ctx* ctx, void* stream, int slot) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    if (slot < 0 || slot >= ctx->c.n_slots) return fail(ctx, "slot %d out of range", slot);
    if (zero_codec(ctx, ctx->dec, slot, st)) return -1;
    if (ctx->c.sem_dim > 0 && zero_codec(ctx, ctx->senc, slot, st)) return -1;
    return 0;
}

extern "C" int vv_connect(vv_ctx* ctx, void* stream, int n, const float* latent_dev, const float* sem_dev, float* out_dev) {
    VV_SHARED;
    hipStream_t st = (hipStream_t)stream;
    const int H = ctx->H, L = ctx->c.latent_dim;
    for (int i0 = 0; i0 < n; i0 += 16) {
        const int nn = std::min(16, n - i0);
        float* out = out_dev + (size_t)i0 * H;
        VVGemm a1 = mk_gemm(ctx->ac_conn.fc1, latent_dev + (size_t)i0 * L, ctx->ct1, nn, H, L, L, H);
        a1.epi = VV_EPI_BIAS; a1.bias = ctx->ac_conn.b1; GEMM(a1);
        VVGemm a2 = mk_gemm(ctx->ac_conn.fc2, ctx->ct1, out, nn, H, H, H, H);
        a2.pro = VV_PRO_RMS; a2.nw = ctx->ac_conn.norm; a2.eps = 1e-6f; a2.epi = VV_EPI_BIAS; a2.bias = ctx->ac_conn.b2; GEMM(a2);
        if (sem_dev) {
            const int S = ctx->c.sem_dim;
            VVGemm s1 = mk_gemm(ctx->sem_conn.fc1, sem_dev + (size_t)i0 * S, ctx->ct1, nn, H, S, S, H);
            s1.epi = VV_EPI_BIAS; s1.bias = ctx->sem_conn.b1; GEMM(s1);
            VVGemm s2 = mk_gemm(ctx->sem_conn.fc2, ctx->ct1, out, nn, H, H, H, H);
            s2.pro = VV_PRO_RMS; s2.nw = ctx->sem_conn.norm; s2.eps = 1e-6f; s2.epi = VV_EPI_RESID; s2.bias = ctx->sem_conn.b2; GEMM(s2);
        }
    }
    return 0;
}

extern "C" int64_t vv_packed_bytes(int N, int K) { return vv_packed_elems(N, K) * 2; }
extern "C" int vv_pack_matrix(void* stream, const float* src_dev, void* dst_dev, int N, int K) {
    return vv_pack_launch(src_dev, 0, dst_dev, N, K, 0, 0, 0, 0, 0, (hipStream_t)stream);
}
extern "C" int vv_gemm_raw(void* stream, const void* w, const void* w2, const float* x, float* y, int T, int N, int K,
                           int ldx, int ldy, int pro, int epi, const float* nw, float eps, const float* bias,
                           const float* nscale, int xsplit, int ksplit, int nontemporal) {
    VVGemm g = mk_gemm(w, x, y, T, N, K, ldx, ldy);
    g.W2 = (const u32x4*)w2; g.pro = pro; g.epi = epi; g.nw = nw; g.eps = eps; g.bias = bias; g.nscale = nscale;
    g.ksplit = ksplit & 0xff; g.nt = 1;
    g.dbg = (unsigned long long*)(uintptr_t)0;
    if (nontemporal > 1) g.dbg = reinterpret_cast<unsigned long long*>(const_cast<float*>(nscale));   // timing builds: nscale slot carries the stamp buffer
    if (g.dbg) g.nscale = nullptr;
    return vv_gemm_launch(g, xsplit, (hipStream_t)stream);
}
// tests: Y = f(X) . W^T through the prefill GEMM (prefill.hip): X fp32 [T][K] is packed (optionally RMS-normalised) into xp_scratch,
// epi STORE/BIAS/RESID write fp32 Y [T][N]; epi SWIGLU (W = gate, W2 = up) writes packed bf16 into yp_scratch, unpacked to Y.
extern "C" int vv_gemm3_raw(vv_ctx* ctx, void* stream, const void* w, const void* w2, const float* x_dev, int T, int N, int K, int epi,
                            const float* nw_dev, float eps, const float* bias_dev, float* y_dev, void* xp_scratch, void* yp_scratch) {
    hipStream_t st = (hipStream_t)stream;
    if (ctx && ksplit_check(ctx, st)) return -1;
    int r = vv_pack_rows_launch(x_dev, K, nw_dev, eps, xp_scratch, T, K, st);
    if (r) return r;
    r = vv_gemm3_launch(w, w2, xp_scratch, y_dev, yp_scratch, bias_dev, T, N, K, N, epi, ctx ? &ctx->gws : nullptr, st);
    if (r) return r;
    if (epi == VV_EPI_SWIGLU) r = vv_unpack_rows_launch(yp_scratch, y_dev, T, N, st);
    return r;
}
extern "C" int vv_profile_begin(vv_ctx* ctx) {
    { VV_SHARED; HIPCHK(ctx, hipDeviceSynchronize()); }     // device-wide: excluded from other contexts' open captures by the lock
    ctx->prof_on = true; ctx->prof_n = 0; ctx->prof_bytes = 0.0; ctx->prof_rec.clear();
    ctx->prof_gemv.clear(); ctx->prof_gemv_bytes = 0.0; ctx->prof_other.clear();
    return 0;
}
extern "C" int vv_profile_end(vv_ctx* ctx, int64_t* launches, double* total_ms, double* bytes) {
    { VV_SHARED; HIPCHK(ctx, hipDeviceSynchronize()); }
    double ms = 0.0, raw_ms = 0.0;
    int64_t n_other = 0; double ms_other = 0.0, by_other = 0.0;
    // The fixed cost of an event pair with nothing in between, measured in the regime the samples were taken in: pairs
    // enqueued back to back on the SAME stream behind a real kernel (an idle-stream, synchronised-per-pair calibration reads
    // ~2x higher and over-corrects).  Subtracted from every sample.
    double ev_over = 0.0;
    {
        const int reps = 64;
        std::vector<hipEvent_t> ev(2 * reps);
        for (auto& e : ev) hipEventCreate(&e);
        hipStream_t ps = ctx->prof_stream;
        if (ctx->tmp1) vv_silu_launch(ctx->tmp1, 64, ps);            // something for the first pair to queue behind
        for (int i = 0; i < reps; ++i) { hipEventRecord(ev[2 * i], ps); hipEventRecord(ev[2 * i + 1], ps); }
        hipStreamSynchronize(ps);
        std::vector<float> d(reps);
        for (int i = 0; i < reps; ++i) { d[i] = 0.f; hipEventElapsedTime(&d[i], ev[2 * i], ev[2 * i + 1]); }
        std::sort(d.begin(), d.end());
        ev_over = d[reps / 2];                                        // median
        for (auto& e : ev) hipEventDestroy(e);
    }
    const char* csv = getenv("VVHIP_PROF_CSV");
    FILE* f = csv ? fopen(csv, "w") : nullptr;
    if (f) fprintf(f, "idx,T,N,K,pro,epi,dual,bytes,us\n");
    for (int i = 0; i < ctx->prof_n; ++i) {
        float e = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&e, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]));
        if (ctx->prof_rec[i].gemv) raw_ms += e;          // event-to-event time as recorded (what rocprofv3's per-kernel duration matches)
        e = (float)std::max(0.0, (double)e - ev_over);
        if (ctx->prof_rec[i].gemv) ms += e;
        else { n_other++; ms_other += e; by_other += ctx->prof_rec[i].bytes; }
        if (f) { const auto& r = ctx->prof_rec[i]; fprintf(f, "%d,%d,%d,%d,%d,%d,%d,%.0f,%.3f\n", i, r.T, r.N, r.K, r.pro, r.epi, r.dual, r.bytes, e * 1e3); }
    }
    if (f) fclose(f);
    // [0] = the dominant kernel (vv_gemv_kernel, decode rows), [1] = the general kernel (T > 4 / unaligned)
    if (launches) { launches[0] = ctx->prof_n - n_other; launches[1] = n_other; }
    if (total_ms) { total_ms[0] = ms; total_ms[1] = ms_other; }
    if (bytes) { bytes[0] = ctx->prof_bytes - by_other; bytes[1] = by_other; }
    ctx->prof_raw_ns = (int64_t)(raw_ms * 1e6); ctx->prof_ev_over_ns = (int64_t)(ev_over * 1e6);
    ctx->prof_on = false;
    return 0;
}
// Launch duration of the dominant kernel in the execution mode of the timed region: the vv_gemv_kernel launches recorded by
// the last profile window are captured, in issue order, into ONE hipGraph (a dependent chain on `stream`, as inside the step
// graphs) and replayed `reps` times between two events.  total_ms / (launches * reps) = start-to-start period of a GEMV
// launch in a dependent chain = kernel time + the kernel boundary, which is what rocprofv3 --kernel-trace reports per kernel
// under graph replay (profiles/): an upper bound on the kernel's own duration.  The replay re-runs residual epilogues on the
// engine's scratch / codec state buffers: call it after the measurements that need those states.
extern "C" int vv_profile_replay_family(vv_ctx* ctx, void* stream, int family, int reps, int64_t* launches, double* total_ms, double* bytes) {
    hipStream_t st = (hipStream_t)stream;
    if (ctx->prof_on) return fail(ctx, "vv_profile_replay: call vv_profile_end first");
    int64_t n = 0; double by = 0.0;
    if (family == 0) { n = (int64_t)ctx->prof_gemv.size(); by = ctx->prof_gemv_bytes; }
    else for (const auto& l : ctx->prof_other) if (l.family == family) { ++n; by += l.bytes; }
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0.0;
    if (bytes) *bytes = 0.0;
    if (n == 0) return family == 0 ? fail(ctx, "vv_profile_replay: the last profile window recorded no GEMV launches") : 0;
    if (reps < 1) reps = 1;
    HIPCHK(ctx, hipStreamSynchronize(st));
    hipGraph_t graph; hipGraphExec_t exec;
    HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    int rr = 0;
    if (family == 0) { for (const VVGemm& g : ctx->prof_gemv) { rr = vv_gemm_launch(g, ctx->c.xsplit, st); if (rr) break; } }
    else for (const auto& l : ctx->prof_other) if (l.family == family) { rr = l.fn(st); if (rr) break; }
    hipError_t e = hipStreamEndCapture(st, &graph);
    if (rr) return fail(ctx, "vv_profile_replay: launch failed (%d)", rr);
    HIPCHK(ctx, e);
    HIPCHK(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphDestroy(graph);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    HIPCHK(ctx, hipGraphLaunch(exec, st));                  // warm-up replay
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) HIPCHK(ctx, hipGraphLaunch(exec, st));
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipGraphExecDestroy(exec);
    if (launches) *launches = n * reps;
    if (total_ms) *total_ms = ms;
    if (bytes) *bytes = by * reps;
    return 0;
}
extern "C" int vv_profile_replay(vv_ctx* ctx, void* stream, int reps, int64_t* launches, double* total_ms, double* bytes) {
    return vv_profile_replay_family(ctx, stream, 0, reps, launches, total_ms, bytes);
}
extern "C" int64_t vv_stat(vv_ctx* ctx, int what) {
    switch (what) {
        case 0: return ctx->launches;
        case 2: return ctx->prof_raw_ns;          // last profile: sum of raw event-pair times over the decode-GEMV launches
        case 3: return ctx->prof_ev_over_ns;      // last profile: time of an empty event pair
        case 5: return ctx->foreign_nodes;        // nodes of the captured graphs that are not kernel launches (expected: 0)
        case 4: return ctx->capture_fallbacks;    // stream captures that did not close and ran eagerly instead (multi-threaded lanes)
        default: return (int64_t)ctx->graphs.size();
    }
}
