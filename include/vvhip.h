/* vvhip.h -- C ABI of libvvhip.so, the MI355X (gfx950) engine behind
 * VibeVoiceForConditionalGenerationInference.generate().
 *
 * The reference (vibevoice-community/VibeVoice) is pure Python: its "operator
 * interface" for this path is the set of nn.Module calls generate() makes per
 * step.  Each entry point below replaces one of those calls (file:line into
 * /root/reference/vibevoice/modular/); INTEGRATION.md shows the ctypes binding a
 * maintainer would add to modeling_vibevoice_inference.py.
 *
 * Conventions: plain C, opaque context, no torch types.  Pointers suffixed _dev
 * are device pointers (e.g. tensor.data_ptr()); the caller keeps them alive until
 * the stream has consumed them.  `stream` is a hipStream_t passed as void*
 * (0 = default stream).  Every function returns 0 on success, <0 on error
 * (vv_last_error() gives the text).  No hidden syncs except where stated; no
 * allocation after vv_create() except lazily-built hipGraphs and scratch on the
 * first call of a given shape.
 */
#ifndef VVHIP_H
#define VVHIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
/* libvvhip.so is built with -fvisibility=hidden: the functions declared in this header are its whole dynamic symbol table (the
 * kernel-launch helpers the translation units share stay internal; __graft_entry__.build() fails on any other exported vv_*). */
typedef struct vv_ctx vv_ctx;

typedef struct vv_config {
    /* Qwen2 decoder (configs/qwen2.5_*.json "decoder_config") */
    int lm_hidden, lm_layers, lm_heads, lm_kv_heads, lm_head_dim, lm_inter, lm_vocab;
    float lm_eps;
    /* diffusion head ("diffusion_head_config") */
    int head_layers, head_ffn, latent_dim;
    float head_eps;
    /* tokenizers ("acoustic_tokenizer_config", "semantic_tokenizer_config") */
    int n_filters, n_ratios, ratios[8];
    int n_stages, enc_depths[8]; /* encoder order; decoder uses the reverse */
    int sem_dim;                 /* semantic vae_dim (0 = no semantic tokenizer) */
    int has_acoustic_encoder;    /* voice-prompt path */
    float codec_eps;
    /* runtime */
    int n_slots;       /* concurrent utterances: 2 KV caches (cond/uncond) + 2 conv states each */
    int max_ctx;       /* KV positions per cache (rounded up to 128) */
    int max_rows;      /* max LM rows per launch (<=16384): decode uses 2 per utterance, prompt prefill fills it */
    int xsplit;        /* activation precision inside MFMA: 1 bf16, 2 ~fp24, 3 fp32-exact */
    int attn_splits;   /* flash-decoding splits along the sequence */
    int enc_frames;    /* frames per chunk of the voice-prompt encoder (>=1) */
    int use_graph;     /* replay captured hipGraphs for repeated shapes */
    /* Streaming-0.5B split LM (modeling_vibevoice_streaming.py:108-164): the last `tts_layers` of the
     * lm_layers stack form the TTS LM (own final norm), the first lm_layers-tts_layers the text LM
     * (final norm = Identity); adds tts_input_types + the binary EOS classifier.  0 = ordinary model. */
    int tts_layers;
} vv_config;

#pragma GCC visibility push(default)
int vv_create(const vv_config* cfg, vv_ctx** out);
/* A second context over the SAME weight storage as `parent` (fully uploaded, not itself shared): its own KV caches, activations,
 * tokenizer state, graphs and staging, sized by cfg's runtime fields (n_slots, max_ctx, max_rows, attn_splits, use_graph); the model
 * fields must equal the parent's.  Two contexts driven on two streams interleave two independent utterance batches on one GPU over
 * one copy of the weights (new surface: the reference shares weights between concurrent generate() calls by being one nn.Module,
 * and forbids the concurrency -- it is not re-entrant).  vv_upload / LoRA merges go through the parent, BEFORE any child
 * exists: a child snapshots what it derives from the parameters (lm_head / tied table, valid-token rows, RoPE table, speech factors), so
 * vv_upload on a parent with live children is refused -- destroy the children, upload, create them again.  Destroying the parent
 * first is allowed (the storage lives until the last child is destroyed). */
int vv_create_shared(const vv_config* cfg, vv_ctx* parent, vv_ctx** out);
void vv_destroy(vv_ctx* ctx);
const char* vv_last_error(vv_ctx* ctx);
/* "VVHIP_BUILD_ID=<sha256[:16] of the sources and flags the library was compiled from>": the Python loader refuses
 * (or rebuilds) a binary whose id differs from the sources beside it.  New surface: the reference has no native
 * code, hence no counterpart (its "is the install current" check is pip's, pyproject.toml:1-40). */
const char* vv_build_id(void);
/* Asynchronous device-side errors, checked on the host: today the one inter-workgroup hand-off on the path (the K-split of the
 * long-prompt GEMM's partial round, prefill.hip) reports a lost producer through a host-mapped word instead of spinning.  Waits
 * for `stream`, and if the word is set: re-arms the hand-off state, clears it and returns <0 (the prompt pass that was in flight
 * produced a wrong tile and must be repeated; the context stays usable).  The same check runs at the head of every vv_lm_forward*.  Callers: after the
 * stream sync that ends a prompt prefill.  New surface (the reference has no native code); its closest counterpart is the
 * CUDA error a torch.cuda.synchronize() raises after modeling_vibevoice_inference.py:467-482. */
int vv_check(vv_ctx* ctx, void* stream);

/* ---- parameters.  Names are the reference state_dict keys with the prefixes
 * model.language_model.->"lm."  model.prediction_head.->"head."
 * model.acoustic_tokenizer.decoder.->"dec."  model.acoustic_tokenizer.encoder.->"aenc."
 * model.semantic_tokenizer.encoder.->"senc."  model.acoustic_connector.->"ac_conn."
 * model.semantic_connector.->"sem_conn."  lm_head.weight (optional: tied -> embed_tokens),
 * plus "lm.rope.inv_freq" (fp32 [head_dim/2], the table HF computes in Qwen2RotaryEmbedding).
 * The engine repacks every matrix into MFMA-fragment tiles (bf16). */
int vv_num_weights(vv_ctx* ctx);
int vv_weight_info(vv_ctx* ctx, int idx, char* name, int name_cap, int64_t* nelem, int* loaded);
/* src may be a host or a device pointer; src_dtype 0 = fp32, 1 = bf16. Synchronous. */
int vv_upload(vv_ctx* ctx, const char* name, const void* src, int src_dtype, int64_t nelem);
/* scalar buffers speech_scaling_factor / speech_bias_factor (modeling_vibevoice.py:131-132) */
int vv_set_speech_factors(vv_ctx* ctx, float scaling, float bias);
/* ids the constrained sampler may emit (VibeVoiceTokenConstraintProcessor, modeling_vibevoice_inference.py:53-66,405-419) */
int vv_set_valid_tokens(vv_ctx* ctx, const int* ids, int n);
/* DPM-Solver++ table for N steps: t[N] (the timestep values fed to the head) and coef[N][5] =
 * {alpha_i, sigma_i, sigma_{i+1}/sigma_i, -alpha_{i+1}(e^{-h}-1), second-order term}
 * (dpm_solver.py:321-423,581-584,669-677,738-764) -- computed by vibevoice_amd/schedule.py */
int vv_set_schedule(vv_ctx* ctx, int n_steps, const float* t, const float* coef, void* stream);
/* The sde-dpmsolver++ table (noise_scheduler.from_config(..., algorithm_type='sde-dpmsolver++'), demo/gradio_demo.py:142-146):
 * coef6[N][6] = {alpha_i, sigma_i, (sigma_{i+1}/sigma_i) e^{-h}, alpha_{i+1}(1 - e^{-2h}), second-order term,
 * sigma_{i+1} sqrt(1 - e^{-2h})} (dpm_solver.py:680-686,785-793); sample with vv_diffusion_sample_sde */
int vv_set_schedule_sde(vv_ctx* ctx, int n_steps, const float* t, const float* coef6, void* stream);

/* ---- KV caches: cache id = 2*slot (+1 for the CFG-negative branch) */
typedef struct vv_row { int cache; int pos; } vv_row;

/* One Qwen2 forward over n_rows tokens, each appended to its cache at rows[i].pos
 * (== the cache length before this token).  Replaces self(**model_inputs, ...)
 * (modeling_vibevoice_inference.py:480-482 -> modeling_vibevoice.py:187-199 -> HF Qwen2Model)
 * for the positive rows and :583-585 for the negative rows -- both in ONE pass
 * over the weights.  hidden_out = last_hidden_state (after the final RMSNorm).
 * Row sets: (a) every row a different cache (decode steps; one fused attention launch per layer), or (b) consecutive
 * positions of one cache (prompt prefill, up to max_rows rows; LDS-staged MFMA GEMM + 64-row prefill attention), or (c) any other
 * mix of at most 64 rows (3-launch attention: all appends land before any row attends). */
int vv_lm_forward(vv_ctx* ctx, void* stream, int n_rows, const vv_row* rows,
                  const float* x_in_dev, float* hidden_out_dev);
/* Same over the layer range [layer_begin, layer_end) only; final_norm selects whether lm.norm is applied.
 * Streaming-0.5B: forward_lm = layers [0, n_lm) without norm (modeling_vibevoice_streaming_inference.py:181-241),
 * forward_tts_lm = layers [n_lm, n_lm+n_tts) with norm (:243-318). */
int vv_lm_forward_range(vv_ctx* ctx, void* stream, int n_rows, const vv_row* rows, const float* x_in_dev,
                        float* hidden_out_dev, int layer_begin, int layer_end, int final_norm);
/* Import n_pos cached positions of one layer into cache `cache` from HF layout k/v [kv_heads][n_pos][head_dim]
 * (keys already rotated, as DynamicCache stores them); src_dtype 0 fp32, 1 bf16.  Used for the voice presets
 * (demo/voices/streaming_model/ presets -> all_prefilled_outputs). */
int vv_kv_import(vv_ctx* ctx, void* stream, int cache, int layer, int n_pos, const void* k_dev, const void* v_dev,
                 int src_dtype);
/* The same for cache positions [pos0, pos0 + n_pos): source row i lands at position pos0 + i.  Lets a caller resume an
 * utterance from KV state computed elsewhere (and the long-context tests / bench place known K/V at chosen positions). */
int vv_kv_import_at(vv_ctx* ctx, void* stream, int cache, int layer, int pos0, int n_pos, const void* k_dev,
                    const void* v_dev, int src_dtype);
/* Cached position src_pos copied onto dst_pos in every layer of `cache` (keys keep the rotation they were computed with).  The one
 * place the reference re-arranges a row's negative cache so that an entry ends up at another index with its ORIGINAL rotation: the
 * correction of a non-diffusing batch row that holds exactly one valid entry (modeling_vibevoice_inference.py:594-624: the mask
 * shifts, :603, and the K/V does not, :613 -- the entry appended at that step stays, the older one is masked out). */
int vv_kv_move(vv_ctx* ctx, void* stream, int cache, int src_pos, int dst_pos);
/* y[t][:] = x[t][:] + tts_input_types[type]  (forward_tts_lm, :293) */
int vv_add_type_embedding(vv_ctx* ctx, void* stream, int n, const float* x_dev, int type, float* out_dev);
/* tts_eos_classifier: fc2(relu(fc1(h))) -> out_dev[n] logits (BinaryClassifier, modeling_vibevoice_streaming.py:42-53) */
int vv_eos_logit(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* out_dev);
/* embed_tokens lookup (modeling_vibevoice_inference.py:218,569); ids on host, 1 <= n <= max(64, max_rows) per call */
int vv_embed(vv_ctx* ctx, void* stream, int n, const int* ids, float* out_dev);
/* logits restricted to the valid ids: replaces lm_head + constraint mask (:241-242,488-490).
 * logits_out_dev [n][n_valid] fp32 in the order given to vv_set_valid_tokens. */
int vv_lm_logits(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* logits_out_dev);
/* lm_head over the whole vocabulary (:241-242, `outputs.logits[:, -1, :]` of :486): logits_out_dev [n][lm_vocab] fp32, 1 <= n <= 16.
 * Needed only when a full-vocabulary logits processor is requested (top-k / top-p / min-p / repetition penalty run before the
 * valid-id constraint in the reference's processor list, :310-319); the default path evaluates the valid rows only. */
int vv_lm_logits_full(vv_ctx* ctx, void* stream, int n, const float* hidden_dev, float* logits_out_dev);

/* sample_speech_tokens (:697-710): cond_dev [2n][H] = n positive then n negative
 * conditions, noise_dev [n][latent], -> latent_out_dev [n][latent] */
int vv_diffusion_sample(vv_ctx* ctx, void* stream, int n, const float* cond_dev, const float* noise_dev,
                        float cfg_scale, float* latent_out_dev);
/* The same sampler under the stochastic solver the gradio demo installs (demo/gradio_demo.py:142-146:
 * noise_scheduler.from_config(config, algorithm_type='sde-dpmsolver++', ...)): scheduler.step() adds
 * sigma_t sqrt(1 - e^{-2h}) * eps_i with eps_i = randn(model_output.shape) drawn per solver step
 * (vibevoice/schedule/dpm_solver.py:680-686, 785-793, 994-997).  step_noise_dev [n_steps][n][latent] fp32
 * = those draws (first n rows of each: only they reach the next step, modeling_vibevoice_inference.py:703-704).
 * Needs a table from vv_set_schedule_sde; vv_diffusion_sample refuses to run on a stochastic table. */
int vv_diffusion_sample_sde(vv_ctx* ctx, void* stream, int n, const float* cond_dev, const float* noise_dev,
                            const float* step_noise_dev, float cfg_scale, float* latent_out_dev);
/* one prediction_head forward (modular_vibevoice_diffusion_head.py:254-280) for tests:
 * noisy [n][latent], t[n] (host), cond [n][H] -> out [n][latent].  Synchronous. */
int vv_head_forward(vv_ctx* ctx, void* stream, int n, const float* noisy_dev, const float* t_host,
                    const float* cond_dev, float* out_dev);

/* acoustic_tokenizer.decode(latent/scale - bias, cache, use_cache=True) (:636-643) for one
 * utterance slot: latent_dev [frames][latent] -> audio_out_dev [frames*hop]. */
int vv_codec_decode(vv_ctx* ctx, void* stream, int slot, int frames, const float* latent_dev,
                    float* audio_out_dev, int apply_speech_factors);
/* semantic_tokenizer.encode(audio, cache, use_cache=True).mean (:658-664) */
int vv_semantic_encode(vv_ctx* ctx, void* stream, int slot, int frames, const float* audio_dev,
                       float* sem_out_dev);
/* One frame of n utterances (1..8) through both tokenizers in one call: the reference decodes / re-encodes the whole set of
 * diffusion rows of a step as ONE batch -- acoustic_tokenizer.decode(scaled_latent, cache=acoustic_cache,
 * sample_indices=diffusion_indices) then semantic_tokenizer.encode(audio_chunk, cache=semantic_cache, sample_indices=...)
 * (modeling_vibevoice_inference.py:636-672; per-sample cache rows: modular_vibevoice_tokenizer.py:76-126).  Row j of
 * latent_dev [n][latent] / audio_out_dev [n][hop] / sem_out_dev [n][sem_dim] belongs to streaming slot slots[j] (distinct).
 * sem_out_dev = NULL: decode only.  Same results as n vv_codec_decode + vv_semantic_encode calls up to bf16 summation order;
 * the weight-heavy stages of both nets read their weights once for the whole batch. */
int vv_codec_chain_batch(vv_ctx* ctx, void* stream, int n, const int* slots, const float* latent_dev,
                         float* audio_out_dev, float* sem_out_dev, int apply_speech_factors);
/* acoustic_tokenizer.encode(wav).mean, non-streaming (:154; modular_vibevoice_tokenizer.py:1081-1085):
 * wav_dev [frames*hop] -> mean_out_dev [frames][latent] */
int vv_acoustic_encode(vv_ctx* ctx, void* stream, int frames, const float* wav_dev, float* mean_out_dev);
/* The same for a signal that does not fill its last frame: valid_samples in ((frames - 1) * hop, frames * hop], wav_dev zero beyond them.
 * The reference right-pads PER strided conv layer (SConv1d.forward -> get_extra_padding_for_conv1d, modular_vibevoice_tokenizer.py):
 * past the end of the signal every strided conv reads zeros, not the activations a zero waveform would produce.  Only the last,
 * partial frame's latent depends on it (every layer is causal). */
int vv_acoustic_encode_ragged(vv_ctx* ctx, void* stream, int frames, long long valid_samples, const float* wav_dev, float* mean_out_dev);
/* Frames the voice-prompt encoder takes per pass, 1..config.enc_frames (the buffers are sized for enc_frames; the default).
 * The reference's non-streaming encode runs the whole prompt as one sequence (modular_vibevoice_tokenizer.py:384-418,
 * 1081-1085); a pass of F frames is that computation on F frames with the causal history carried over, so the result does
 * not depend on F -- tests/test_gpu_shipped.py holds every pass size to the oracle's one-sequence encode. */
int vv_set_enc_pass_frames(vv_ctx* ctx, int frames_per_pass);
/* 16-bit PCM of n chunks of `samples` fp32 samples each, on device, before the chunk leaves for the host: the arithmetic
 * of convert_to_16_bit_wav (demo/gradio_demo.py:1058-1073, applied per streamed chunk at :404-418): peak = max|x| of the
 * chunk; x /= peak when peak > 1; int16(trunc(x * 32767)).  audio_dev [n][samples] fp32 -> pcm_out_dev [n][samples] int16. */
int vv_audio_to_pcm16(vv_ctx* ctx, void* stream, int n, int samples, const float* audio_dev, int16_t* pcm_out_dev);
/* acoustic_cache.set_to_zero + semantic_cache.set_to_zero for a slot (:542-546) */
int vv_codec_reset(vv_ctx* ctx, void* stream, int slot);
/* acoustic_connector(latent) [+ semantic_connector(sem)] (:667-669, :161); sem_dev may be NULL */
int vv_connect(vv_ctx* ctx, void* stream, int n, const float* latent_dev, const float* sem_dev,
               float* embeds_out_dev);

/* ---- low-level entry points (tests, microbenchmarks) */
/* pack a row-major fp32 [N][K] matrix into fragment tiles; dst needs vv_packed_bytes(N,K) */
int64_t vv_packed_bytes(int N, int K);
int vv_pack_matrix(void* stream, const float* src_dev, void* dst_dev, int N, int K);
/* Y[T][N] = X[T][K] . W^T with the given prologue/epilogue ids (vv_common.h) */
int vv_gemm_raw(void* stream, const void* w_packed_dev, const void* w2_packed_dev, const float* x_dev,
                float* y_dev, int T, int N, int K, int ldx, int ldy, int pro, int epi,
                const float* nw_dev, float eps, const float* bias_dev, const float* nscale_dev,
                int xsplit, int ksplit, int nontemporal);
/* The prompt-prefill GEMM (bf16-activation mode): x_dev fp32 [T][K] is packed (RMS-normalised when nw_dev != NULL) into
 * xp_scratch (vv_packed_bytes(T, K)); epi 0/1/4 (store / bias / residual) -> fp32 y_dev [T][N]; epi 3 (SwiGLU: w = gate,
 * w2 = up) -> packed bf16 in yp_scratch (vv_packed_bytes(T, N), zero-initialised), unpacked into y_dev.  K % 8 == 0, N % 4 == 0.
 * ctx lends its K-split workspace (the partial last round of the 256 x 256 kernel is split along K, as in the prompt prefill);
 * ctx == NULL: every tile is computed whole. */
int vv_gemm3_raw(vv_ctx* ctx, void* stream, const void* w_packed_dev, const void* w2_packed_dev, const float* x_dev, int T, int N, int K,
                 int epi, const float* nw_dev, float eps, const float* bias_dev, float* y_dev, void* xp_scratch, void* yp_scratch);
/* hipEvent timing of every GEMM launch issued between begin and end (graphs are bypassed meanwhile):
 * number of launches, summed kernel time, summed algorithmic bytes (packed weights once + activations
 * in + result out).  Each output is a 2-element array: [0] the decode kernel vv_gemv_kernel with T <= 4,
 * [1] the general vv_gemm_kernel.  Used by bench.py's roofline. */
int vv_profile_begin(vv_ctx* ctx);
int vv_profile_end(vv_ctx* ctx, int64_t* launches, double* total_ms, double* bytes);
/* The vv_gemv_kernel launches recorded by the last begin/end window, captured in issue order into one hipGraph and replayed
 * `reps` times between two events on `stream`: launches = recorded x reps, total_ms, algorithmic bytes.  total_ms/launches
 * is the start-to-start period of a GEMV launch inside a dependent graph chain (kernel + boundary) -- the execution mode of
 * the timed region, and what rocprofv3 --kernel-trace reports per kernel under graph replay.  Clobbers engine scratch and
 * streaming-codec state: call it last. */
int vv_profile_replay(vv_ctx* ctx, void* stream, int reps, int64_t* launches, double* total_ms, double* bytes);
/* The same replay for the other timed kernel families of the window: family 0 = the GEMV kernel as above, 1 = vv_gemv16p_kernel,
 * the packed-activation projections of batch decode with 5..16 rows, 2 = decode attention: vv_attn_fused_kernel and, for contexts
 * split over several workgroups, its vv_attn_merge2_kernel: one unit per layer; algorithmic bytes = every cached position's K and
 * V once.  launches = 0 when the window recorded none of that family.  bench.py's roofline for batch decode and for the
 * attention kernels. */
int vv_profile_replay_family(vv_ctx* ctx, void* stream, int family, int reps, int64_t* launches, double* total_ms, double* bytes);
/* what = 0: kernel launches issued by the last engine call (graph nodes when replayed); 1: hipGraph executables cached; 2 / 3: raw
 * timings of the last profile window; 4: stream captures that did not close cleanly and were run eagerly instead (seen when
 * several host threads drive several contexts: another thread's activity can invalidate a capture; the result is unaffected);
 * 5: nodes of the captured graphs that are not kernel launches -- 0 by construction: copies and fills inside captured sequences are
 * kernels of the library, because a memset node of a replayed graph was seen to fill with stale words */
int64_t vv_stat(vv_ctx* ctx, int what);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
