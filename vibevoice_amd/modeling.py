"""Host-side mirror of the reference's inference class for the hot path.

`VibeVoiceForConditionalGenerationInference` here keeps the reference's public
surface for this path (vibevoice/modular/modeling_vibevoice_inference.py):

    from_pretrained(path, torch_dtype=..., device_map=..., attn_implementation=...)   demo/inference_from_file.py:297-317
    eval(), set_ddpm_inference_steps(num_steps)                                         :146-147
    generate(**processor_outputs, max_new_tokens, cfg_scale, tokenizer, generation_config,
             verbose, is_prefill, audio_streamer, stop_check_fn, refresh_negative, ...)  :326-348
      -> VibeVoiceGenerationOutput(sequences, speech_outputs, reach_max_step_sample)     :38-51,691-695

but every tensor op of the loop body (:432-675) and of sample_speech_tokens
(:697-710) executes in libvvhip.so.  The Python below only does what the
reference also does on the host: token bookkeeping, stop checks, streamer calls.

Differences from the reference that do not change results:
  * per-utterance compact KV caches instead of a left-padded batch + masks
    (pads carry no information; position = cumsum(mask)-1 = compact index);
  * the CFG-negative LM row is evaluated speculatively in the same weight pass
    as the positive row (it consumes the same embedding, :579-581); if the sampled
    token turns out not to be <speech_diffusion> the appended cache entry is
    dropped by not advancing the negative length -- exactly what the reference's
    mask fix-ups (:594-624) achieve;
  * finished rows are not forwarded (the reference forwards them and discards);
  * lm_head is evaluated only on the <=5 ids the constraint processor allows
    (:53-66, :405-419): identical argmax / identical softmax over the allowed set.
"""
import json
import os
import time
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np
import torch

from .engine import Engine, EngineConfig, map_param_name


@dataclass
class VibeVoiceGenerationOutput:
    """modeling_vibevoice_inference.py:38-51"""
    sequences: torch.LongTensor = None
    speech_outputs: Optional[List[Optional[torch.Tensor]]] = None
    reach_max_step_sample: Optional[torch.BoolTensor] = None


def engine_config_from_reference(cfg: dict, **runtime) -> EngineConfig:
    """cfg: the dict form of VibeVoiceConfig (vibevoice/configs/qwen2.5_*.json)."""
    d = cfg["decoder_config"]
    h = cfg["diffusion_head_config"]
    a = cfg["acoustic_tokenizer_config"]
    s = cfg.get("semantic_tokenizer_config")
    depths = a["encoder_depths"]
    depths = [int(x) for x in depths.split("-")] if isinstance(depths, str) else list(depths)
    if a.get("decoder_depths") not in (None, "", []):
        dd = a["decoder_depths"]
        dd = [int(x) for x in dd.split("-")] if isinstance(dd, str) else list(dd)
        if dd != list(reversed(depths)):
            raise ValueError("decoder_depths other than reversed(encoder_depths) are not supported")
    if s is not None:
        sd = s["encoder_depths"]
        sd = [int(x) for x in sd.split("-")] if isinstance(sd, str) else list(sd)
        if sd != depths or list(s["encoder_ratios"]) != list(a["encoder_ratios"]) or s["encoder_n_filters"] != a["encoder_n_filters"]:
            raise ValueError("semantic and acoustic encoders must share depths/ratios/filters")
    if a.get("decoder_ratios") not in (None, []) and list(a["decoder_ratios"]) != list(a["encoder_ratios"]):
        raise ValueError("decoder_ratios != encoder_ratios is not supported")
    for k, v in (("mixer_layer", "depthwise_conv"), ("layernorm", "RMSNorm"), ("pad_mode", "constant"), ("conv_norm", "none")):
        if a.get(k, v) != v:
            raise ValueError(f"acoustic_tokenizer_config.{k}={a.get(k)!r} is not supported by the HIP codec")
    kw = dict(
        lm_hidden=d["hidden_size"], lm_layers=d["num_hidden_layers"], lm_heads=d["num_attention_heads"],
        lm_kv_heads=d["num_key_value_heads"], lm_inter=d["intermediate_size"], lm_vocab=d["vocab_size"],
        lm_eps=d.get("rms_norm_eps", 1e-6), rope_theta=d.get("rope_theta", 1e6),
        head_layers=h.get("head_layers", 4), head_ffn_ratio=h.get("head_ffn_ratio", 3.0),
        latent_dim=h.get("latent_size", 64), head_eps=h.get("rms_norm_eps", 1e-5),
        n_filters=a.get("decoder_n_filters", 32), ratios=tuple(a["encoder_ratios"]), enc_depths=tuple(depths),
        sem_dim=(s["vae_dim"] if s is not None else 0), codec_eps=a.get("layernorm_eps", 1e-5),
        max_ctx=d.get("max_position_embeddings", 32768),
    )
    kw.update(runtime)
    return EngineConfig(**kw)


class VibeVoiceForConditionalGenerationInference:
    """Drop-in for the reference class on the generate() path, backed by libvvhip.so."""

    def __init__(self, config: dict, engine: Engine, model_dtype=torch.bfloat16):
        self.config_dict = config
        self.engine = engine
        self.dtype = model_dtype
        self.device = engine.device
        self.ddpm_inference_steps = config["diffusion_head_config"].get("ddpm_num_inference_steps", 20)
        self.max_position_embeddings = config["decoder_config"].get("max_position_embeddings", 32768)
        self.acoustic_vae_dim = config.get("acoustic_vae_dim", 64)
        self.fix_std = config["acoustic_tokenizer_config"].get("fix_std", 0.5)
        self.std_dist_type = config["acoustic_tokenizer_config"].get("std_dist_type", "gaussian")
        self.speech_scaling_factor = float("nan")
        self.speech_bias_factor = float("nan")
        self._valid_key = None
        H = engine.cfg.lm_hidden
        e = engine
        R = e.cfg.max_rows
        self._x_in = e.new(R, H)
        self._hidden = e.new(R, H)
        self._logits = e.new(R, 16)
        self._cond = e.new(16, H)
        self._noise = e.new(8, engine.cfg.latent_dim)
        self._latent = e.new(8, engine.cfg.latent_dim)
        self._audio = e.new(8, engine.cfg.hop)
        self._sem = e.new(8, max(1, engine.cfg.sem_dim))
        self._emb_out = e.new(8, H)
        self._start_emb = e.new(1, H)
        self._neg_hidden = e.new(8, H)
        # pinned host staging: the logits come back without a blocking copy, noise goes out without one
        self._logits_pin = torch.empty(R, 16, dtype=torch.float32).pin_memory()
        self._noise_pin = [torch.empty(8, engine.cfg.latent_dim, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._noise_i = 0
        self._lg_event = torch.cuda.Event()
        self._fork_ev = torch.cuda.Event()
        self._join_ev = torch.cuda.Event()
        self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(min(8, engine.cfg.n_slots))] if engine.cfg.n_slots > 1 else []
        self.concurrent_codecs = os.environ.get("VVHIP_SERIAL_CODECS") is None
        self.speculate_sampling = os.environ.get("VVHIP_NO_SPEC") is None
        self.last_stats = {}

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_state_dict(cls, config: dict, state_dict, model_dtype=torch.bfloat16, device=None, **runtime):
        """state_dict: mapping (or iterable of (key, tensor)) keyed like the reference checkpoint."""
        runtime.setdefault("max_rows", 512)          # prompt rows per LM weight pass (MFMA tile GEMM above ~48 workgroups)
        ecfg = engine_config_from_reference(config, **runtime)
        eng = Engine(ecfg, device)
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict
        exp = eng.expected_weights()
        scaling = bias = None
        for k, v in items:
            if k == "model.speech_scaling_factor":
                scaling = float(v)
                continue
            if k == "model.speech_bias_factor":
                bias = float(v)
                continue
            name = map_param_name(k)
            if name is not None and name in exp:
                eng.upload(name, v)
        miss = eng.missing_weights()
        if miss:
            raise RuntimeError(f"checkpoint is missing {len(miss)} parameters, e.g. {miss[:4]}")
        m = cls(config, eng, model_dtype)
        if scaling is not None and bias is not None:
            m.set_speech_factors(scaling, bias)
        return m

    @classmethod
    def from_pretrained(cls, path, torch_dtype=torch.bfloat16, device_map=None, attn_implementation=None, **runtime):
        """Reads config.json + *.safetensors written by the reference's converter
        (scripts/convert_nnscaler_checkpoint_to_transformers.py:119-123)."""
        from safetensors import safe_open
        with open(os.path.join(path, "config.json")) as f:
            config = json.load(f)
        files = sorted(fn for fn in os.listdir(path) if fn.endswith(".safetensors"))
        if not files:
            raise FileNotFoundError(f"no .safetensors shards under {path}")

        def it():
            for fn in files:
                with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as sf:
                    for k in sf.keys():
                        yield k, sf.get_tensor(k)
        device = None
        if isinstance(device_map, (str, torch.device)) and str(device_map) not in ("auto", "cpu"):
            device = torch.device(device_map)
        m = cls.from_state_dict(config, it(), torch_dtype or torch.bfloat16, device, **runtime)
        m.source_path = path

        def base_tensor(key):          # lazy access to the checkpoint's own tensors (LoRA merge: vibevoice_amd/lora.py)
            for fn in files:
                with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as sf:
                    if key in sf.keys():
                        return sf.get_tensor(key)
            raise KeyError(key)
        m.base_tensor = base_tensor
        return m

    def set_speech_factors(self, scaling, bias):
        self.speech_scaling_factor = float(scaling)
        self.speech_bias_factor = float(bias)
        self.engine.set_speech_factors(scaling, bias)

    def eval(self):
        return self

    def set_ddpm_inference_steps(self, num_steps=None):
        self.ddpm_inference_steps = num_steps or self.config_dict["diffusion_head_config"].get("ddpm_num_inference_steps", 20)

    # ------------------------------------------------------------------ helpers
    def _embed_ids(self, ids: List[int], out: torch.Tensor):
        for i0 in range(0, len(ids), 64):
            self.engine.embed(ids[i0:i0 + 64], out[i0:])

    def _stage_noise(self, nz, n):
        """noise rows -> device without blocking the host (pageable H2D copies are synchronous with the stream)"""
        if nz.is_cuda:
            self._noise[:n].copy_(nz[:n].to(torch.float32))
            return
        pin = self._noise_pin[self._noise_i]
        self._noise_i = (self._noise_i + 1) % len(self._noise_pin)
        pin[:n].copy_(nz[:n])
        self._noise[:n].copy_(pin[:n], non_blocking=True)

    def _process_speech_inputs(self, speech_tensors, speech_masks, prefill_noise=None):
        """_process_speech_inputs (:149-163): encode voice prompts, sample, scale, connect."""
        e = self.engine
        hop = e.cfg.hop
        n_spk, S = speech_tensors.shape
        frames = S // hop
        if frames * hop != S:
            # the reference's non-streaming conv right-pads to a whole number of frames
            pad = (frames + 1) * hop - S
            speech_tensors = torch.nn.functional.pad(speech_tensors, (0, pad))
            frames += 1
        wav = speech_tensors.to(self.device, torch.float32).contiguous()
        mean = e.new(n_spk, frames, e.cfg.latent_dim)
        for i in range(n_spk):
            e.acoustic_encode(frames, wav[i], mean[i])
        if self.std_dist_type == "gaussian":
            if prefill_noise is None:
                # VibeVoiceTokenizerEncoderOutput.sample('gaussian'), modular_vibevoice_tokenizer.py:980-989:
                # two draws from the device generator
                # two draws from the device generator.  The reference's `mean` is latents.permute(0, 2, 1) (:1085) and its noise is
                # randn_like(mean): the same call on a tensor with the same strides is what stays on the reference's RNG stream
                # (a contiguous draw takes a different generator path).  Pinned by tests/golden/generate_sampled_b1.npz.
                r1 = torch.randn(n_spk, device=self.device, dtype=torch.float32)
                r2 = torch.randn_like(torch.empty(n_spk, mean.shape[2], mean.shape[1], device=self.device, dtype=torch.float32).permute(0, 2, 1))
            else:
                r1, r2 = (t.to(self.device, torch.float32) for t in prefill_noise)
            lat = mean + (r1 * (self.fix_std / 0.8))[:, None, None] * r2
        elif self.std_dist_type == "fix":
            r2 = (torch.randn_like(torch.empty(n_spk, mean.shape[2], mean.shape[1], device=self.device).permute(0, 2, 1))
                  if prefill_noise is None else prefill_noise[1].to(self.device))
            lat = mean + self.fix_std * r2
        else:
            lat = mean
        feats = ((lat + self.speech_bias_factor) * self.speech_scaling_factor).contiguous()
        sel = feats[speech_masks.to(self.device)].contiguous()            # [n_valid, 64]
        out = e.new(sel.shape[0], e.cfg.lm_hidden)
        e.connect(sel.shape[0], sel, None, out)
        return feats, out

    # ------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 prefix_allowed_tokens_fn=None, synced_gpus=None, assistant_model=None, audio_streamer=None,
                 negative_prompt_ids=None, negative_prompt_attention_mask=None, speech_tensors=None,
                 speech_masks=None, speech_input_mask=None, is_prefill=True, return_speech=True,
                 cfg_scale=1.0, stop_check_fn: Optional[Callable[[], bool]] = None, tqdm_class=None, **kwargs):
        e = self.engine
        tokenizer = kwargs.pop("tokenizer", None)
        kwargs.pop("parsed_scripts", None)
        kwargs.pop("all_speakers_list", None)
        max_length_times = kwargs.pop("max_length_times", 2)
        verbose = kwargs.get("verbose", False)
        if not kwargs.get("refresh_negative", True):
            raise NotImplementedError("refresh_negative=False is not supported by the HIP path")
        forced_tokens = kwargs.pop("_forced_tokens", None)        # test/bench hook
        noise_fn = kwargs.pop("_noise_fn", None)                  # test hook: explicit diffusion noise
        prefill_noise = kwargs.pop("_prefill_noise", None)
        trace = kwargs.pop("_trace", None)
        step_cb = kwargs.pop("_step_callback", None)             # bench hook: called at the top of every step
        kv_start = kwargs.pop("_kv_start", 0)                     # bench hook: long-context decode measurement
        input_ids = kwargs["input_ids"] if inputs is None else inputs
        attention_mask = kwargs.get("attention_mask")
        input_ids = input_ids.cpu()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        attention_mask = attention_mask.cpu()
        B, L0 = input_ids.shape
        if B > e.cfg.n_slots:
            raise ValueError(f"batch {B} exceeds the engine's n_slots={e.cfg.n_slots}")
        if 2 * B > e.cfg.max_rows:
            raise ValueError(f"batch {B} needs {2*B} LM rows > max_rows={e.cfg.max_rows}")
        gc = dict(generation_config) if isinstance(generation_config, dict) else {}
        do_sample = bool(gc.get("do_sample", False))
        if kwargs.get("max_new_tokens", None) is None:
            max_new_tokens = self.max_position_embeddings - L0
        else:
            max_new_tokens = kwargs["max_new_tokens"]
        max_length = L0 + max_new_tokens
        if max_length > e.max_ctx:
            max_length = e.max_ctx
        start_id, end_id, diff_id = tokenizer.speech_start_id, tokenizer.speech_end_id, tokenizer.speech_diffusion_id
        eos_id = tokenizer.eos_token_id
        bos_id = getattr(tokenizer, "bos_token_id", None)
        valid = [start_id, end_id, diff_id, eos_id] + ([bos_id] if bos_id is not None else [])
        if self._valid_key != tuple(valid):
            e.set_valid_tokens(valid)
            self._valid_key = tuple(valid)
        nv = len(valid)
        valid_t = torch.tensor(valid, dtype=torch.long)
        e.set_num_steps(self.ddpm_inference_steps, t_cast_bf16=(self.dtype == torch.bfloat16 and kwargs.get("_t_cast", True)))

        init_len = attention_mask.sum(-1)
        max_steps = min(max_length - L0, int(max_length_times * L0))
        max_step_per_sample = torch.min(max_length - init_len, (max_length_times * init_len).long())
        finished = torch.zeros(B, dtype=torch.bool)
        reach_max = torch.zeros(B, dtype=torch.bool)
        pos_len = [0] * B
        neg_len = [0] * B
        audio_chunks = [[] for _ in range(B)]
        seq = input_ids.clone()
        H = e.cfg.lm_hidden
        have_embeds = False
        n_frames = 0
        time_prefill = os.environ.get("VVHIP_TIME_PREFILL") is not None     # debug: sync + time the two prefill phases

        if tqdm_class is not None and kwargs.get("show_progress_bar", True):
            progress = tqdm_class(range(max_steps), desc="Generating", leave=False)
        else:
            progress = range(max_steps)

        with torch.cuda.stream(e.stream):
            for b in range(B):
                e.codec_reset(b)
            e.embed([start_id], self._start_emb)
            for step in progress:
                if step_cb is not None:
                    step_cb(step)
                if stop_check_fn is not None and stop_check_fn():
                    if audio_streamer is not None:
                        audio_streamer.end()
                    break
                if audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and any(audio_streamer.finished_flags):
                    break
                if bool(finished.all()):
                    break
                if seq.shape[-1] >= max_length:
                    reach_max[~finished] = True
                    break
                act = [b for b in range(B) if not finished[b]]
                nA = len(act)
                # ---------------- positive (+ speculative negative) LM pass ----------------
                if step == 0:
                    sp_embeds = None
                    t_pf = [time.perf_counter()] if time_prefill else None
                    if is_prefill and speech_tensors is not None and speech_masks is not None:
                        _, sp_embeds = self._process_speech_inputs(speech_tensors, speech_masks, prefill_noise)
                    if time_prefill:
                        e.sync(); t_pf.append(time.perf_counter())
                    sp_off = 0
                    for b in range(B):
                        m = attention_mask[b].bool()
                        ids = input_ids[b][m].tolist()
                        n = len(ids)
                        emb = e.new(n, H)
                        self._embed_ids(ids, emb)
                        if sp_embeds is not None and speech_input_mask is not None:
                            sm = speech_input_mask[b][m].to(self.device)
                            cnt = int(sm.sum())
                            if cnt:
                                emb[sm] = sp_embeds[sp_off:sp_off + cnt]
                                sp_off += cnt
                        CH = e.cfg.max_rows          # prompt rows per weight pass (the 16-row GEMV form walks row tiles)
                        hid = e.new(CH, H)
                        for i0 in range(0, n, CH):
                            k = min(CH, n - i0)
                            e.lm_forward([(2 * b, i0 + j) for j in range(k)], emb[i0:i0 + k], hid)
                        pos_len[b] = max(n, kv_start)
                        self._hidden[b].copy_(hid[(n - 1) % CH])
                    if time_prefill:
                        e.sync(); t_pf.append(time.perf_counter())
                        self.last_prefill = {"voice_encode_s": round(t_pf[1] - t_pf[0], 5), "lm_prefill_s": round(t_pf[2] - t_pf[1], 5)}
                    spec = False
                else:
                    rows = [(2 * b, pos_len[b]) for b in act]
                    spec = have_embeds
                    if spec:
                        rows += [(2 * b + 1, neg_len[b]) for b in act]
                        self._x_in[nA:2 * nA].copy_(self._x_in[:nA])
                    e.lm_forward(rows, self._x_in, self._hidden)
                    for b in act:
                        pos_len[b] += 1
                e.lm_logits(nA, self._hidden, self._logits)
                self._logits_pin.copy_(self._logits, non_blocking=True)      # whole (contiguous) buffer: a true async D2H
                self._lg_event.record(e.stream)
                # ---- speculative sampling: a row that has just emitted <speech_diffusion>/<speech_start> almost always
                # emits <speech_diffusion> next.  The sampler (stateless: cond + noise -> latent) is enqueued behind the LM
                # pass BEFORE the host waits for the logits, so the token decision below overlaps GPU work instead of
                # leaving the GPU idle; if the guess is wrong the latent is discarded and the RNG state restored.
                spec_sample, rng_state = False, None
                if (spec and self.speculate_sampling and step > 0
                        and not (do_sample and noise_fn is None and forced_tokens is None)   # keep the reference's RNG draw order
                        and all(int(seq[b, -1]) in (diff_id, start_id) for b in act)):
                    if noise_fn is not None:
                        nz = noise_fn(step, 2 * nA)
                    else:
                        rng_state = torch.get_rng_state()
                        nz = torch.randn(2 * nA, e.cfg.latent_dim)
                    self._stage_noise(nz, nA)
                    # all active rows diffusing, in order: cond rows == [hidden[:nA]; hidden[nA:2nA]]
                    e.diffusion_sample(nA, self._hidden, self._noise, cfg_scale, self._latent)
                    spec_sample = True
                self._lg_event.synchronize()
                logits = self._logits_pin[:nA, :nv].clone()
                if trace is not None:
                    trace.pos_hidden.append(self._hidden[:nA].cpu())
                # ---------------- token selection (:488-501) ----------------
                nxt = torch.full((B,), eos_id, dtype=torch.long)
                if forced_tokens is not None:
                    for b in act:
                        nxt[b] = forced_tokens[b][step] if step < len(forced_tokens[b]) else eos_id
                elif do_sample:
                    # the reference samples torch.multinomial(softmax(scores)) over the FULL vocabulary row (-inf outside the
                    # valid ids, :490-496) on the model's device.  One-sample multinomial spends one exponential variate per
                    # category, so the same call on the same-shaped tensor is what keeps a seeded run on the same RNG stream
                    # (pinned on CPU by tests/golden/generate_sampled_b1.npz)
                    full = torch.full((nA, e.cfg.lm_vocab), float("-inf"), device=self.device, dtype=torch.float32)
                    full[:, valid_t.to(self.device)] = self._logits[:nA, :nv].float()
                    pick_ids = torch.multinomial(torch.softmax(full, dim=-1), num_samples=1).squeeze(1).cpu()
                    for i, b in enumerate(act):
                        nxt[b] = pick_ids[i]
                else:
                    pick = torch.argmax(logits, dim=-1)
                    for i, b in enumerate(act):
                        nxt[b] = valid_t[pick[i]]
                seq = torch.cat([seq, nxt[:, None]], dim=-1)
                if trace is not None:
                    trace.tokens.append(nxt.clone())
                # ---------------- bookkeeping (:518-539) ----------------
                new_eos = (nxt == eos_id) & ~finished
                if new_eos.any():
                    finished |= new_eos
                    if verbose:
                        print(f"Samples {new_eos.nonzero().flatten().tolist()} reached EOS token at step {step + 1}.", flush=True)
                    if audio_streamer is not None:
                        audio_streamer.end(new_eos.nonzero().flatten())
                hit = (step >= max_step_per_sample) & ~finished
                if hit.any():
                    finished |= hit
                    reach_max |= hit
                    if verbose:
                        print(f"Samples {hit.nonzero().flatten().tolist()} reached max generation length at step {step + 1}.", flush=True)
                    if audio_streamer is not None:
                        audio_streamer.end(hit.nonzero().flatten())
                for b in (nxt == end_id).nonzero().flatten().tolist():
                    e.codec_reset(b)
                for b in (~finished & (nxt == start_id)).nonzero().flatten().tolist():
                    # :549-565 -- the reference masks the whole negative cache and un-masks only the slot of the NEXT token, so the
                    # negative context restarts empty and the next negative pass re-feeds <speech_start> at position 0
                    # (pinned against the reference's generate(): tests/golden/generate_forced_*.npz)
                    neg_len[b] = 0
                # ---------------- next input embeddings (:569) ----------------
                live = [b for b in range(B) if not finished[b]]
                diff = [b for b in live if int(nxt[b]) == diff_id]
                # rows of _x_in are re-packed to the next step's active order
                nxt_x = e.new(max(1, len(live)), H)
                plain = [b for b in live if b not in diff]
                if plain:
                    tmp = e.new(len(plain), H)
                    self._embed_ids([int(nxt[b]) for b in plain], tmp)
                    for i, b in enumerate(plain):
                        nxt_x[live.index(b)].copy_(tmp[i])
                if spec_sample and diff != act:
                    spec_sample = False                                  # wrong guess: drop the latent, undo the draw
                    if rng_state is not None:
                        torch.set_rng_state(rng_state)
                cond_used = self._hidden if spec_sample else self._cond
                if diff and spec_sample:
                    n = len(diff)
                    for b in diff:
                        neg_len[b] += 1
                elif diff:
                    n = len(diff)
                    # ---- negative condition ----
                    for j, b in enumerate(diff):
                        ai = act.index(b)
                        self._cond[j].copy_(self._hidden[ai])
                        if spec:
                            self._cond[n + j].copy_(self._hidden[nA + ai])
                        else:
                            # first negative step: the lone <speech_start> prompt token (:379-386) or, later,
                            # the embedding the positive pass just consumed
                            src = self._start_emb if not have_embeds else self._x_in[ai:ai + 1]
                            e.lm_forward([(2 * b + 1, neg_len[b])], src, self._neg_hidden[j:j + 1])
                            self._cond[n + j].copy_(self._neg_hidden[j])
                        neg_len[b] += 1
                    # ---- diffusion sampling (:697-710) ----
                    if noise_fn is not None:
                        nz = noise_fn(step, 2 * n)
                    else:
                        nz = torch.randn(2 * n, e.cfg.latent_dim)      # CPU global RNG, as the reference (:701)
                    self._stage_noise(nz, n)
                    e.diffusion_sample(n, self._cond, self._noise, cfg_scale, self._latent)
                if diff:
                    # ---- codec decode, semantic encode, connectors (:636-672) ----
                    if len(diff) > 1 and self.concurrent_codecs:
                        # each utterance's tokenizer chain (decode -> semantic re-encode) is an independent, launch-latency
                        # bound graph: fork them onto side streams so they overlap, join before the connectors
                        self._fork_ev.record(e.stream)
                        for j, b in enumerate(diff):
                            ss = self._side_streams[j % len(self._side_streams)]
                            ss.wait_event(self._fork_ev)
                            e.codec_decode(b, self._latent[j:j + 1], self._audio[j], stream=ss)
                            if e.cfg.sem_dim > 0:
                                e.semantic_encode(b, self._audio[j], self._sem[j], stream=ss)
                        for ss in self._side_streams[:min(len(diff), len(self._side_streams))]:
                            self._join_ev.record(ss)
                            e.stream.wait_event(self._join_ev)
                    else:
                        for j, b in enumerate(diff):
                            e.codec_decode(b, self._latent[j:j + 1], self._audio[j])
                            if e.cfg.sem_dim > 0:
                                e.semantic_encode(b, self._audio[j], self._sem[j])
                    e.connect(n, self._latent, self._sem if e.cfg.sem_dim > 0 else None, self._emb_out)
                    chunk = self._audio[:n].clone()
                    for j, b in enumerate(diff):
                        audio_chunks[b].append(chunk[j])
                        nxt_x[live.index(b)].copy_(self._emb_out[j])
                    if audio_streamer is not None:
                        audio_streamer.put(chunk[:, None, :].to(self.dtype), torch.tensor(diff))
                    n_frames += n
                    if trace is not None:
                        trace.neg_hidden.append(cond_used[n:2 * n].cpu())
                        trace.latents.append(self._latent[:n].cpu())
                        trace.semantic.append(self._sem[:n].cpu())
                if live:
                    self._x_in[:len(live)].copy_(nxt_x[:len(live)])
                    if trace is not None:
                        trace.next_embeds.append(nxt_x[:len(live)].cpu())
                have_embeds = True
            if audio_streamer is not None:
                audio_streamer.end()
            outs = []
            for c in audio_chunks:
                outs.append(torch.cat(c, dim=-1)[None].to(self.dtype) if c else None)
        e.sync()
        self.last_stats = {"frames": n_frames, "steps": seq.shape[-1] - L0}
        return VibeVoiceGenerationOutput(
            sequences=seq.to(self.device), speech_outputs=outs if return_speech else None,
            reach_max_step_sample=reach_max.to(self.device))
