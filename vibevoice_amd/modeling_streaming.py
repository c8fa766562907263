"""Host-side mirror of `VibeVoiceStreamingForConditionalGenerationInference.generate`
(vibevoice/modular/modeling_vibevoice_streaming_inference.py:412-725) on the HIP engine.

Text is fed in windows of 5 tokens through the text LM (lower layers, no final norm) and the
TTS LM (upper layers); after each window 6 speech frames are produced: CFG diffusion sample
-> acoustic decode -> connector -> TTS-LM step on the positive and the negative branch (ONE
weight pass for both rows) -> binary EOS head.  The four prefilled branches of a voice preset
(`all_prefilled_outputs`) are imported into the engine's KV caches with `vv_kv_import`.
"""
from typing import Callable, Optional

import torch

import json
import os

import numpy as np

from .engine import Engine, EngineConfig
from .modeling import VibeVoiceGenerationOutput, WeightHandle, _Ns, _SchedulerView, engine_config_from_reference

TTS_TEXT_WINDOW_SIZE = 5
TTS_SPEECH_WINDOW_SIZE = 6

LM_CACHE, TTS_CACHE, NEG_TTS_CACHE = 0, 1, 2


def map_streaming_param_name(key: str, n_lm: int):
    """reference state_dict key -> engine parameter name (None = not used on this path)."""
    import re
    m = re.match(r"model\.language_model\.layers\.(\d+)\.(.*)", key)
    if m:
        return f"lm.layers.{int(m.group(1))}.{m.group(2)}"
    m = re.match(r"model\.tts_language_model\.layers\.(\d+)\.(.*)", key)
    if m:
        return f"lm.layers.{n_lm + int(m.group(1))}.{m.group(2)}"
    table = {
        "model.language_model.embed_tokens.weight": "lm.embed_tokens.weight",
        "model.tts_language_model.norm.weight": "lm.norm.weight",
        "model.tts_input_types.weight": "tts_input_types.weight",
    }
    if key in table:
        return table[key]
    for a, b in (("tts_eos_classifier.", "eos."), ("model.prediction_head.", "head."),
                 ("model.acoustic_tokenizer.decoder.", "dec."), ("model.acoustic_connector.", "ac_conn.")):
        if key.startswith(a):
            return b + key[len(a):]
    return None


_STREAM_PREFIXES = (("tts_eos_classifier.", "eos."), ("model.prediction_head.", "head."),
                    ("model.acoustic_tokenizer.decoder.", "dec."), ("model.acoustic_connector.", "ac_conn."))


def unmap_streaming_param_name(name: str, n_lm: int):
    """engine parameter name -> reference state_dict key (inverse of map_streaming_param_name)."""
    import re
    m = re.match(r"lm\.layers\.(\d+)\.(.*)", name)
    if m:
        i = int(m.group(1))
        return (f"model.language_model.layers.{i}.{m.group(2)}" if i < n_lm
                else f"model.tts_language_model.layers.{i - n_lm}.{m.group(2)}")
    table = {"lm.embed_tokens.weight": "model.language_model.embed_tokens.weight",
             "lm.norm.weight": "model.tts_language_model.norm.weight", "tts_input_types.weight": "model.tts_input_types.weight"}
    if name in table:
        return table[name]
    for a, b in _STREAM_PREFIXES:
        if name.startswith(b):
            return a + name[len(b):]
    return None


class _StreamingModelNamespace:
    """`model.model` (VibeVoiceStreamingModel, modeling_vibevoice_streaming.py:108-164) as far as callers read it:
    the two halves of the split decoder, the diffusion head, the acoustic tokenizer / connector and the input-type embedding
    as weight-snapshot handles, `language_model.config._attn_implementation` (demo/streaming_inference_from_file.py:285-286)."""

    def __init__(self, owner, config: dict, attn_implementation: str):
        d = dict(config.get("decoder_config", {}))
        d["_attn_implementation"] = attn_implementation
        n_lm = owner.n_lm
        kw = dict(to_engine=lambda k: map_streaming_param_name(k, n_lm), to_reference=lambda n: unmap_streaming_param_name(n, n_lm))
        self.language_model = WeightHandle(owner, "model.language_model.", dict(d, num_hidden_layers=n_lm), **kw)
        self.tts_language_model = WeightHandle(owner, "model.tts_language_model.", dict(d, num_hidden_layers=owner.n_tts), **kw)
        self.tts_input_types = WeightHandle(owner, "model.tts_input_types.", **kw)
        self.prediction_head = WeightHandle(owner, "model.prediction_head.", config.get("diffusion_head_config"), **kw)
        self.acoustic_tokenizer = WeightHandle(owner, "model.acoustic_tokenizer.", config.get("acoustic_tokenizer_config"), **kw)
        self.acoustic_connector = WeightHandle(owner, "model.acoustic_connector.", **kw)
        self._owner = owner

    @property
    def speech_scaling_factor(self):
        return torch.tensor(self._owner.speech_scaling_factor)

    @property
    def speech_bias_factor(self):
        return torch.tensor(self._owner.speech_bias_factor)

    @property
    def noise_scheduler(self):
        return self._owner.noise_scheduler


def _kv_layers(past):
    """[(k, v)] per layer from an HF DynamicCache (4.x `.key_cache`, 5.x `.layers`) or a plain list."""
    if hasattr(past, "key_cache"):
        return list(zip(past.key_cache, past.value_cache))
    if hasattr(past, "layers"):
        return [(l.keys, l.values) for l in past.layers]
    return list(past)


class VibeVoiceStreamingForConditionalGenerationInference:
    """Drop-in for the reference class on the streaming generate() path (demo/streaming_inference_from_file.py:226-355:
    from_pretrained -> eval -> set_ddpm_inference_steps -> model.model.language_model.config._attn_implementation ->
    generate(**processor_inputs, all_prefilled_outputs=preset, ...)), backed by libvvhip.so."""

    def __init__(self, config: dict, engine: Engine, model_dtype=torch.bfloat16, attn_implementation: Optional[str] = None):
        self.config_dict = config
        self.config = _Ns(config)
        self.engine = engine
        self.dtype = model_dtype
        self.device = engine.device
        self.n_tts = engine.cfg.tts_layers
        self.n_lm = engine.cfg.lm_layers - self.n_tts
        # what the attention really is on this path; the value the caller asked for is kept beside it
        self.requested_attn_implementation = attn_implementation
        self._param_placeholder = torch.empty(0, dtype=model_dtype, device=self.device)
        self.model = _StreamingModelNamespace(self, config, "vvhip_mfma_flash_decoding_gfx950")
        self.tts_eos_classifier = WeightHandle(self, "tts_eos_classifier.",
                                               to_engine=lambda k: map_streaming_param_name(k, self.n_lm),
                                               to_reference=lambda n: unmap_streaming_param_name(n, self.n_lm))
        self.ddpm_inference_steps = config["diffusion_head_config"].get("ddpm_num_inference_steps", 20)
        self.max_position_embeddings = config["decoder_config"].get("max_position_embeddings", 8192)
        self.speech_scaling_factor = float("nan")
        self.speech_bias_factor = float("nan")
        e = engine
        H = e.cfg.lm_hidden
        self._x = e.new(8, H)
        self._h = e.new(8, H)
        self._hid = e.new(8, H)
        self._cond = e.new(2, H)
        self._noise = e.new(1, e.cfg.latent_dim)
        self._latent = e.new(1, e.cfg.latent_dim)
        self._audio = e.new(1, e.cfg.hop)
        self._emb = e.new(1, H)
        self._eos = e.new(1)

    # ---- reference properties (modeling_vibevoice_streaming_inference.py:119-141) ----
    @property
    def noise_scheduler(self):
        hcfg = self.config_dict["diffusion_head_config"]
        return _SchedulerView({"num_train_timesteps": hcfg.get("ddpm_num_steps", 1000), "beta_schedule": hcfg.get("ddpm_beta_schedule", "cosine"),
                               "prediction_type": hcfg.get("prediction_type", "v_prediction")}, self.ddpm_inference_steps)

    prediction_head = property(lambda self: self.model.prediction_head)
    acoustic_tokenizer = property(lambda self: self.model.acoustic_tokenizer)
    acoustic_connector = property(lambda self: self.model.acoustic_connector)

    def parameters(self):
        yield self._param_placeholder

    def to(self, *a, **k):
        return self

    @classmethod
    def from_pretrained(cls, path, torch_dtype=torch.bfloat16, device_map=None, attn_implementation=None, **runtime):
        """config.json + *.safetensors of a VibeVoice-Realtime / Streaming-0.5B checkpoint directory, with the reference's
        keyword arguments (demo/streaming_inference_from_file.py:244-276)."""
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
        m = cls.from_state_dict(config, it(), torch_dtype or torch.bfloat16, device, attn_implementation=attn_implementation, **runtime)
        m.source_path = path
        return m

    @classmethod
    def from_state_dict(cls, config: dict, state_dict, model_dtype=torch.bfloat16, device=None, attn_implementation=None, **runtime):
        n_tts = config["tts_backbone_num_hidden_layers"]
        runtime.setdefault("n_slots", 2)
        ecfg = engine_config_from_reference(dict(config, semantic_tokenizer_config=None), tts_layers=n_tts,
                                            has_acoustic_encoder=False, **runtime)
        eng = Engine(ecfg, device)
        n_lm = ecfg.lm_layers - n_tts
        exp = eng.expected_weights()
        scaling = bias = None
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict
        for k, v in items:
            if k == "model.speech_scaling_factor":
                scaling = float(v)
            elif k == "model.speech_bias_factor":
                bias = float(v)
            else:
                name = map_streaming_param_name(k, n_lm)
                if name is not None and name in exp:
                    eng.upload(name, v)
        miss = eng.missing_weights()
        if miss:
            raise RuntimeError(f"checkpoint is missing {len(miss)} parameters, e.g. {miss[:4]}")
        m = cls(config, eng, model_dtype, attn_implementation=attn_implementation)
        if scaling is not None and bias is not None:
            m.set_speech_factors(scaling, bias)
        return m

    def fork(self, **runtime):
        """A second session over THIS model's weights (Engine.fork -> vv_create_shared: one copy in HBM), with its own engine context
        (KV caches, decoder state, graphs, stream): generate() on the fork and on the original may run at the same time from two host
        threads -- the reference class is one session at a time (B = 1 only, modeling_vibevoice_streaming_inference.py:511)."""
        m = type(self)(self.config_dict, self.engine.fork(**runtime), self.dtype, self.requested_attn_implementation)
        m.set_speech_factors(self.speech_scaling_factor, self.speech_bias_factor)
        m.set_ddpm_inference_steps(self.ddpm_inference_steps)
        return m

    def set_speech_factors(self, scaling, bias):
        self.speech_scaling_factor, self.speech_bias_factor = float(scaling), float(bias)
        self.engine.set_speech_factors(scaling, bias)

    def eval(self):
        return self

    def set_ddpm_inference_steps(self, num_steps=None):
        self.ddpm_inference_steps = num_steps or self.config_dict["diffusion_head_config"].get("ddpm_num_inference_steps", 20)

    def _import_branch(self, out, cache, l0):
        n = 0
        for j, (k, v) in enumerate(_kv_layers(out.past_key_values)):
            k = k.to(self.device)
            v = v.to(self.device)
            self.engine.kv_import(cache, l0 + j, k[0], v[0])
            n = k.shape[2]
        return n

    @torch.no_grad()
    def generate(self, inputs=None, generation_config=None, audio_streamer=None, tts_text_ids=None,
                 return_speech=True, cfg_scale=1.0, stop_check_fn: Optional[Callable[[], bool]] = None, **kwargs):
        e = self.engine
        all_pre = kwargs.pop("all_prefilled_outputs")
        # the processor's prompt ids (vibevoice_streaming_processor.py:180-325): the preset's caches already hold them; the
        # reference returns them in front of the generated ids (`sequences=tts_lm_input_ids`, :722)
        prompt_ids = kwargs.pop("tts_lm_input_ids", None)
        noise_fn = kwargs.pop("_noise_fn", None)
        trace = kwargs.pop("_trace", None)
        teacher = kwargs.pop("_teacher_latents", None)   # test hook (teacher-forced per step): frame -> latent [1, L]
        marks = kwargs.pop("_marks", None)            # bench hook: first-audio timestamp
        verbose = kwargs.get("verbose", False)
        tts_text_ids = tts_text_ids.reshape(-1).cpu()
        H = e.cfg.lm_hidden
        e.set_num_steps(self.ddpm_inference_steps, t_cast_bf16=(self.dtype == torch.bfloat16))
        with torch.cuda.stream(e.stream):
            e.codec_reset(0)
            lm_len = self._import_branch(all_pre["lm"], LM_CACHE, 0)
            tts_len = self._import_branch(all_pre["tts_lm"], TTS_CACHE, self.n_lm)
            neg_len = self._import_branch(all_pre["neg_tts_lm"], NEG_TTS_CACHE, self.n_lm)
            self._cond[0].copy_(all_pre["tts_lm"].last_hidden_state[0, -1].to(self.device, torch.float32))
            self._cond[1].copy_(all_pre["neg_tts_lm"].last_hidden_state[0, -1].to(self.device, torch.float32))
            if kwargs.get("max_new_tokens", None) is None:
                max_length = self.max_position_embeddings
            else:
                max_length = tts_len + kwargs["max_new_tokens"]
            max_length = min(max_length, e.max_ctx)
            n_tok = tts_len
            finished = False
            reach_max = False
            chunks = []
            w = 0
            frame = 0
            seq_tail = []
            while True:
                if stop_check_fn is not None and stop_check_fn():
                    if audio_streamer is not None:
                        audio_streamer.end()
                    break
                if finished:
                    break
                cur = tts_text_ids[w * TTS_TEXT_WINDOW_SIZE:(w + 1) * TTS_TEXT_WINDOW_SIZE].tolist()
                w += 1
                if cur:
                    n_tok += len(cur)
                    seq_tail.extend(cur)        # the reference concatenates the window's ids BEFORE it checks the length cap (:573-582):
                    if n_tok > max_length:      # a cap that falls on a text window leaves that window's ids in `sequences`
                        reach_max = True
                        break
                    k = len(cur)
                    e.embed(cur, self._x)
                    e.lm_forward_range([(LM_CACHE, lm_len + j) for j in range(k)], self._x, self._h, 0, self.n_lm, False)
                    lm_len += k
                    e.add_type_embedding(k, self._h, 1, self._x)
                    e.lm_forward_range([(TTS_CACHE, tts_len + j) for j in range(k)], self._x, self._hid,
                                       self.n_lm, self.n_lm + self.n_tts, True)
                    tts_len += k
                    self._cond[0].copy_(self._hid[k - 1])
                for i in range(TTS_SPEECH_WINDOW_SIZE):
                    nz = noise_fn(frame, 2) if noise_fn is not None else torch.randn(2, e.cfg.latent_dim)
                    self._noise[0].copy_(nz[0].to(torch.float32))
                    e.diffusion_sample(1, self._cond, self._noise, cfg_scale, self._latent)
                    own_latent = None
                    if teacher is not None:
                        # the frame's own latent is what gets compared; everything downstream of it (decoder, connector,
                        # TTS-LM step) consumes the oracle's, so a bf16-mode run is held step by step without AR compounding
                        own_latent = self._latent.cpu().clone()
                        tl = teacher(frame)
                        if tl is not None:
                            self._latent.copy_(tl.to(self.device, torch.float32).reshape(self._latent.shape))
                    e.codec_decode(0, self._latent, self._audio[0])
                    chunk = self._audio.clone()
                    if marks is not None and "first_audio" not in marks:
                        e.sync()
                        import time as _t
                        marks["first_audio"] = _t.perf_counter()
                    if not finished:
                        chunks.append(chunk)
                    if audio_streamer is not None:
                        audio_streamer.put(chunk[:, None, :].to(self.dtype), torch.tensor([0]))
                    e.connect(1, self._latent, None, self._emb)
                    frame += 1
                    n_tok += 1
                    seq_tail.append(1)
                    if n_tok > max_length:
                        break
                    # positive and negative TTS-LM rows consume the same embedding: one weight pass
                    e.add_type_embedding(1, self._emb, 0, self._x)
                    self._x[1].copy_(self._x[0])
                    e.lm_forward_range([(TTS_CACHE, tts_len), (NEG_TTS_CACHE, neg_len)], self._x, self._hid,
                                       self.n_lm, self.n_lm + self.n_tts, True)
                    tts_len += 1
                    neg_len += 1
                    self._cond.copy_(self._hid[:2])
                    e.eos_logit(1, self._hid, self._eos)
                    logit = float(self._eos.cpu()[0])
                    if trace is not None:
                        trace.append({"latent": own_latent if own_latent is not None else self._latent.cpu().clone(),
                                      "tts_last": self._hid[0].cpu().clone(), "eos": logit})
                    if torch.sigmoid(torch.tensor(logit)).item() > 0.5:
                        finished = True
                        if audio_streamer is not None:
                            audio_streamer.end(torch.tensor([0]))
                if n_tok > max_length:
                    if not finished:
                        reach_max = True
                    break
            if audio_streamer is not None:
                audio_streamer.end()
            audio = torch.cat(chunks, dim=-1).to(self.dtype) if chunks else None
        e.sync()
        seq = torch.tensor([seq_tail], dtype=torch.long)
        if prompt_ids is not None:
            seq = torch.cat([prompt_ids.reshape(1, -1).to("cpu", torch.long), seq], dim=-1)
        return VibeVoiceGenerationOutput(sequences=seq,
                                         speech_outputs=[audio] if return_speech else None,
                                         reach_max_step_sample=torch.tensor([reach_max]))
