"""Host-side mirror of the reference's inference class for the hot path.

`VibeVoiceForConditionalGenerationInference` here keeps the reference's public
surface for this path (vibevoice/modular/modeling_vibevoice_inference.py):

    from_pretrained(path, torch_dtype=..., device_map=..., attn_implementation=...)   demo/inference_from_file.py:297-317
    .model.{language_model, prediction_head, acoustic_connector, semantic_connector, ...}   :87-117, lora_loading.py:88-169
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

Beyond the reference: `generate_continuous()` keeps up to n_slots
utterances in flight on one GPU and admits the next queued utterance into a slot the
moment its occupant finishes, without draining the batch; every utterance comes out
exactly as generate() would have produced it alone (the reference's batched loop has
no cross-sample arithmetic, :393-394,549,573,594).
"""
import json
import contextlib
import os
import time
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from .engine import Engine, EngineConfig, map_param_name

_WARNED_QUEUED_RNG = False


@contextlib.contextmanager
def _end_streamer_on_error(streamer):
    """an exception out of the generate loops (an engine error, a failed request) must not leave an AudioStreamer consumer blocked in
    get() forever: every stream is ended before the exception travels on"""
    try:
        yield
    except BaseException:
        if streamer is not None:
            try:
                streamer.end()
            except Exception:
                pass
        raise


MAX_BATCH = 8          # rows of one diffusion-head pass (2 per utterance, 16-row MFMA tile); vv_diffusion_sample's limit


@dataclass
class BenchHooks:
    """What a measurement harness (bench.py) may hang on generate() / generate_continuous() through `_bench_hooks=`: nothing
    in here changes what is computed for the positions generate() itself fills.
      step_callback(step)      called at the top of every loop iteration (timed-region marks, profiler windows, host-delay probe)
      kv_start, kv_fill_fn     long-context decode measurement: after the real prompt prefill the positive cache is declared
                               kv_start positions long and kv_fill_fn(engine, cache, p0, p1) writes the positions in between"""
    step_callback: Optional[Callable[[int], None]] = None
    kv_start: int = 0
    kv_fill_fn: Optional[Callable] = None


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


# ---------------------------------------------------------------------- the attribute surface callers read
class _Ns(dict):
    """dict with attribute access (config objects: `model.config.decoder_config.hidden_size`)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return _Ns(v) if isinstance(v, dict) and not isinstance(v, _Ns) else v

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return dict(self)


class WeightHandle:
    """What `model.model.<component>` is on the HIP path: the reference keeps an nn.Module there; here the weights live
    repacked inside the engine, and this handle is the write side of that snapshot.  It supports what the reference's
    callers do with the attribute (lora_loading.py:71-84,112-131,163-169; demo/inference_from_file.py:367-368):
    `load_state_dict(sd, strict=False)` (uploads -> the engine re-packs), `.to(device)`, `.eval()`, `.config`,
    `.device`, `parameters()` (one placeholder tensor, enough for `next(model.parameters()).device`)."""

    def __init__(self, owner, ref_prefix: str, config: Optional[dict] = None, to_engine=None, to_reference=None):
        self._owner = owner
        self._ref_prefix = ref_prefix
        self.config = _Ns(config or {})
        # reference state_dict key <-> engine parameter name (the streaming model splits its layers differently)
        self._to_engine = to_engine or map_param_name
        self._to_reference = to_reference or _engine_name_to_reference

    @property
    def device(self):
        return self._owner.device

    @property
    def dtype(self):
        return self._owner.dtype

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def parameters(self):
        yield self._owner._param_placeholder

    def expected_keys(self):
        """state_dict keys (relative to this component) the engine holds"""
        out = []
        for name in self._owner.engine.expected_weights():
            key = self._to_reference(name)
            if key is not None and key.startswith(self._ref_prefix):
                out.append(key[len(self._ref_prefix):])
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        """Upload a (partial) state dict of this component; returns (missing, unexpected) like nn.Module."""
        eng = self._owner.engine
        exp = eng.expected_weights()
        seen, unexpected = set(), []
        for k, v in state_dict.items():
            name = self._to_engine(self._ref_prefix + k)
            if name is None or name not in exp:
                unexpected.append(k)
                continue
            eng.upload(name, v)
            seen.add(k)
        missing = [k for k in self.expected_keys() if k not in seen]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict({self._ref_prefix}): missing {missing[:4]}, unexpected {unexpected[:4]}")
        import collections
        return collections.namedtuple("IncompatibleKeys", "missing_keys unexpected_keys")(missing, unexpected)


_PREFIX_BACK = tuple((a, b) for a, b in (
    ("model.language_model.", "lm."), ("model.prediction_head.", "head."),
    ("model.acoustic_tokenizer.decoder.", "dec."), ("model.acoustic_tokenizer.encoder.", "aenc."),
    ("model.semantic_tokenizer.encoder.", "senc."), ("model.acoustic_connector.", "ac_conn."),
    ("model.semantic_connector.", "sem_conn.")))


class _SchedulerView:
    """`noise_scheduler` as callers use it (modeling_vibevoice_inference.py:91-93; demo/gradio_demo.py:142-146): `.config` (the
    DPMSolverMultistepScheduler init arguments the model class passes, modeling_vibevoice.py:138-142, plus that class's
    defaults), `.num_inference_steps`, `.timesteps`, and `.from_config(config, **overrides)` -> a new view, which the gradio
    demo assigns back to `model.model.noise_scheduler` to switch the solver to 'sde-dpmsolver++'.  The arithmetic itself
    lives in the engine's coefficient table (vibevoice_amd/schedule.py)."""

    DEFAULTS = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="cosine", trained_betas=None,
                    solver_order=2, prediction_type="v_prediction", thresholding=False, dynamic_thresholding_ratio=0.995,
                    sample_max_value=1.0, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
                    euler_at_final=False, use_karras_sigmas=False, use_lu_lambdas=False, final_sigmas_type="zero",
                    lambda_min_clipped=-float("inf"), variance_type=None, timestep_spacing="linspace", steps_offset=0,
                    rescale_betas_zero_snr=False)

    def __init__(self, config: dict, num_inference_steps: Optional[int] = None):
        self.config = _Ns(dict(self.DEFAULTS, **{k: v for k, v in dict(config).items() if k in self.DEFAULTS}))
        self.num_inference_steps = num_inference_steps

    def from_config(self, config, **kwargs):
        return _SchedulerView(dict(dict(config), **kwargs), self.num_inference_steps)

    @property
    def timesteps(self):
        from . import schedule as _schedule
        tv, _ = _schedule.make_table(self.num_inference_steps or 20, False)
        return torch.from_numpy(np.asarray(tv)).long()

    def check_supported(self):
        """the configurations the HIP sampler implements; anything else must fail loudly, not run a different solver"""
        from . import schedule as _schedule
        c = self.config
        want = dict(num_train_timesteps=1000, trained_betas=None, solver_order=2, prediction_type="v_prediction", thresholding=False,
                    solver_type="midpoint", lower_order_final=True, euler_at_final=False, use_karras_sigmas=False,
                    use_lu_lambdas=False, final_sigmas_type="zero", timestep_spacing="linspace", steps_offset=0,
                    rescale_betas_zero_snr=False)
        bad = {k: c[k] for k, v in want.items() if c[k] != v}
        if c["beta_schedule"] not in ("cosine", "squaredcos_cap_v2"):          # the same Glide cosine betas (dpm_solver.py:240-242)
            bad["beta_schedule"] = c["beta_schedule"]
        if c["algorithm_type"] not in _schedule.ALGORITHMS:
            bad["algorithm_type"] = c["algorithm_type"]
        if bad:
            raise NotImplementedError(f"noise scheduler configuration not implemented by the HIP sampler: {bad}")


def _engine_name_to_reference(name: str):
    for a, b in _PREFIX_BACK:
        if name.startswith(b):
            return a + name[len(b):]
    return None


class _ModelNamespace:
    """`model.model` (VibeVoiceModel, modeling_vibevoice.py:108-146) as far as callers read it."""

    def __init__(self, owner, config: dict, attn_implementation: str):
        d = dict(config.get("decoder_config", {}))
        d["_attn_implementation"] = attn_implementation
        self.language_model = WeightHandle(owner, "model.language_model.", d)
        self.prediction_head = WeightHandle(owner, "model.prediction_head.", config.get("diffusion_head_config"))
        self.acoustic_tokenizer = WeightHandle(owner, "model.acoustic_tokenizer.", config.get("acoustic_tokenizer_config"))
        self.acoustic_connector = WeightHandle(owner, "model.acoustic_connector.")
        if config.get("semantic_tokenizer_config") is not None:
            self.semantic_tokenizer = WeightHandle(owner, "model.semantic_tokenizer.", config.get("semantic_tokenizer_config"))
            self.semantic_connector = WeightHandle(owner, "model.semantic_connector.")
        self._owner = owner

    @property
    def speech_scaling_factor(self):
        return torch.tensor(self._owner._scaling)

    @property
    def speech_bias_factor(self):
        return torch.tensor(self._owner._bias)

    @property
    def noise_scheduler(self):
        return self._owner.noise_scheduler

    @noise_scheduler.setter
    def noise_scheduler(self, view):
        # demo/gradio_demo.py:142-146: model.model.noise_scheduler = model.model.noise_scheduler.from_config(config, algorithm_type=...)
        if not isinstance(view, _SchedulerView):
            raise TypeError("model.model.noise_scheduler takes the object noise_scheduler.from_config(...) returns")
        view.check_supported()
        self._owner._sched_cfg = dict(view.config)


class _LaneStreamer:
    """one lane's view of the caller's AudioStreamer in generate_interleaved: the lane's request j is the caller's sample idx[j]; the
    lane's closing end() ends only the lane's own samples (the caller's streamer is closed once, after every lane has finished)"""

    def __init__(self, inner, idx):
        self.inner, self.idx = inner, [int(i) for i in idx]
        self._ended = set()

    @property
    def finished_flags(self):
        ff = getattr(self.inner, "finished_flags", None)
        return [bool(ff[i]) for i in self.idx] if ff is not None else [False] * len(self.idx)

    def put(self, chunks, sample_indices):
        self.inner.put(chunks, torch.tensor([self.idx[int(i)] for i in sample_indices.tolist()]))

    def end(self, sample_indices=None):
        ids = self.idx if sample_indices is None else [self.idx[int(i)] for i in sample_indices.tolist()]
        ids = [i for i in ids if i not in self._ended]          # one end per sample, as one generate() call gives
        self._ended.update(ids)
        if ids:
            self.inner.end(torch.tensor(ids))


class _Utt:
    """One utterance in flight: its engine slot (KV caches 2*slot / 2*slot+1, codec states), lengths and outputs."""
    __slots__ = ("idx", "slot", "ids", "seq_len0", "init_len", "max_length", "max_steps", "max_step_sample", "step", "pos_len",
                 "neg_len", "have_embeds", "finished", "reach_max", "tokens", "chunks", "last", "forced", "noise_fn", "req",
                 "t_admit", "t_done", "neg_book")

    def __init__(self, idx, slot, ids, seq_len0, max_length, max_length_times, start_id):
        self.idx, self.slot, self.ids = idx, slot, ids
        self.seq_len0 = seq_len0                      # width of the (padded) prompt batch this utterance arrived in
        self.init_len = len(ids)
        self.max_length = max_length
        self.max_steps = min(max_length - seq_len0, int(max_length_times * seq_len0))               # :421 (loop length)
        self.max_step_sample = min(max_length - self.init_len, int(max_length_times * self.init_len))    # :422
        self.step = 0
        self.pos_len = self.neg_len = 0
        self.have_embeds = False
        self.finished = self.reach_max = False
        self.tokens, self.chunks = [], []
        self.last = ids[-1] if ids else start_id
        self.forced = self.noise_fn = self.req = None
        self.t_admit = self.t_done = None
        # the reference's bookkeeping of this row's negative cache, without the tensors: [attention mask incl. the next token's slot,
        # entries ever appended, corrections so far (correct_cnt)] -- to recognise the one correction that keeps THIS step's entry (vv_kv_move)
        self.neg_book = [[1], 0, 0]


class VibeVoiceForConditionalGenerationInference:
    """Drop-in for the reference class on the generate() path, backed by libvvhip.so."""

    def __init__(self, config: dict, engine: Engine, model_dtype=torch.bfloat16, attn_implementation: Optional[str] = None):
        self.config_dict = config
        self.config = _Ns(config)
        self.engine = engine
        self.dtype = model_dtype
        self.device = engine.device
        self.ddpm_inference_steps = config["diffusion_head_config"].get("ddpm_num_inference_steps", 20)
        self.max_position_embeddings = config["decoder_config"].get("max_position_embeddings", 32768)
        self.acoustic_vae_dim = config.get("acoustic_vae_dim", 64)
        self.fix_std = config["acoustic_tokenizer_config"].get("fix_std", 0.5)
        self.std_dist_type = config["acoustic_tokenizer_config"].get("std_dist_type", "gaussian")
        self._scaling = float("nan")
        self._bias = float("nan")
        self._valid_key = None
        # what the attention really is on this path; the value the caller asked for is kept beside it
        self.requested_attn_implementation = attn_implementation
        self._param_placeholder = torch.empty(0, dtype=model_dtype, device=self.device)
        hcfg = config["diffusion_head_config"]
        # the scheduler the model class builds (modeling_vibevoice.py:138-142)
        self._sched_cfg = dict(_SchedulerView({"num_train_timesteps": hcfg.get("ddpm_num_steps", 1000),
                                               "beta_schedule": hcfg.get("ddpm_beta_schedule", "cosine"),
                                               "prediction_type": hcfg.get("prediction_type", "v_prediction")}).config)
        self._sde_flat = None                    # per-step variance noise of the stochastic solver, [64 * MAX_BATCH * latent]
        self.model = _ModelNamespace(self, config, "vvhip_mfma_flash_decoding_gfx950")
        H = engine.cfg.lm_hidden
        e = engine
        NB = MAX_BATCH
        self._x_in = e.new(2 * NB, H)
        self._hidden = e.new(2 * NB, H)
        self._hid_fresh = e.new(max(1, engine.cfg.n_slots), H)       # last prompt row of a freshly prefilled utterance, per slot
        self._logits = e.new(2 * NB * 16)                            # dense [n][n_valid] blocks, as vv_lm_logits writes them
        self._cond = e.new(2 * NB, H)
        self._noise = e.new(NB, engine.cfg.latent_dim)
        self._latent = e.new(NB, engine.cfg.latent_dim)
        self._audio = e.new(NB, engine.cfg.hop)
        self._sem = e.new(NB, max(1, engine.cfg.sem_dim))
        self._emb_out = e.new(NB, H)
        self._start_emb = e.new(1, H)
        self._neg_hidden = e.new(NB, H)
        self._nxt_x = e.new(NB, H)
        self._tmp_emb = e.new(NB, H)
        # pinned host staging: the logits come back without a blocking copy, noise goes out without one
        self._logits_pin = torch.empty(2 * NB * 16, dtype=torch.float32).pin_memory()
        self._noise_pin = [torch.empty(NB, engine.cfg.latent_dim, dtype=torch.float32).pin_memory() for _ in range(4)]
        self._noise_i = 0
        self._lg_event = torch.cuda.Event()
        self._fork_ev = torch.cuda.Event()
        self._join_ev = torch.cuda.Event()
        self._side_streams = [torch.cuda.Stream(device=self.device) for _ in range(min(NB, engine.cfg.n_slots))] if engine.cfg.n_slots > 1 else []
        self._audio_blocks = []                                       # output frames, FRAME_BLOCK steps per block (no per-step allocation)
        self.frame_block = 64
        # plain attributes (tests flip them): per-utterance tokenizer chains on forked streams; several utterances' chains of a step
        # as ONE engine call (vv_codec_chain_batch); the sampler enqueued speculatively behind the LM pass
        self.concurrent_codecs = True
        self.batched_codecs = hasattr(engine, "codec_chain_batch")
        # one utterance's decode -> semantic re-encode as ONE engine call (one captured sequence instead of two: a graph-to-graph transition
        # costs ~7 us against ~1.5 us between two kernels of one graph); VVHIP_CHAIN_SINGLE=0: the two calls (the A/B of round 6)
        self.chain_single = os.environ.get("VVHIP_CHAIN_SINGLE", "1") != "0"
        self.speculate_sampling = True
        self.last_stats = {}

    # ------------------------------------------------------------------ construction
    @classmethod
    def from_state_dict(cls, config: dict, state_dict, model_dtype=torch.bfloat16, device=None, attn_implementation=None, **runtime):
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
        m = cls(config, eng, model_dtype, attn_implementation=attn_implementation)
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
        # the drop-in entry point: any batch generate() can take (MAX_BATCH utterances) works without an extra argument; the KV
        # caches of 8 slots at the model's full context are 30 GB for the 7B model -- a tenth of the HBM
        runtime.setdefault("n_slots", MAX_BATCH)
        do_warm = runtime.pop("warmup", True)
        m = cls.from_state_dict(config, it(), torch_dtype or torch.bfloat16, device, attn_implementation=attn_implementation, **runtime)
        m.source_path = path
        if do_warm and hasattr(m.engine, "acoustic_encode") and getattr(m.engine, "lib", None) is not None:
            try:
                m.warmup()
            except Exception as ex:         # a failed warm-up costs the first request its latency, never the model load --
                import warnings             # unless the device itself is gone: then the load fails here, with the cause
                try:
                    torch.cuda.synchronize(m.device)
                    m.engine.sync()
                except Exception as dead:
                    raise RuntimeError(f"vibevoice_amd: warm-up failed ({ex!r}) and the device does not answer any more") from dead
                warnings.warn(f"vibevoice_amd: warm-up failed ({ex!r}); the first generate() will pay the cold-start costs")

        def base_tensor(key):          # lazy access to the checkpoint's own tensors (LoRA merge: vibevoice_amd/lora.py)
            for fn in files:
                with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as sf:
                    if key in sf.keys():
                        return sf.get_tensor(key)
            raise KeyError(key)
        m.base_tensor = base_tensor
        return m

    @classmethod
    def from_reference(cls, live_model, device=None, **runtime):
        """Snapshot a live reference `VibeVoiceForConditionalGenerationInference` (an nn.Module, e.g. one that already carries
        merged adapters) into a HIP-path model: its config.to_dict() + state_dict() + scalar speech factors."""
        cfg = live_model.config.to_dict() if hasattr(live_model.config, "to_dict") else dict(live_model.config)
        sd = live_model.state_dict()
        dt = next(iter(live_model.parameters())).dtype
        runtime.setdefault("n_slots", MAX_BATCH)
        m = cls.from_state_dict(cfg, sd, dt, device, **runtime)
        m.base_tensor = lambda key: sd[key].detach().cpu()
        steps = getattr(live_model, "ddpm_inference_steps", None)
        if steps:
            m.set_ddpm_inference_steps(steps)
        return m

    # ---- reference properties (:87-117) ----
    @property
    def speech_scaling_factor(self):
        return self._scaling

    @property
    def speech_bias_factor(self):
        return self._bias

    @property
    def noise_scheduler(self):
        return _SchedulerView(self._sched_cfg, self.ddpm_inference_steps)

    prediction_head = property(lambda self: self.model.prediction_head)
    acoustic_tokenizer = property(lambda self: self.model.acoustic_tokenizer)
    semantic_tokenizer = property(lambda self: getattr(self.model, "semantic_tokenizer", None))
    acoustic_connector = property(lambda self: self.model.acoustic_connector)
    semantic_connector = property(lambda self: getattr(self.model, "semantic_connector", None))

    def parameters(self):
        yield self._param_placeholder

    def to(self, *a, **k):
        return self

    def tie_weights(self):
        """the engine reads lm_head rows from embed_tokens when no lm_head.weight was uploaded (vv_set_valid_tokens)"""
        self._valid_key = None

    def set_speech_factors(self, scaling, bias):
        self._scaling = float(scaling)
        self._bias = float(bias)
        self.engine.set_speech_factors(scaling, bias)

    def eval(self):
        return self

    def set_ddpm_inference_steps(self, num_steps=None):
        self.ddpm_inference_steps = num_steps or self.config_dict["diffusion_head_config"].get("ddpm_num_inference_steps", 20)

    def warmup(self, prompt_rows: Optional[Sequence[int]] = None, voice_frames: int = 75):
        """Touch everything the first request needs, so that its time to first audio is a warm process's: the first launch of
        a kernel pays for its code object and launch attributes, and the first use of a torch op on the device (the Gaussian
        draw of the voice-prompt latents, the masked scatter of the speech rows, ...) loads its module -- measured on the 7B
        north-star request: voice-prompt encode 0.098 s cold against 0.03-0.04 s warm.  Runs (a) one LM prompt pass per entry
        of `prompt_rows` (default: a full max_rows pass and a ragged remainder) and (b) a REAL generate() on a synthetic
        one-speaker request (`voice_frames`-frame silent voice prompt, short prompt, three forced frames, explicit noise), i.e.
        the request path itself: encoder, connectors, prompt scatter, decode steps under the graph keys generate() uses,
        sampler, both tokenizers, the streamer hand-off.  Draws nothing from any RNG (noise is passed in; the device
        generator's state is saved and restored around it) and leaves no state behind: caches are overwritten by the next
        prefill, codec states are zeroed at the start of every generate()."""
        import types
        e = self.engine
        H, hop, L = e.cfg.lm_hidden, e.cfg.hop, e.cfg.latent_dim
        R = e.cfg.max_rows
        cap = max(1, min(R, e.max_ctx - 16))                    # never more rows than one LM launch takes (max_rows may be 2..7)
        if prompt_rows is None:
            prompt_rows = sorted({cap, max(1, min(cap, cap // 3 + 5))}, reverse=True)
        rng = torch.cuda.get_rng_state(self.device)
        saved_valid, saved_stats = getattr(e, "_valid_ids", None), dict(self.last_stats)
        try:
            with torch.cuda.stream(e.stream):
                for n in prompt_rows:
                    n = int(max(1, min(n, cap)))
                    x = torch.zeros(n, H, dtype=torch.float32, device=self.device)
                    hid = torch.empty_like(x)
                    if hasattr(e, "lm_forward_span"):
                        e.lm_forward_span(0, 0, n, x, hid)
                    else:
                        e.lm_forward([(0, j) for j in range(n)], x, hid)
                    del x, hid
            e.sync()
            has_voice = bool(getattr(e.cfg, "has_acoustic_encoder", False)) and voice_frames > 0
            vf = voice_frames if has_voice else 0
            n_prompt = int(min(cap, e.max_ctx - 16, vf + 24))
            vf = max(0, min(vf, n_prompt - 8))
            tok = types.SimpleNamespace(speech_start_id=0, speech_end_id=1, speech_diffusion_id=2, eos_token_id=3, bos_token_id=None)
            ids = torch.full((1, n_prompt), 4 % e.cfg.lm_vocab, dtype=torch.long)
            sim = torch.zeros(1, n_prompt, dtype=torch.bool)
            kw = {}
            if vf > 0:
                ids[0, 4:4 + vf] = tok.speech_diffusion_id
                sim[0, 4:4 + vf] = True
                kw = dict(speech_tensors=torch.zeros(1, vf * hop), speech_masks=torch.ones(1, vf, dtype=torch.bool), speech_input_mask=sim,
                          _prefill_noise=(torch.zeros(1), torch.zeros(1, vf, L)))
            ids[0, -1] = tok.speech_start_id
            D = tok.speech_diffusion_id
            scaled = not (self._scaling != self._scaling)              # speech factors may not be set yet (NaN): use neutral ones here
            if not scaled:
                self.engine.set_speech_factors(1.0, 0.0)
                self._scaling, self._bias = 1.0, 0.0
            # through an AudioStreamer, as a serving request goes: its copy stream exists and its pinned ring buffers sit in the
            # streamer module's pool afterwards, so the first real request's first chunk does not wait for pinned allocations
            from .streamer import AudioStreamer
            st = AudioStreamer(batch_size=1)
            try:
                self.generate(input_ids=ids, attention_mask=torch.ones_like(ids), tokenizer=tok, cfg_scale=1.3,
                              generation_config={"do_sample": False}, max_new_tokens=4, show_progress_bar=False, audio_streamer=st,
                              _forced_tokens=[[D, D, D, tok.eos_token_id]], _noise_fn=lambda step, n2: torch.zeros(n2, L), **kw)
            finally:
                st.close()
            if not scaled:
                self._scaling = self._bias = float("nan")
            self._valid_key = None
        finally:
            torch.cuda.set_rng_state(rng, self.device)
            if saved_valid:                                       # the synthetic request's control-token ids must not outlive it
                e.set_valid_tokens(saved_valid)
            elif hasattr(e, "_valid_ids"):
                e._valid_ids = None
            self.last_stats = saved_stats
        e.sync()

    # ------------------------------------------------------------------ helpers
    def _embed_ids(self, ids: List[int], out: torch.Tensor):
        ch = getattr(self.engine, "embed_chunk", 64)         # one call per prompt chunk, not per 64 ids
        for i0 in range(0, len(ids), ch):
            self.engine.embed(ids[i0:i0 + ch], out[i0:])

    def _stage_noise(self, nz, n):
        """noise rows -> device without blocking the host (pageable H2D copies are synchronous with the stream)"""
        if nz.is_cuda:
            self._noise[:n].copy_(nz[:n].to(torch.float32))
            return
        pin = self._noise_pin[self._noise_i]
        self._noise_i = (self._noise_i + 1) % len(self._noise_pin)
        pin[:n].copy_(nz[:n])
        self._noise[:n].copy_(pin[:n], non_blocking=True)

    def _process_speech_inputs(self, speech_tensors, speech_masks, prefill_noise=None, dev_gen=None):
        """_process_speech_inputs (:149-163): encode voice prompts, sample, scale, connect.
        dev_gen: a lane's own device generator (generate_interleaved); None = the device's global generator, as the reference.
        Every host -> device copy of the call (waveform, frame selection, explicit noise) goes out BEFORE the encoder is enqueued:
        a pageable copy blocks the host until the stream has drained, and a boolean-mask gather synchronises to count its rows --
        behind the encoder either one keeps the prompt pass from being enqueued while the encoder runs."""
        e = self.engine
        hop = e.cfg.hop
        n_spk, S = speech_tensors.shape
        frames = S // hop
        valid = None
        if frames * hop != S:
            # the batch tensor does not end on a frame: the engine takes whole frames of zero-padded waveform plus the signal length --
            # the reference right-pads per strided conv LAYER (zeros, not the activations of a zero waveform), which only the last,
            # partial frame's latent can tell (vv_acoustic_encode_ragged)
            pad = (frames + 1) * hop - S
            speech_tensors = torch.nn.functional.pad(speech_tensors, (0, pad))
            frames += 1
            valid = S
        if tuple(speech_masks.shape) != (n_spk, frames):
            # the reference's feats[speech_masks] raises on a mask that does not have the encoder output's shape; a flattened gather
            # would silently pick other rows
            raise ValueError(f"speech_masks has shape {tuple(speech_masks.shape)}; {n_spk} voice samples of {S} samples give "
                             f"({n_spk}, {frames}) frames (ceil(S / {hop}))")
        wav = speech_tensors.to(self.device, torch.float32).contiguous()
        # rows of the [n_spk * frames] encoder output that are real frames (speech_masks is the processor's host data)
        sel_idx = speech_masks.reshape(-1).to(torch.bool).cpu().nonzero().squeeze(1).to(self.device)
        if prefill_noise is not None:
            prefill_noise = tuple(t.to(self.device, torch.float32) for t in prefill_noise)
        mean = e.new(n_spk, frames, e.cfg.latent_dim)
        for i in range(n_spk):
            e.acoustic_encode(frames, wav[i], mean[i], valid_samples=valid)
        if self.std_dist_type == "gaussian":
            if prefill_noise is None:
                # VibeVoiceTokenizerEncoderOutput.sample('gaussian'), modular_vibevoice_tokenizer.py:980-989:
                # two draws from the device generator.  The reference's `mean` is latents.permute(0, 2, 1) (:1085) and its noise is
                # randn_like(mean): the same call on a tensor with the same strides is what stays on the reference's RNG stream
                # (a contiguous draw takes a different generator path).  Pinned by tests/golden/generate_sampled_b1.npz.
                if dev_gen is not None:
                    r1 = torch.randn(n_spk, device=self.device, dtype=torch.float32, generator=dev_gen)
                    r2 = torch.randn(tuple(mean.shape), device=self.device, dtype=torch.float32, generator=dev_gen)
                else:
                    r1 = torch.randn(n_spk, device=self.device, dtype=torch.float32)
                    r2 = torch.randn_like(torch.empty(n_spk, mean.shape[2], mean.shape[1], device=self.device, dtype=torch.float32).permute(0, 2, 1))
            else:
                r1, r2 = prefill_noise
            lat = mean + (r1 * (self.fix_std / 0.8))[:, None, None] * r2
        elif self.std_dist_type == "fix":
            if prefill_noise is not None:
                r2 = prefill_noise[1]
            elif dev_gen is not None:
                r2 = torch.randn(tuple(mean.shape), device=self.device, dtype=torch.float32, generator=dev_gen)
            else:
                r2 = torch.randn_like(torch.empty(n_spk, mean.shape[2], mean.shape[1], device=self.device).permute(0, 2, 1))
            lat = mean + self.fix_std * r2
        else:
            lat = mean
        feats = ((lat + self._bias) * self._scaling).contiguous()
        sel = feats.reshape(n_spk * frames, -1).index_select(0, sel_idx)   # [n_valid, 64]: feats[speech_masks] without the count sync
        out = e.new(sel.shape[0], e.cfg.lm_hidden)
        e.connect(sel.shape[0], sel, None, out)
        return feats, out

    @staticmethod
    def _generation_options(generation_config):
        """generation_config: None, a dict (what every reference caller passes: the reference does
        GenerationConfig(**generation_config), :261-266) or an object with to_dict() (an HF GenerationConfig)."""
        if generation_config is None:
            gc = {}
        elif isinstance(generation_config, dict):
            gc = dict(generation_config)
        elif hasattr(generation_config, "to_dict"):
            gc = {k: v for k, v in generation_config.to_dict().items() if v is not None}
            # an HF GenerationConfig object carries its class defaults (top_k=50, ...): only what differs from them is a request
            for k, dflt in (("top_k", 50), ("top_p", 1.0), ("repetition_penalty", 1.0), ("temperature", 1.0)):
                if gc.get(k) == dflt:
                    gc.pop(k)
        else:
            raise TypeError(f"generation_config must be a dict, None or have to_dict(); got {type(generation_config).__name__}")
        do_sample = bool(gc.get("do_sample", False))
        temperature = float(gc.get("temperature", 1.0) or 1.0)
        # processors this path does not implement: refuse them instead of silently decoding from something else
        for k in ("typical_p", "epsilon_cutoff", "eta_cutoff", "top_h", "no_repeat_ngram_size", "encoder_no_repeat_ngram_size",
                  "bad_words_ids", "num_beams", "num_beam_groups", "penalty_alpha", "diversity_penalty", "sequence_bias",
                  "suppress_tokens", "begin_suppress_tokens", "forced_bos_token_id", "forced_eos_token_id", "min_length",
                  "min_new_tokens", "exponential_decay_length_penalty", "guidance_scale", "encoder_repetition_penalty",
                  "renormalize_logits", "watermarking_config"):
            v = gc.get(k)
            if v not in (None, 0, 0.0, 1, 1.0, [], ()):
                raise NotImplementedError(f"generation_config[{k!r}]={v!r} is not implemented on the HIP path")
        # The processors that act on the FULL vocabulary in front of the valid-token constraint (HF's list, :310-319, then the
        # constraint appended at :416-419): repetition penalty (also without sampling), and with do_sample the warpers
        # temperature -> top-k -> top-p -> min-p.  transformers==4.51.3 (the reference's pin) defaults top_k to 50 whenever
        # do_sample is set: an absent top_k means 50 here too; top_k=0 / None switches it off.  Any of them routes the token
        # choice through vv_lm_logits_full (one pass over the whole lm_head per step) instead of the <= 16 valid rows.
        rep = float(gc.get("repetition_penalty") or 1.0)
        warp = None
        if do_sample:
            top_k = int((gc["top_k"] if "top_k" in gc else 50) or 0)
            top_p = float(gc["top_p"]) if gc.get("top_p") is not None else 1.0
            min_p = float(gc.get("min_p") or 0.0)
            if top_k < 0 or not (0.0 <= top_p <= 1.0) or not (0.0 <= min_p <= 1.0):
                raise ValueError(f"top_k={top_k}, top_p={top_p}, min_p={min_p}: out of range")
            if top_k > 0 or top_p < 1.0 or min_p > 0.0 or rep != 1.0:
                warp = dict(top_k=top_k, top_p=top_p, min_p=min_p, repetition_penalty=rep)
        elif rep != 1.0:
            warp = dict(top_k=0, top_p=1.0, min_p=0.0, repetition_penalty=rep)
        if rep <= 0.0:
            raise ValueError(f"repetition_penalty={rep} must be > 0")
        return do_sample, temperature, warp

    def _full_vocab_scores(self, hidden: torch.Tensor, order, S) -> torch.Tensor:
        """[n, lm_vocab] scores of the given positive rows after the reference's full-vocabulary logits processors, in HF's order
        (generation/utils.py `_get_logits_processor`): repetition penalty over the row's input_ids (left padding, prompt and
        generated tokens), then -- with do_sample -- temperature, top-k, top-p, min-p (min_tokens_to_keep = 1).  The valid-token
        constraint comes after them (the caller)."""
        e, w = self.engine, S["warp"]
        n, V = hidden.shape[0], e.cfg.lm_vocab
        if getattr(self, "_full_logits", None) is None or self._full_logits.numel() < 16 * V:
            self._full_logits = torch.empty(16 * V, dtype=torch.float32, device=self.device)
        hidden = hidden.to(torch.float32).contiguous()
        parts = []
        for i0 in range(0, n, 16):
            k = min(16, n - i0)
            e.lm_logits_full(k, hidden[i0:i0 + k], self._full_logits)
            parts.append(self._full_logits[:k * V].view(k, V).clone())
        scores = torch.cat(parts)                    # (the step loop runs under torch.cuda.stream(engine.stream): one ordered stream)
        if w["repetition_penalty"] != 1.0:
            # RepetitionPenaltyLogitsProcessor: every id present in the row's input_ids (left padding, prompt, generated tokens) is
            # penalised once.  The row's "seen" set is a [V] mask kept on the device for the session and extended by the tokens
            # generated since the last step: O(1) per step, not O(history) (a 90-minute utterance has ~40 K of them).
            pen = w["repetition_penalty"]
            seen = S.setdefault("_seen", {})
            for i, u in enumerate(order):
                ent = seen.get(u.idx)
                if ent is None:
                    base = list(u.ids) + ([S["pad_id"]] if (u.seq_len0 > u.init_len and S["pad_id"] is not None) else [])
                    mask = torch.zeros(V, dtype=torch.bool, device=scores.device)
                    mask[torch.tensor([int(t) for t in base if 0 <= int(t) < V], dtype=torch.long, device=scores.device)] = True
                    ent = seen[u.idx] = [mask, 0]
                fresh_tok = [int(t) for t in u.tokens[ent[1]:] if 0 <= int(t) < V]
                if fresh_tok:
                    ent[0][torch.tensor(fresh_tok, dtype=torch.long, device=scores.device)] = True
                ent[1] = len(u.tokens)
                sc = scores[i]
                scores[i] = torch.where(ent[0], torch.where(sc < 0, sc * pen, sc / pen), sc)
        if S["do_sample"]:
            if S["temperature"] != 1.0:
                scores = scores / S["temperature"]
            if w["top_k"] > 0:
                kth = torch.topk(scores, min(w["top_k"], V))[0][..., -1, None]
                scores = scores.masked_fill(scores < kth, float("-inf"))
            if w["top_p"] < 1.0:
                srt, idx = torch.sort(scores, descending=False)
                remove = srt.softmax(dim=-1).cumsum(dim=-1) <= (1.0 - w["top_p"])
                remove[..., -1:] = False
                scores = scores.masked_fill(remove.scatter(1, idx, remove), float("-inf"))
            if w["min_p"] > 0.0:
                probs = torch.softmax(scores, dim=-1)
                remove = probs < w["min_p"] * probs.amax(dim=-1, keepdim=True)
                idx = torch.argsort(scores, descending=True, dim=-1)
                srt_remove = torch.gather(remove, dim=-1, index=idx)
                srt_remove[..., :1] = False
                scores = scores.masked_fill(srt_remove.scatter(1, idx, srt_remove), float("-inf"))
        return scores

    # ------------------------------------------------------------------ prompt prefill of one utterance
    def _prefill(self, u: _Utt, ids: List[int], speech_rows: Optional[torch.Tensor], speech_pos: Optional[torch.Tensor],
                 kv_start: int = 0, kv_fill_fn=None):
        """LM prefill of one utterance's prompt into KV cache 2*slot; the last row's hidden state -> _hid_fresh[slot]."""
        e = self.engine
        H = e.cfg.lm_hidden
        n = len(ids)
        CH = e.cfg.max_rows          # prompt rows per weight pass
        if n <= CH:
            # one-pass prompts (the common case) reuse two persistent [max_rows, H] buffers: a fresh 2 x 150 MB allocation per
            # request is milliseconds of hipMalloc; every row is overwritten by the embedding lookup, no zero fill needed
            if getattr(self, "_pf_buf", None) is None or self._pf_buf.shape[1] < n:
                with torch.cuda.stream(e.stream):
                    self._pf_buf = torch.empty(2, CH, H, dtype=torch.float32, device=self.device)
            emb = self._pf_buf[0, :n]
            hid = self._pf_buf[1, :n]
        else:
            emb = e.new(n, H)
            hid = e.new(CH, H)
        self._embed_ids(ids, emb)
        if speech_rows is not None and speech_pos is not None and speech_rows.shape[0]:
            if speech_pos.dtype == torch.bool:
                emb[speech_pos] = speech_rows                   # boolean mask: counts its rows on the host (a sync)
            else:
                emb.index_copy_(0, speech_pos, speech_rows)     # int64 positions, uploaded by the caller ahead of the encoder
        timed = os.environ.get("VVHIP_TIME_PREFILL") is not None
        if timed:                    # split the reported prefill time: embedding + voice-row scatter | LM passes
            e.sync(); torch.cuda.current_stream(self.device).synchronize(); t_emb = time.perf_counter()
        for i0 in range(0, n, CH):
            k = min(CH, n - i0)
            if hasattr(e, "lm_forward_span"):
                e.lm_forward_span(2 * u.slot, i0, k, emb[i0:i0 + k], hid)
            else:
                e.lm_forward([(2 * u.slot, i0 + j) for j in range(k)], emb[i0:i0 + k], hid)
        self._hid_fresh[u.slot].copy_(hid[(n - 1) % CH])
        if timed:
            e.sync(); torch.cuda.current_stream(self.device).synchronize()
            self._t_lm_pass = getattr(self, "_t_lm_pass", 0.0) + (time.perf_counter() - t_emb)
        u.pos_len = n
        if kv_start > n:             # bench hook: decode measured at a long context (kv_fill_fn supplies the cache contents)
            if kv_fill_fn is not None:
                timed = os.environ.get("VVHIP_TIME_PREFILL") is not None
                if timed:            # keep the bench-only cache fill out of the reported prompt-prefill time
                    e.sync(); t0 = time.perf_counter()
                kv_fill_fn(e, 2 * u.slot, n, kv_start)
                if timed:
                    e.sync(); torch.cuda.current_stream(self.device).synchronize()
                    self._t_kv_fill = getattr(self, "_t_kv_fill", 0.0) + (time.perf_counter() - t0)
            u.pos_len = kv_start

    def _prefill_checked(self, jobs):
        """_prefill for every (utterance, ids, speech rows, speech positions, kv_start, kv_fill_fn) of `jobs`; prompts long enough for the
        prefill GEMM's K-split round (>= 1024 rows) are waited for and checked before the first frame is enqueued: a lost hand-off
        (vv_check: the engine re-arms itself and says the pass in flight is invalid) is answered by ONE repeat of the prompt passes --
        they only overwrite the same cache positions.  The wait costs nothing measurable: the first frame's launches would have
        queued behind the prompt pass anyway."""
        e = self.engine
        for attempt in (0, 1):
            for j in jobs:
                self._prefill(*j)
            if max(len(j[1]) for j in jobs) < 1024:
                return
            try:
                e.sync()
                return
            except RuntimeError as ex:
                if attempt or "K-split" not in str(ex):
                    raise
                import warnings
                warnings.warn(f"vibevoice_amd: {ex}; repeating the prompt pass once", RuntimeWarning)

    def _block_rows(self, i: int):
        """row i of the frame store: [utterances in flight, hop] fp32, frame_block rows per block"""
        b, r = divmod(i, self.frame_block)
        w = getattr(self, "_frame_w", min(MAX_BATCH, max(1, self.engine.cfg.n_slots)))
        if any(t is not None and t.shape[1] != w for t in self._audio_blocks):
            self._audio_blocks = []
        while len(self._audio_blocks) <= b:
            self._audio_blocks.append(None)
        if self._audio_blocks[b] is None:
            self._audio_blocks[b] = self.engine.new(self.frame_block, w, self.engine.cfg.hop)
        return self._audio_blocks[b][r]

    def _release_frames(self):
        """the frame store holds one row per diffusion iteration of the call that just ended; its outputs have been copied out
        (torch.cat), so only the first block stays allocated for the next call"""
        self._audio_blocks = [t for t in self._audio_blocks if t is not None][:1]
        self._first_row = None

    # ------------------------------------------------------------------ one iteration of the hot loop over the active utterances
    def _iterate(self, S, act: List[_Utt]):
        """modeling_vibevoice_inference.py:466-672 for the utterances in `act`; returns the utterances still live."""
        e = self.engine
        nv, valid_t = S["nv"], S["valid_t"]
        start_id, end_id, diff_id, eos_id = S["start_id"], S["end_id"], S["diff_id"], S["eos_id"]
        cfg_scale, trace, audio_streamer, verbose = S["cfg_scale"], S["trace"], S["audio_streamer"], S["verbose"]
        run = [u for u in act if u.have_embeds]
        fresh = [u for u in act if not u.have_embeds]
        order = run + fresh
        nR, nA = len(run), len(order)
        # ---------------- positive (+ speculative negative) LM pass ----------------
        if run:
            rows = [(2 * u.slot, u.pos_len) for u in run] + [(2 * u.slot + 1, u.neg_len) for u in run]
            self._x_in[nR:2 * nR].copy_(self._x_in[:nR])
            e.lm_forward(rows, self._x_in, self._hidden)
            for u in run:
                u.pos_len += 1
            e.lm_logits(nR, self._hidden, self._logits)
        if fresh:
            if not run and [u.slot for u in fresh] == list(range(len(fresh))):
                e.lm_logits(len(fresh), self._hid_fresh, self._logits)
            else:
                for i, u in enumerate(fresh):
                    e.lm_logits(1, self._hid_fresh[u.slot:u.slot + 1], self._logits[(nR + i) * nv:])
        self._logits_pin.copy_(self._logits, non_blocking=True)      # whole (contiguous) buffer: a true async D2H
        self._lg_event.record(e.stream)
        refresh = S["refresh_negative"]
        if not refresh:
            # refresh_negative=False (:503-516): the negative pass runs at EVERY step for every row on the input the positive pass
            # consumed -- the speculative rows above for `run`; at step 0 (inputs_embeds still None, :395) the lone <speech_start>
            # prompt token
            for i, u in enumerate(fresh):
                e.lm_forward([(2 * u.slot + 1, u.neg_len)], self._start_emb, self._neg_hidden[i:i + 1])

        def pos_hidden(i):                      # hidden state of order[i]'s positive row, [1, H]
            return self._hidden[i:i + 1] if i < nR else self._hid_fresh[order[i].slot:order[i].slot + 1]
        # ---- speculative sampling: a row that has just emitted <speech_diffusion>/<speech_start> almost always
        # emits <speech_diffusion> next.  The sampler (stateless: cond + noise -> latent) is enqueued behind the LM
        # pass BEFORE the host waits for the logits, so the token decision below overlaps GPU work instead of
        # leaving the GPU idle; if the guess is wrong the latent is discarded and the RNG state restored.
        spec_sample, rng_state = False, None
        do_sample = S["do_sample"]
        if (run and not fresh and self.speculate_sampling and not S["sde"]      # a discarded guess would spend device-RNG draws
                and not (do_sample and S["noise_fn"] is None and S["forced"] is None)   # keep the reference's RNG draw order
                and all(u.last in (diff_id, start_id) for u in run)):
            nz = self._draw_noise(S, run)
            if nz is None:
                # the draw is undone if the guess is wrong: on the session's OWN generator when it has one (a lane of
                # generate_interleaved) -- rewinding the process-global generator from one lane would hand another lane draws it has
                # already consumed
                cg = S.get("cpu_gen")
                rng_state = cg.get_state() if cg is not None else torch.get_rng_state()
                nz = torch.randn(2 * nR, e.cfg.latent_dim, generator=cg)
            self._stage_noise(nz, nR)
            # all active rows diffusing, in order: cond rows == [hidden[:nR]; hidden[nR:2nR]]
            e.diffusion_sample(nR, self._hidden, self._noise, cfg_scale, self._latent)
            spec_sample = True
        self._lg_event.synchronize()
        logits = self._logits_pin[:nA * nv].view(nA, nv).clone()
        if trace is not None:
            trace.pos_hidden.append(torch.cat([pos_hidden(i) for i in range(nA)]).cpu())
            if hasattr(trace, "logits"):
                trace.logits.append(logits.clone())               # [rows, n_valid]: the scores the token decision is taken from
        # ---------------- token selection (:488-501) ----------------
        if S["forced"] is not None or any(u.forced is not None for u in order):
            for u in order:
                f = u.forced if u.forced is not None else S["forced"][u.idx]
                u.last = int(f[u.step]) if u.step < len(f) else eos_id
        elif do_sample or S["warp"] is not None:
            # the reference samples torch.multinomial(softmax(scores)) over the FULL vocabulary rows of the WHOLE batch (-inf
            # outside the valid ids, :490-496; finished rows included, their draw is overwritten by eos, :499) on the model's
            # device.  One-sample multinomial spends one exponential variate per (row, category), so the same call on the
            # same-shaped tensor keeps a seeded run on the reference's RNG stream (pinned on CPU by generate_sampled_b1.npz)
            rows_of = S["sample_rows"](order)            # batch mode: every batch row; continuous mode: one row per utterance
            full = torch.full((len(rows_of), e.cfg.lm_vocab), float("-inf"), device=self.device, dtype=torch.float32)
            vt = valid_t.to(self.device)
            full[:, vt] = 0.0                              # finished rows: any proper distribution, the draw is discarded
            if S["warp"] is None:
                lg = self._logits[:nA * nv].view(nA, nv).float() / S["temperature"]
            else:
                # full-vocabulary processors, then the constraint: what survives of the valid ids (-inf where a filter removed one;
                # a row that loses ALL its valid ids has NaN probabilities in the reference too -- torch.multinomial raises)
                scores = self._full_vocab_scores(torch.cat([pos_hidden(i) for i in range(nA)]), order, S)
                lg = scores[:, vt]
                if not bool(torch.isfinite(lg).any(dim=-1).all()):
                    raise RuntimeError("the full-vocabulary logits processors (top_k / top_p / min_p) removed every valid speech token "
                                       "of a row: nothing is left to sample from (the reference fails in torch.multinomial here: "
                                       "'probability tensor contains either `inf`, `nan` or element < 0')")
            for i, u in enumerate(order):
                full[rows_of.index(u.idx), vt] = lg[i]
            if do_sample:
                pick_ids = torch.multinomial(torch.softmax(full, dim=-1), num_samples=1, generator=S.get("dev_gen")).squeeze(1).cpu()
            else:
                pick_ids = torch.argmax(full, dim=-1).cpu()
            for u in order:
                u.last = int(pick_ids[rows_of.index(u.idx)])
        else:
            pick = torch.argmax(logits, dim=-1)
            for i, u in enumerate(order):
                u.last = int(valid_t[pick[i]])
        for u in order:
            u.tokens.append(u.last)
        if trace is not None:
            nxt = torch.full((S["n_rows"],), eos_id, dtype=torch.long)
            for u in order:
                nxt[u.idx] = u.last
            trace.tokens.append(nxt)
        # ---------------- bookkeeping (:518-539) ----------------
        new_eos = [u for u in order if u.last == eos_id]
        if new_eos:
            for u in new_eos:
                u.finished = True
            if verbose:
                print(f"Samples {sorted(u.idx for u in new_eos)} reached EOS token at step {new_eos[0].step + 1}.", flush=True)
            if audio_streamer is not None:
                audio_streamer.end(torch.tensor(sorted(u.idx for u in new_eos)))
        hit = [u for u in order if not u.finished and u.step >= u.max_step_sample]
        if hit:
            for u in hit:
                u.finished = u.reach_max = True
            if verbose:
                print(f"Samples {sorted(u.idx for u in hit)} reached max generation length at step {hit[0].step + 1}.", flush=True)
            if audio_streamer is not None:
                audio_streamer.end(torch.tensor(sorted(u.idx for u in hit)))
        if S.get("_seen"):
            for u in order:
                if u.finished:
                    S["_seen"].pop(u.idx, None)       # the repetition-penalty mask of a finished utterance ([V] bools on the device)
        for u in order:
            if u.last == end_id:
                e.codec_reset(u.slot)
            if refresh and not u.finished and u.last == start_id:
                # :549-565 -- the reference masks the whole negative cache and un-masks only the slot of the NEXT token, so the
                # negative context restarts empty and the next negative pass re-feeds <speech_start> at position 0
                # (pinned against the reference's generate(): tests/golden/generate_forced_*.npz)
                u.neg_len = 0
        # ---------------- next input embeddings (:569) ----------------
        live = [u for u in order if not u.finished]
        diff = [u for u in live if u.last == diff_id]
        nxt_x = self._nxt_x                     # rows re-packed to the next step's active order
        plain = [u for u in live if u.last != diff_id]
        if plain:
            self._embed_ids([u.last for u in plain], self._tmp_emb)
            for i, u in enumerate(plain):
                nxt_x[live.index(u)].copy_(self._tmp_emb[i])
        if spec_sample and diff != order:
            spec_sample = False                                  # wrong guess: drop the latent, undo the draw
            if rng_state is not None:
                if S.get("cpu_gen") is not None:
                    S["cpu_gen"].set_state(rng_state)
                else:
                    torch.set_rng_state(rng_state)
        cond_used = self._hidden if spec_sample else self._cond
        n = len(diff)
        if S.get("lockstep", True) and len(order) > 1:
            self._negative_bookkeeping(S, order, live, diff, refresh, start_id)
        if not refresh:
            # the entry the negative pass appended at this step stays for a live row that does not diffuse, unless some row of the
            # batch does: then the reference's correction of :590-624 shifts it back out (pinned by generate_norefresh_b2.npz)
            for u in plain:
                if not diff:
                    u.neg_len += 1
        if diff and spec_sample:
            for u in diff:
                u.neg_len += 1
        elif diff:
            # ---- negative condition ----
            for j, u in enumerate(diff):
                oi = order.index(u)
                self._cond[j].copy_(pos_hidden(oi)[0])
                if u.have_embeds:
                    self._cond[n + j].copy_(self._hidden[nR + oi])
                elif not refresh:
                    self._cond[n + j].copy_(self._neg_hidden[oi - nR])
                else:
                    # first negative step of a fresh utterance: the lone <speech_start> prompt token (:379-386)
                    e.lm_forward([(2 * u.slot + 1, u.neg_len)], self._start_emb, self._neg_hidden[j:j + 1])
                    self._cond[n + j].copy_(self._neg_hidden[j])
                u.neg_len += 1
            # ---- diffusion sampling (:697-710) ----
            nz = self._draw_noise(S, diff)
            if nz is None:
                nz = torch.randn(2 * n, e.cfg.latent_dim, generator=S.get("cpu_gen"))      # CPU global RNG, as the reference (:701), unless the session has its own
            self._stage_noise(nz, n)
            if S["sde"]:
                e.diffusion_sample(n, self._cond, self._noise, cfg_scale, self._latent, step_noise=self._sde_draws(S, n))
            else:
                e.diffusion_sample(n, self._cond, self._noise, cfg_scale, self._latent)
        if diff and S.get("lockstep", True):
            # The reference's VibeVoiceTokenizerStreamingCache.get (modular_vibevoice_tokenizer.py:198-207) answers "no history" for
            # EVERY row of a decode / encode call as soon as ONE of its rows has no entry yet: in a lock-step batch, a row that
            # diffuses for the first time costs the rows decoded with it their conv history for that frame (both tokenizers).  It
            # cannot fire on processor-built prompts (every row takes its first frame at step 0); pinned by
            # tests/golden/generate_late_start_b2*.npz.  A queue of independent requests (generate_continuous) keeps every row's own
            # history instead -- each request ends as generate() on it alone would.
            started = [u for u in diff if u.chunks]
            if started and len(started) < len(diff):
                for u in started:
                    e.codec_reset(u.slot)
        if diff:
            # ---- codec decode, semantic encode, connectors (:636-672) ----
            if self.batched_codecs and (len(diff) > 1 or self.chain_single):
                # the reference decodes / re-encodes the step's diffusion rows as one batch (:636-672): one engine call, the
                # weight-heavy tokenizer stages read their weights once for all rows
                e.codec_chain_batch([u.slot for u in diff], self._latent[:n], self._audio[:n],
                                    self._sem[:n] if e.cfg.sem_dim > 0 else None)
            elif len(diff) > 1 and self.concurrent_codecs and self._side_streams:
                # each utterance's tokenizer chain (decode -> semantic re-encode) is an independent, launch-latency
                # bound graph: fork them onto side streams so they overlap, join before the connectors
                self._fork_ev.record(e.stream)
                for j, u in enumerate(diff):
                    ss = self._side_streams[j % len(self._side_streams)]
                    ss.wait_event(self._fork_ev)
                    e.codec_decode(u.slot, self._latent[j:j + 1], self._audio[j], stream=ss)
                    if e.cfg.sem_dim > 0:
                        e.semantic_encode(u.slot, self._audio[j], self._sem[j], stream=ss)
                for ss in self._side_streams[:min(len(diff), len(self._side_streams))]:
                    self._join_ev.record(ss)
                    e.stream.wait_event(self._join_ev)
            else:
                for j, u in enumerate(diff):
                    e.codec_decode(u.slot, self._latent[j:j + 1], self._audio[j])
                    if e.cfg.sem_dim > 0:
                        e.semantic_encode(u.slot, self._audio[j], self._sem[j])
            # every live row diffuses, in order (the steady state of a speech segment): the connectors write the next step's LM input
            # rows in place -- no staging through _emb_out / nxt_x, two device copies fewer on the step's dependency line
            direct = diff == live and S["teacher"] is None
            e.connect(n, self._latent, self._sem if e.cfg.sem_dim > 0 else None, self._x_in if direct else self._emb_out)
            chunk = self._block_rows(S["frame_rows"])
            S["frame_rows"] += 1
            chunk[:n].copy_(self._audio[:n])
            fr = getattr(self, "_first_row", None)
            for j, u in enumerate(diff):
                if fr is not None and not u.chunks:
                    fr[id(u)] = S["frame_rows"] - 1
                u.chunks.append(chunk[j])
                if not direct:
                    nxt_x[live.index(u)].copy_(self._emb_out[j])
            if audio_streamer is not None:
                audio_streamer.put(chunk[:n, None, :].to(self.dtype), torch.tensor([u.idx for u in diff]))
            S["n_frames"] += n
            if trace is not None:
                trace.neg_hidden.append(cond_used[n:2 * n].cpu())
                trace.latents.append(self._latent[:n].cpu())
                trace.semantic.append(self._sem[:n].cpu())
        if live:
            if S["teacher"] is not None:
                # test hook (SURVEY 8d "teacher-forced per step"): the next step consumes the embeddings the oracle fed its LM at
                # this step, so a bf16-mode run is compared step by step without the autoregressive feedback compounding
                te = S["teacher"](S["step"], [u.idx for u in live])
                if te is not None:
                    nxt_x[:len(live)].copy_(te.to(self.device, torch.float32))
            if not (diff and direct):
                self._x_in[:len(live)].copy_(nxt_x[:len(live)])
            if trace is not None:
                trace.next_embeds.append(self._x_in[:len(live)].cpu())
        for u in order:
            u.step += 1
        for u in live:
            u.have_embeds = True
        return live

    def _negative_bookkeeping(self, S, order, live, diff, refresh, start_id):
        """The reference's array bookkeeping of the negative branch (oracle.generate.NegativeRow restates it with the tensors), masks
        and counters only.  Its correction of a non-diffusing row (:594-624) guards the mask shift and the K/V shift differently
        (:603 vs :613): for a row holding exactly one valid entry the mask moves and the K/V does not, so the reference KEEPS the entry
        appended at this step and masks the older one out.  Everywhere else the net effect is "this step's entry is dropped".  Needs a one-frame speech segment (or a non-diffusion token
        at step 1 with refresh_negative=False) in one row of a batch while another row diffuses; found by
        tools/fuzz_generate_vs_reference.py.  Followed with vv_kv_move: the entry of this step is moved onto the older one."""
        def fwd():
            for u in order:
                b = u.neg_book
                b[1] += 1
                b[0].append(1)
        if not refresh:
            fwd()
        else:
            for u in live:
                if u.last == start_id:
                    u.neg_book[0] = [0] * (len(u.neg_book[0]) - 1) + [1]
        if not diff:
            return
        if refresh:
            fwd()
        for u in live:
            if u in diff:
                continue
            mask, c, cnt = u.neg_book
            if c - cnt == 2 and mask[cnt] == 1:
                # the row's one valid entry sits at compact position 0, this step's entry (written by the speculative negative row of
                # the LM pass above) at position 1: the reference keeps the NEW one.  Position 1 -> 0 with its rotation, length stays 1
                assert u.neg_len == 1, (u.idx, u.neg_len)
                self.engine.kv_move(2 * u.slot + 1, 1, 0)
            if cnt + 1 < len(mask) - 1:
                mask[cnt + 1:] = mask[cnt:-1]
            mask[cnt] = 0
            u.neg_book[2] = cnt + 1

    def _sde_draws(self, S, n):
        """The variance noise of one frame's solver steps, [N, n, latent] fp32 on the device.  scheduler.step() draws
        randn(model_output.shape = [2n, latent], device=model_output.device, float32) once per solver step on the device's
        global generator (dpm_solver.py:994-997; generate() passes no generator); only the first n rows survive the next step's
        `speech[:n]` (modeling_vibevoice_inference.py:703-704).  Same calls, same order -> same stream for a seeded run."""
        e = self.engine
        N, L = self.ddpm_inference_steps, e.cfg.latent_dim
        if self._sde_flat is None:
            self._sde_flat = e.new(64 * MAX_BATCH * L)
        buf = self._sde_flat[:N * n * L].view(N, n, L)
        if S["sde_noise_fn"] is not None:                     # test hook: the recorded draws, [N, 2n, latent]
            buf.copy_(S["sde_noise_fn"](S["step"], N, 2 * n)[:, :n].to(buf.device, torch.float32))
            return buf
        for i in range(N):
            buf[i].copy_(torch.randn(2 * n, L, device=self.device, dtype=torch.float32, generator=S.get("dev_gen"))[:n])
        return buf

    @staticmethod
    def _draw_noise(S, utts):
        """explicit noise for `utts` (test / bench hooks), or None: draw torch.randn as the reference does"""
        if any(u.noise_fn is not None for u in utts):
            rows = [u.noise_fn(u.step, 2)[:1].to(torch.float32) for u in utts]
            return torch.cat(rows + rows)
        if S["noise_fn"] is not None:
            return S["noise_fn"](S["step"], 2 * len(utts))
        return None

    def _session(self, tokenizer, generation_config, cfg_scale, kwargs, audio_streamer, n_rows):
        e = self.engine
        if tokenizer is None:
            raise ValueError("generate() needs tokenizer= (speech_start_id / speech_end_id / speech_diffusion_id / eos_token_id)")
        do_sample, temperature, warp = self._generation_options(generation_config)
        start_id, end_id, diff_id = tokenizer.speech_start_id, tokenizer.speech_end_id, tokenizer.speech_diffusion_id
        eos_id = tokenizer.eos_token_id
        bos_id = getattr(tokenizer, "bos_token_id", None)
        valid = [start_id, end_id, diff_id, eos_id] + ([bos_id] if bos_id is not None else [])
        if self._valid_key != tuple(valid):
            e.set_valid_tokens(valid)
            self._valid_key = tuple(valid)
        algo = self._sched_cfg["algorithm_type"]
        e.set_num_steps(self.ddpm_inference_steps, t_cast_bf16=(self.dtype == torch.bfloat16 and kwargs.get("_t_cast", True)),
                        **({} if algo == "dpmsolver++" else {"algorithm_type": algo}))
        return dict(sde=(algo == "sde-dpmsolver++"), sde_noise_fn=kwargs.pop("_sde_noise_fn", None), nv=len(valid), valid_t=torch.tensor(valid, dtype=torch.long), start_id=start_id, end_id=end_id, diff_id=diff_id,
                    eos_id=eos_id, cfg_scale=cfg_scale, do_sample=do_sample, temperature=temperature, warp=warp,
                    pad_id=getattr(tokenizer, "pad_token_id", None),
                    trace=kwargs.pop("_trace", None), audio_streamer=audio_streamer, verbose=kwargs.get("verbose", False),
                    forced=kwargs.pop("_forced_tokens", None), noise_fn=kwargs.pop("_noise_fn", None), n_rows=n_rows,
                    teacher=kwargs.pop("_teacher_embeds", None), refresh_negative=bool(kwargs.get("refresh_negative", True)),
                    frame_rows=0, n_frames=0, step=0, sample_rows=None,
                    # a session's own generators (generate_interleaved gives every lane a pair): None = the process-global CPU / device
                    # generators, i.e. the reference's RNG streams
                    cpu_gen=(kwargs.get("_generators") or (None, None))[0], dev_gen=(kwargs.pop("_generators", None) or (None, None))[1])

    # ------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate(self, inputs=None, generation_config=None, logits_processor=None, stopping_criteria=None,
                 prefix_allowed_tokens_fn=None, synced_gpus=None, assistant_model=None, audio_streamer=None,
                 negative_prompt_ids=None, negative_prompt_attention_mask=None, speech_tensors=None,
                 speech_masks=None, speech_input_mask=None, is_prefill=True, return_speech=True,
                 cfg_scale=1.0, stop_check_fn: Optional[Callable[[], bool]] = None, tqdm_class=None, **kwargs):
        """logits_processor / stopping_criteria are accepted and unused, as in the reference (its generate() overwrites the
        arguments with the lists it builds itself, :375-377)."""
        e = self.engine
        tokenizer = kwargs.pop("tokenizer", None)
        kwargs.pop("parsed_scripts", None)
        kwargs.pop("all_speakers_list", None)
        max_length_times = kwargs.pop("max_length_times", 2)
        prefill_noise = kwargs.pop("_prefill_noise", None)
        hooks = kwargs.pop("_bench_hooks", None) or BenchHooks()  # measurement harness only (bench.py): see BenchHooks
        step_cb, kv_start, kv_fill_fn = hooks.step_callback, hooks.kv_start, hooks.kv_fill_fn
        input_ids = kwargs["input_ids"] if inputs is None else inputs
        attention_mask = kwargs.get("attention_mask")
        input_ids = input_ids.cpu()
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        attention_mask = attention_mask.cpu()
        B, L0 = input_ids.shape
        if B > MAX_BATCH or B > e.cfg.n_slots or 2 * B > e.cfg.max_rows:
            # the reference's batch is unbounded (:393-394); one engine pass carries MAX_BATCH utterances (and this engine was created with
            # n_slots / max_rows), so a larger batch is decoded through the continuous-admission queue and handed back in the batch's own
            # output form
            return self._generate_queued(input_ids, attention_mask, tokenizer, generation_config, cfg_scale, audio_streamer,
                                         speech_tensors, speech_masks, speech_input_mask, is_prefill, return_speech, stop_check_fn,
                                         max_length_times, prefill_noise, step_cb, kwargs)
        S = self._session(tokenizer, generation_config, cfg_scale, kwargs, audio_streamer, B)
        S["sample_rows"] = lambda order: list(range(B))
        self._frame_w = B                                 # frame-store rows are as wide as this call's batch
        if kwargs.get("max_new_tokens", None) is None:
            max_new_tokens = self.max_position_embeddings - L0
        else:
            max_new_tokens = kwargs["max_new_tokens"]
        max_length = min(L0 + max_new_tokens, e.max_ctx)
        utts = []
        for b in range(B):
            m = attention_mask[b].bool()
            utts.append(_Utt(b, b, input_ids[b][m].tolist(), L0, max_length, max_length_times, S["start_id"]))
        max_steps = min(max_length - L0, int(max_length_times * L0))
        time_prefill = os.environ.get("VVHIP_TIME_PREFILL") is not None     # debug: sync + time the two prefill phases
        if tqdm_class is not None and kwargs.get("show_progress_bar", True):
            progress = tqdm_class(range(max_steps), desc="Generating", leave=False)
        else:
            progress = range(max_steps)
        n_steps = 0
        with torch.cuda.stream(e.stream), _end_streamer_on_error(audio_streamer):
            for b in range(B):
                e.codec_reset(b)
            e.embed([S["start_id"]], self._start_emb)
            active = list(utts)
            for step in progress:
                S["step"] = step
                if step_cb is not None:
                    step_cb(step)
                if stop_check_fn is not None and stop_check_fn():
                    if audio_streamer is not None:
                        audio_streamer.end()
                    break
                if audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and any(audio_streamer.finished_flags):
                    break
                if not active:
                    break
                if L0 + step >= max_length:
                    for u in active:
                        u.reach_max = True
                    break
                if step == 0:
                    # ---------------- prompt prefill (:467-474, _process_speech_inputs) ----------------
                    sp_embeds = None
                    t_pf = [time.perf_counter()] if time_prefill else None
                    self._t_kv_fill = 0.0
                    self._t_lm_pass = 0.0
                    with_voice = is_prefill and speech_tensors is not None and speech_masks is not None
                    # masks are host data (the processor's output): positions and counts on the host (a device-side .sum() costs a
                    # lazy kernel-module load (~20 ms) on its first use and a sync on every use), uploaded while the stream is
                    # still idle -- behind the encoder a pageable copy would hold the host until the encoder has finished
                    sp_pos = {}
                    if with_voice and speech_input_mask is not None:
                        for u in utts:
                            sm_cpu = speech_input_mask[u.idx].cpu()[attention_mask[u.idx].bool().cpu()]
                            idx = sm_cpu.to(torch.bool).nonzero().squeeze(1)
                            if idx.numel():
                                sp_pos[u.idx] = (int(idx.numel()), idx.to(self.device))
                    if with_voice:
                        _, sp_embeds = self._process_speech_inputs(speech_tensors, speech_masks, prefill_noise)
                    if time_prefill:
                        e.sync(); torch.cuda.current_stream(self.device).synchronize(); t_pf.append(time.perf_counter())
                    sp_off = 0
                    jobs = []
                    for u in utts:
                        rows = pos = None
                        if sp_embeds is not None and u.idx in sp_pos:
                            cnt, pos = sp_pos[u.idx]
                            rows = sp_embeds[sp_off:sp_off + cnt]
                            sp_off += cnt
                        jobs.append((u, u.ids, rows, pos, kv_start, kv_fill_fn))
                    self._prefill_checked(jobs)
                    if time_prefill:
                        e.sync(); t_pf.append(time.perf_counter())
                        self.last_prefill = {"voice_encode_s": round(t_pf[1] - t_pf[0], 5),
                                             "lm_prefill_s": round(t_pf[2] - t_pf[1] - self._t_kv_fill, 5),
                                             "lm_passes_s": round(self._t_lm_pass, 5),      # the LM launches alone (the rest of
                                             "bench_kv_fill_s": round(self._t_kv_fill, 5)}  # lm_prefill_s: embedding + row scatter)
                active = self._iterate(S, active)
                n_steps += 1
            if audio_streamer is not None:
                audio_streamer.end()
            outs = [torch.cat(u.chunks, dim=-1)[None].to(self.dtype) if u.chunks else None for u in utts]
            seq = torch.full((B, L0 + n_steps), S["eos_id"], dtype=torch.long)
            seq[:, :L0] = input_ids
            for u in utts:
                if u.tokens:
                    seq[u.idx, L0:L0 + len(u.tokens)] = torch.tensor(u.tokens, dtype=torch.long)
        e.sync()
        self._release_frames()
        self.last_stats = {"frames": S["n_frames"], "steps": n_steps}
        return VibeVoiceGenerationOutput(
            sequences=seq.to(self.device), speech_outputs=outs if return_speech else None,
            reach_max_step_sample=torch.tensor([u.reach_max for u in utts], dtype=torch.bool).to(self.device))

    def _generate_queued(self, input_ids, attention_mask, tokenizer, generation_config, cfg_scale, audio_streamer, speech_tensors,
                         speech_masks, speech_input_mask, is_prefill, return_speech, stop_check_fn, max_length_times, prefill_noise,
                         step_cb, kwargs):
        """generate() for a batch of more than MAX_BATCH rows (the reference's batch is unbounded, :393-394): every row becomes a
        one-utterance request of generate_continuous() -- up to n_slots of them in flight, a finished row's slot refilled at once --
        and the results are assembled into ONE VibeVoiceGenerationOutput as the batched loop returns it (sequences [B, L0 + steps]
        padded with eos after a row's end, :499; speech_outputs one entry per row; reach_max_step_sample [B]).  Rows are
        independent in the reference's loop (no cross-sample arithmetic, :393-394,549,573,594) with ONE exception this path does not
        reproduce: a row whose first frame comes later than another's costs the rows decoded with it their tokenizer conv history
        for that frame (the cache quirk described in _iterate; impossible on processor-built prompts, where every row takes its first
        frame at step 0) -- and one consequence of the queue itself: the batch-level early exit with an audio_streamer (the loop ends when
        ANY stream has finished, :443-447) can fire while rows are still waiting for a slot; those rows return as their prompt without
        audio (the lock-step batch would have advanced them to that step) and a RuntimeWarning says so.
        So under greedy / forced decoding each row is exactly what the batched call gives it; RNG-dependent draws (diffusion noise, do_sample) are consumed in queue
        order instead of the batch's lock-step order.  Loop lengths follow the batch: every row's cap uses the batch's padded
        width L0 (:421-422)."""
        B, L0 = input_ids.shape
        if prefill_noise is not None:
            raise NotImplementedError("_prefill_noise (test hook) is per call; not supported for batches above MAX_BATCH")
        forced = kwargs.pop("_forced_tokens", None)
        noise_fn = kwargs.pop("_noise_fn", None)
        global _WARNED_QUEUED_RNG
        if not _WARNED_QUEUED_RNG and (noise_fn is None or self._generation_options(generation_config)[0]):
            _WARNED_QUEUED_RNG = True
            import warnings
            warnings.warn(f"generate(): a batch of {B} rows exceeds one engine pass ({min(MAX_BATCH, self.engine.cfg.n_slots, self.engine.cfg.max_rows // 2)} "
                          "utterances) and is decoded through the continuous-admission queue: every row is what generate() gives it alone, but "
                          "random draws (diffusion noise, do_sample) are consumed in queue order, not in the lock-step batch's order -- a seeded "
                          "run does not reproduce the reference's batch bit for bit", UserWarning, stacklevel=3)
        # voice-prompt rows: speaker i of speech_tensors contributes speech_masks[i].sum() frames; the rows' speech positions
        # consume those frames in row-major order (_process_speech_inputs + the masked scatter, :149-163,470-474)
        spk_of_row = [[] for _ in range(B)]
        if is_prefill and speech_tensors is not None and speech_masks is not None and speech_input_mask is not None:
            need = [int((speech_input_mask[b].cpu() & attention_mask[b].bool()).sum()) for b in range(B)]
            have = [int(speech_masks[i].sum()) for i in range(speech_masks.shape[0])]
            i = 0
            for b in range(B):
                got = 0
                while got < need[b]:
                    if i >= len(have):
                        raise ValueError("speech_masks hold fewer frames than speech_input_mask marks")
                    spk_of_row[b].append(i)
                    got += have[i]
                    i += 1
                if got != need[b]:
                    raise ValueError("a voice prompt spans two batch rows: the rows' speech positions must consume whole speakers")
        reqs = []
        for b in range(B):
            r = {"input_ids": input_ids[b:b + 1], "attention_mask": attention_mask[b:b + 1]}
            if spk_of_row[b]:
                idx = torch.tensor(spk_of_row[b], dtype=torch.long)
                r["speech_tensors"] = speech_tensors[idx]
                r["speech_masks"] = speech_masks[idx]
                r["speech_input_mask"] = speech_input_mask[b:b + 1]
            if forced is not None:
                r["_forced_tokens"] = forced[b]
            if noise_fn is not None:
                r["_noise_fn"] = noise_fn                # per utterance here: noise_fn(its own step, 2) -> [2, latent]
            reqs.append(r)
        kw = {k: v for k, v in kwargs.items() if k in ("verbose", "refresh_negative", "_trace", "_teacher_embeds", "_t_cast", "_sde_noise_fn")}
        outs = self.generate_continuous(reqs, tokenizer=tokenizer, generation_config=generation_config, cfg_scale=cfg_scale,
                                        audio_streamer=audio_streamer, is_prefill=is_prefill, return_speech=return_speech,
                                        max_new_tokens=kwargs.get("max_new_tokens"), max_length_times=max_length_times,
                                        stop_check_fn=stop_check_fn, _bench_hooks=BenchHooks(step_callback=step_cb), _batch_exit=True, **kw)
        eos = tokenizer.eos_token_id
        width = max(int(o.sequences.shape[1]) for o in outs)
        seq = torch.full((B, width), eos, dtype=torch.long, device=self.device)
        for b, o in enumerate(outs):
            seq[b, :o.sequences.shape[1]] = o.sequences[0]
        speech = [o.speech_outputs[0] if o.speech_outputs else None for o in outs] if return_speech else None
        return VibeVoiceGenerationOutput(sequences=seq, speech_outputs=speech,
                                         reach_max_step_sample=torch.cat([o.reach_max_step_sample.reshape(1) for o in outs]).to(self.device))

    # ------------------------------------------------------------------ two decode chains over one weight copy
    def fork(self, **runtime):
        """A second model object over THIS model's weights (Engine.fork -> vv_create_shared: one copy in HBM) with its own engine
        context -- KV caches, tokenizer state, graphs, stream -- and its own host-side buffers.  runtime: n_slots / max_ctx / max_rows
        overrides.  generate() on the fork and on the original may run at the same time from two host threads."""
        m = type(self)(self.config_dict, self.engine.fork(**runtime), self.dtype, self.requested_attn_implementation)
        m.set_speech_factors(self._scaling, self._bias)
        m.set_ddpm_inference_steps(self.ddpm_inference_steps)
        m._sched_cfg = dict(self._sched_cfg)
        m.concurrent_codecs, m.batched_codecs, m.speculate_sampling = self.concurrent_codecs, self.batched_codecs, self.speculate_sampling
        m.chain_single = self.chain_single
        return m

    def generate_interleaved(self, requests: List[dict], lanes: int = 2, audio_streamer=None, **kwargs) -> List[VibeVoiceGenerationOutput]:
        """generate_continuous() over `lanes` engine contexts that share this model's weights, one host thread and one stream per lane.
        A decode step is a chain of ~300-450 DEPENDENT launches, each paying a fixed boundary cost the chip idles through; a second,
        independent chain fills those boundaries (measured with two processes on one GPU in round 4: 1.63 x the aggregate at 1.5B).
        The queue is split longest-prompt-first over the lanes (parallel.shard_utterances); with greedy / forced decoding and explicit
        noise every request ends exactly as generate() on it alone (the lanes share nothing but read-only weights).  Random draws:
        every lane owns a CPU and a device torch.Generator seeded from the process-global CPU generator when the call starts, so a
        seeded call is reproducible and no lane consumes (or, undoing a speculative draw, rewinds) another lane's stream -- but the
        noise a request sees is its lane's, not what the same request would draw on the global generators through generate().  Returns the outputs in request order.  The lanes are
        created on first use (each owns KV caches for its n_slots) and kept: `model.close_lanes()` releases them."""
        import threading
        from .parallel import shard_utterances
        lanes = max(1, min(int(lanes), len(requests)))
        if lanes == 1:
            return self.generate_continuous(requests, audio_streamer=audio_streamer, **kwargs)
        pool = getattr(self, "_lanes", None) or []
        while len(pool) < lanes - 1:
            pool.append(self.fork())
        self._lanes = pool
        models = [self] + pool[:lanes - 1]
        shards = shard_utterances([int(r["input_ids"].shape[-1]) for r in requests], lanes)
        outs: List[Optional[VibeVoiceGenerationOutput]] = [None] * len(requests)
        errs: List[Optional[BaseException]] = [None] * lanes
        # every lane draws from its OWN generators (diffusion noise and the speculative draw's rewind on the CPU one; voice-prompt
        # sampling, multinomial and the sde variance noise on the device one), seeded here, on the caller's thread, from the
        # process-global CPU generator: a seeded call (torch.manual_seed) is reproducible, no lane can rewind or consume another
        # lane's draws, and the global device generator is not touched from the lane threads
        seeds = torch.randint(0, 2 ** 62, (lanes, 2), dtype=torch.int64).tolist()
        gens = []
        for k in range(lanes):
            cg = torch.Generator()
            cg.manual_seed(seeds[k][0])
            dg = torch.Generator(device=self.device)
            dg.manual_seed(seeds[k][1])
            gens.append((cg, dg))

        def run(k):
            try:
                torch.cuda.set_device(self.device)
                st = _LaneStreamer(audio_streamer, shards[k]) if audio_streamer is not None else None
                res = models[k].generate_continuous([requests[i] for i in shards[k]], audio_streamer=st, _generators=gens[k], **kwargs)
                for i, o in zip(shards[k], res):
                    outs[i] = o
            except BaseException as ex:                 # noqa: BLE001 -- handed to the caller's thread below
                errs[k] = ex
        threads = [threading.Thread(target=run, args=(k,), name=f"vv-lane-{k}") for k in range(1, lanes)]
        for t in threads:
            t.start()
        run(0)
        for t in threads:
            t.join()
        for ex in errs:
            if ex is not None:
                if audio_streamer is not None:
                    audio_streamer.end()
                raise ex
        if audio_streamer is not None:
            audio_streamer.end()
        self.last_stats = {"lanes": lanes, "frames": sum(m.last_stats.get("frames", 0) for m in models),
                           "capture_fallbacks": [m.engine.stat(4) if hasattr(m.engine, "stat") else 0 for m in models],
                           "per_lane": [dict(m.last_stats) for m in models], "shards": shards}
        return outs

    def close_lanes(self):
        for m in getattr(self, "_lanes", None) or []:
            m.engine.close()
        self._lanes = []

    # ------------------------------------------------------------------ continuous batching
    @torch.no_grad()
    def generate_continuous(self, requests: List[dict], tokenizer=None, generation_config=None, cfg_scale=1.0,
                            audio_streamer=None, is_prefill=True, return_speech=True, max_new_tokens=None,
                            max_length_times=2, stop_check_fn: Optional[Callable[[], bool]] = None,
                            max_concurrent: Optional[int] = None, **kwargs) -> List[VibeVoiceGenerationOutput]:
        """Decode a queue of single-utterance requests (each a dict of processor outputs with batch dimension 1) with up to
        `max_concurrent` (default: the engine's n_slots, at most 8) in flight.  A slot freed by EOS / length cap is refilled by
        the next queued request on the following iteration -- the running utterances never wait for a batch to drain; all rows
        in flight share every LM / diffusion-head weight pass.  Each request ends exactly as generate() on it alone would
        (greedy / forced decoding; with do_sample the draws interleave on the global generator).  Returns one
        VibeVoiceGenerationOutput per request, in request order; `audio_streamer` (batch_size = len(requests)) sees
        sample index = request index."""
        e = self.engine
        n_req = len(requests)
        cap = min(max_concurrent or e.cfg.n_slots, e.cfg.n_slots, MAX_BATCH, e.cfg.max_rows // 2)
        if cap < 1:
            raise ValueError("no engine slot available")
        kwargs = dict(kwargs)
        step_cb = (kwargs.pop("_bench_hooks", None) or BenchHooks()).step_callback
        batch_exit = bool(kwargs.pop("_batch_exit", False))      # _generate_queued: the reference's batch-level early exit applies
        if not kwargs.get("refresh_negative", True):
            # with refresh_negative=False a row's negative cache depends on whether ANOTHER row of the same batch diffuses at that
            # step (the correction of :590-624): defined for the lock-step batch of generate(), not for a queue of requests
            raise NotImplementedError("refresh_negative=False is a rule over the rows of one lock-step batch: use generate() with at "
                                      f"most {MAX_BATCH} rows (continuous admission / larger batches refuse it)")
        S = self._session(tokenizer, generation_config, cfg_scale, kwargs, audio_streamer, n_req)
        S["lockstep"] = False                   # independent requests: no cross-row tokenizer-cache coupling (see _iterate)
        S["sample_rows"] = lambda order: [u.idx for u in order]
        self._frame_w = cap
        self._first_row = {}
        queue = list(range(n_req))
        free = list(range(cap))
        done = [None] * n_req
        active: List[_Utt] = []
        it = 0
        stats = {"iterations": 0, "admissions": [], "max_in_flight": 0}
        with torch.cuda.stream(e.stream), _end_streamer_on_error(audio_streamer):
            e.embed([S["start_id"]], self._start_emb)
            while queue or active:
                S["step"] = it
                if step_cb is not None:
                    step_cb(it)
                if stop_check_fn is not None and stop_check_fn():
                    if audio_streamer is not None:
                        audio_streamer.end()
                    break
                if batch_exit and audio_streamer is not None and hasattr(audio_streamer, "finished_flags") and any(audio_streamer.finished_flags):
                    # standing in for ONE batched generate(): its loop ends with the first finished stream (:443-447).  The reference's
                    # lock-step batch has advanced EVERY row to this step by then; a queue has not -- rows still waiting for a slot come
                    # back as their prompt with no audio.  Said once, loudly: a caller that streams a batch this large wants to know.
                    if queue:
                        import warnings
                        warnings.warn(f"generate(): the batch-level early exit (a finished audio stream, modeling_vibevoice_inference.py:443-447) fired "
                                      f"while {len(queue)} of {n_req} rows were still queued for an engine slot: the reference's lock-step batch would "
                                      "have decoded them up to this step, here they return as their prompt without audio.  Use batches of at most "
                                      f"{cap} rows with an audio_streamer, or generate_continuous() (no batch-level exit).", RuntimeWarning, stacklevel=3)
                    break
                # ---- retire by the loop-level conditions of a batch-1 generate(): range(max_steps) exhausted / max_length ----
                keep, keep_rows = [], []
                for i, u in enumerate(active):
                    if u.step >= u.max_steps:
                        u.finished = True
                    elif u.seq_len0 + u.step >= u.max_length:
                        u.finished = u.reach_max = True
                    if not u.finished:
                        keep.append(u)
                        keep_rows.append(i)
                    elif audio_streamer is not None:
                        audio_streamer.end(torch.tensor([u.idx]))
                if keep and len(keep) != len(active):
                    # _iterate packed the next-step embeddings in the order of the utterances it returned (`active`) and reads
                    # them back by position: the rows of the survivors move up to the survivors' new positions
                    sel = torch.tensor(keep_rows, dtype=torch.long, device=self._x_in.device)
                    self._x_in[:len(keep)] = self._x_in.index_select(0, sel)
                active = keep
                # ---- refill free slots ----
                in_flight = {u.slot for u in active}
                free = [s for s in range(cap) if s not in in_flight]
                while queue and free:
                    ri = queue.pop(0)
                    slot = free.pop(0)
                    r = requests[ri]
                    ids_t = r["input_ids"].cpu()
                    am = r.get("attention_mask")
                    am = torch.ones_like(ids_t) if am is None else am.cpu()
                    if ids_t.shape[0] != 1:
                        raise ValueError("generate_continuous: every request carries exactly one utterance")
                    L0 = ids_t.shape[1]
                    mnt = r.get("max_new_tokens", max_new_tokens)
                    mnt = self.max_position_embeddings - L0 if mnt is None else mnt
                    u = _Utt(ri, slot, ids_t[0][am[0].bool()].tolist(), L0, min(L0 + mnt, e.max_ctx), max_length_times, S["start_id"])
                    u.forced, u.noise_fn, u.req = r.get("_forced_tokens"), r.get("_noise_fn"), r
                    u.t_admit = it
                    e.codec_reset(slot)
                    rows = pos = None
                    if is_prefill and r.get("speech_tensors") is not None and r.get("speech_masks") is not None:
                        _, sp = self._process_speech_inputs(r["speech_tensors"], r["speech_masks"], r.get("_prefill_noise"), dev_gen=S.get("dev_gen"))
                        sim = r.get("speech_input_mask")
                        if sim is not None:
                            idx = sim[0].cpu()[am[0].bool().cpu()].to(torch.bool).nonzero().squeeze(1)      # host data: no device count
                            if idx.numel():
                                pos = idx.to(self.device)
                                rows = sp[:int(idx.numel())]
                    self._prefill_checked([(u, u.ids, rows, pos, 0, None)])
                    done[ri] = u
                    stats["admissions"].append((it, ri, slot))
                    if u.max_steps > 0 and u.seq_len0 < u.max_length:
                        active.append(u)
                    else:
                        u.finished = True
                        u.reach_max = u.seq_len0 >= u.max_length
                if not active:
                    continue
                stats["max_in_flight"] = max(stats["max_in_flight"], len(active))
                before = active
                active = self._iterate(S, active)
                it += 1
                # a finished utterance's frames leave the shared frame store at once (its own contiguous tensor); blocks no
                # live utterance points into are dropped, so the store follows the audio IN FLIGHT, not the queue's total
                ended = [u for u in before if u.finished]
                for u in ended:
                    if u.chunks:
                        u.chunks = [torch.cat(u.chunks, dim=-1)]
                if ended:
                    rows0 = [self._first_row[id(u)] for u in active if id(u) in self._first_row]
                    lo = min(rows0 + [S["frame_rows"]]) // self.frame_block
                    for b in range(min(lo, len(self._audio_blocks))):
                        self._audio_blocks[b] = None
                    for u in ended:
                        self._first_row.pop(id(u), None)
            if audio_streamer is not None:
                audio_streamer.end()
            outs = []
            for ri in range(n_req):
                u = done[ri]
                if u is None:                       # stopped before admission
                    outs.append(VibeVoiceGenerationOutput(sequences=requests[ri]["input_ids"].to(self.device), speech_outputs=[None],
                                                          reach_max_step_sample=torch.tensor([False], device=self.device)))
                    continue
                ids_t = requests[ri]["input_ids"].cpu()
                seq = torch.cat([ids_t, torch.tensor([u.tokens], dtype=torch.long)], dim=-1) if u.tokens else ids_t
                audio = torch.cat(u.chunks, dim=-1)[None].to(self.dtype) if u.chunks else None
                outs.append(VibeVoiceGenerationOutput(sequences=seq.to(self.device), speech_outputs=[audio] if return_speech else None,
                                                      reach_max_step_sample=torch.tensor([u.reach_max], device=self.device)))
        e.sync()
        self._release_frames()
        stats["iterations"] = it
        self.last_stats = {"frames": S["n_frames"], "steps": it, **stats}
        return outs
