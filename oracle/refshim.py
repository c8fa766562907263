"""Import shims that let the reference's modules load -- and its two generate() loops run -- under transformers 5.x
without diffusers (SURVEY.md "Oracle import recipe").

Only tests/golden/make_golden.py uses this, and only inside the build container where /root/reference exists.
No arithmetic of the path lives here; the one computation is the attention-mask-derived position_ids that
transformers 4.51.3 (the version the reference pins) produced inside prepare_inputs_for_generation and 5.x no longer does.
"""
import dataclasses
import enum
import functools
import inspect
import sys
import types

import torch

REF_ROOT = "/root/reference"


def install():
    if "vibevoice.modular" in sys.modules:
        return
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF_ROOT)
    from transformers.models.auto import auto_factory
    _orig = auto_factory._LazyAutoMapping.register
    auto_factory._LazyAutoMapping.register = (
        lambda self, k, v, exist_ok=False: _orig(self, k, v, exist_ok=True))

    def _mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    _mod("diffusers")
    cu = _mod("diffusers.configuration_utils")
    u = _mod("diffusers.utils")
    tu = _mod("diffusers.utils.torch_utils")
    _mod("diffusers.schedulers")
    su = _mod("diffusers.schedulers.scheduling_utils")

    class _Cfg(dict):
        __getattr__ = dict.__getitem__

    class ConfigMixin:
        config = property(lambda self: self._internal_dict)

        def register_to_config(self, **kw):
            self.__dict__.setdefault("_internal_dict", _Cfg()).update(kw)

        @classmethod
        def from_config(cls, config, **kwargs):
            # diffusers' ConfigMixin.from_config: the recorded __init__ arguments with the overrides applied
            # (demo/gradio_demo.py:142-146 swaps the solver's algorithm_type this way).  Plumbing only.
            names = set(inspect.signature(cls.__init__).parameters) - {"self"}
            d = {k: v for k, v in dict(config).items() if k in names}
            d.update({k: v for k, v in kwargs.items() if k in names})
            return cls(**d)

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **kw):
            ba = inspect.signature(init).bind(self, *a, **kw)
            ba.apply_defaults()
            ConfigMixin.register_to_config(
                self, **{k: v for k, v in ba.arguments.items() if k != "self"})
            init(self, *a, **kw)
        return inner

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    u.deprecate = lambda *a, **k: None
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(
        shape, generator=generator, device=device, dtype=dtype)
    su.KarrasDiffusionSchedulers = enum.Enum(
        "KarrasDiffusionSchedulers", "DPMSolverMultistepScheduler")
    su.SchedulerMixin = type("SchedulerMixin", (), {})
    su.SchedulerOutput = dataclasses.make_dataclass(
        "SchedulerOutput", [("prev_sample", torch.Tensor)])
    pkg = types.ModuleType("vibevoice.modular")
    pkg.__path__ = [REF_ROOT + "/vibevoice/modular"]
    sys.modules["vibevoice.modular"] = pkg


def install_generate_shims():
    """Lets the reference's VibeVoiceForConditionalGenerationInference.generate() itself run under transformers 5.x
    (it was written against 4.51.3).  Pure API adaptation -- signatures that drifted, a module that moved -- so that
    tests/golden/make_golden.py can record the REFERENCE's loop on tiny seeded weights.  No arithmetic."""
    install()
    import transformers

    def _stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    try:
        import transformers.models.qwen2.tokenization_qwen2_fast  # noqa: F401
    except Exception:
        _stub("transformers.models.qwen2.tokenization_qwen2_fast",
              Qwen2TokenizerFast=type("Qwen2TokenizerFast", (transformers.PreTrainedTokenizerFast,), {}))
    try:
        from transformers.models.qwen2.tokenization_qwen2 import Qwen2Tokenizer  # noqa: F401
    except Exception:
        _stub("transformers.models.qwen2.tokenization_qwen2",
              Qwen2Tokenizer=type("Qwen2Tokenizer", (transformers.PreTrainedTokenizer,), {}))
    from transformers.generation.configuration_utils import GenerationMode
    from transformers.generation.utils import GenerationMixin
    from vibevoice.modular.modeling_vibevoice_inference import VibeVoiceForConditionalGenerationInference as Ref
    if getattr(Ref, "_vv_shimmed", False):
        return Ref
    # 4.51: _prepare_generation_config(gc, use_model_defaults, **kw); 5.x: (gc, **kw)
    Ref._prepare_generation_config = lambda self, gc, use_model_defaults=None, **kw: \
        GenerationMixin._prepare_generation_config(self, gc, **kw)
    # 4.51: (..., assistant_model, batch_size, max_cache_length, device); 5.x: (..., generation_mode, batch_size, max_cache_length)
    Ref._prepare_cache_for_generation = lambda self, gc, mk, assistant, bs, mcl, device=None: \
        GenerationMixin._prepare_cache_for_generation(self, gc, mk, GenerationMode.GREEDY_SEARCH, bs, mcl)
    # 4.51 always returned an 'inputs_embeds' entry (None when ids are used); 5.x omits the key
    _pifg = Ref.prepare_inputs_for_generation

    def _prepare_inputs(self, input_ids, *a, **k):
        # transformers 4.51.3 (the version the reference pins, pyproject.toml:22), generation/utils.py
        # prepare_inputs_for_generation step 3: with an attention_mask and no position_ids it sets
        #     position_ids = attention_mask.long().cumsum(-1) - 1 ; masked_fill_(attention_mask == 0, 1)
        # (sliced to the new tokens further down).  5.x dropped this and lets the model count positions from the
        # cache length -- which changes the reference's behaviour wherever the mask has holes: left-padded batches
        # and the negative branch after its <speech_start> reset (:549-565).  Restore the pinned behaviour.
        am = k.get("attention_mask")
        if am is not None and k.get("position_ids") is None:
            pos = am.long().cumsum(-1) - 1
            pos.masked_fill_(am == 0, 1)
            pkv = k.get("past_key_values")
            seen = pkv.get_seq_length() if pkv is not None else 0
            n_new = max(1, input_ids.shape[1] - int(seen))        # 4.51 slices to the not-yet-cached tokens (cache_position)
            k["position_ids"] = pos[:, -n_new:]
            k.setdefault("next_sequence_length", n_new)           # 5.x slices input_ids only when told how many are new
        out = _pifg(self, input_ids, *a, **k)
        out.setdefault("inputs_embeds", None)
        return out
    Ref.prepare_inputs_for_generation = _prepare_inputs
    # 4.51's DynamicCache exposed per-layer tensor lists (the reference edits them in place, :549-565, :609-616); 5.x keeps
    # them as cache.layers[i].keys / .values -- same tensors, so in-place edits through these views behave identically
    from transformers.cache_utils import DynamicCache
    if not hasattr(DynamicCache, "key_cache"):
        # (4.51's lists only hold layers that have been written: an untouched cache is an empty list)
        DynamicCache.key_cache = property(lambda self: [l.keys for l in self.layers if getattr(l, "keys", None) is not None])
        DynamicCache.value_cache = property(lambda self: [l.values for l in self.layers if getattr(l, "values", None) is not None])
    _tie = Ref.tie_weights
    Ref.tie_weights = lambda self, *a, **k: _tie(self)          # 5.x passes recompute_mapping=
    Ref._vv_shimmed = True
    return Ref


def expose_text_config(cfg):
    """transformers 5.x sizes its DynamicCache from config.num_hidden_layers & co; VibeVoiceConfig keeps them in
    decoder_config (4.51 did not look)."""
    for a in ("num_hidden_layers", "num_attention_heads", "num_key_value_heads", "hidden_size", "max_position_embeddings"):
        setattr(cfg, a, getattr(cfg.decoder_config, a))
    return cfg


def install_streaming_shims():
    """Same adaptation for the reference's Streaming-0.5B inference class (row Z)."""
    install_generate_shims()
    from transformers.generation.configuration_utils import GenerationMode
    from transformers.generation.utils import GenerationMixin
    import vibevoice.modular.modeling_vibevoice_streaming_inference as msi
    Ref = msi.VibeVoiceStreamingForConditionalGenerationInference
    if getattr(Ref, "_vv_shimmed", False):
        return Ref
    Ref._prepare_generation_config = lambda self, gc, use_model_defaults=None, **kw: \
        GenerationMixin._prepare_generation_config(self, gc, **kw)
    Ref._prepare_cache_for_generation = lambda self, gc, mk, assistant, bs, mcl, device=None: \
        GenerationMixin._prepare_cache_for_generation(self, gc, mk, GenerationMode.GREEDY_SEARCH, bs, mcl)
    _pifg = Ref.prepare_inputs_for_generation

    def _prepare_inputs(self, input_ids, *a, **k):          # see install_generate_shims: 4.51.3's mask-derived position_ids
        am = k.get("attention_mask")
        if am is not None and k.get("position_ids") is None:
            pos = am.long().cumsum(-1) - 1
            pos.masked_fill_(am == 0, 1)
            pkv = k.get("past_key_values")
            seen = pkv.get_seq_length() if pkv is not None else 0
            n_new = max(1, input_ids.shape[1] - int(seen))
            k["position_ids"] = pos[:, -n_new:]
            k.setdefault("next_sequence_length", n_new)
        out = _pifg(self, input_ids, *a, **k)
        out.setdefault("inputs_embeds", None)
        return out
    Ref.prepare_inputs_for_generation = _prepare_inputs
    _tie = Ref.tie_weights
    Ref.tie_weights = lambda self, *a, **k: _tie(self)
    Ref._vv_shimmed = True
    return Ref
