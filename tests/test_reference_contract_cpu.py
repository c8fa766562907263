"""The checkpoint contract at REAL widths, against the reference's own classes (build container only: needs /root/reference;
skipped elsewhere).  No released checkpoint is available offline, so what can be pinned is the table a checkpoint is read through:
the reference's model, instantiated from its shipped config files (vibevoice/configs/qwen2.5_1.5b_64k.json, qwen2.5_7b_32k.json) with
the LM cut to two layers, must expose exactly the parameter names and shapes that vibevoice_amd's loader (engine.map_param_name,
from_pretrained) and the synthetic generator of bench.py (synthetic.param_shapes) use -- and the config mapper must accept those
files as they are."""
import copy
import json
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "vibevoice", "configs")), reason="needs the reference checkout (build container)")

CASES = {"1.5b": "qwen2.5_1.5b_64k.json", "7b": "qwen2.5_7b_32k.json"}


def _flat(d, p=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(_flat(v, p + k + "."))
        else:
            out[p + k] = v
    return out


@pytest.mark.parametrize("tag", sorted(CASES))
def test_shipped_config_files_map_onto_the_engine_config(tag):
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import engine_config_from_reference
    ref = json.load(open(os.path.join(REF, "vibevoice", "configs", CASES[tag])))
    ec = engine_config_from_reference(ref)                      # accepted as it is
    d = ref["decoder_config"]
    assert (ec.lm_hidden, ec.lm_layers, ec.lm_heads, ec.lm_kv_heads, ec.lm_inter, ec.lm_vocab) == (
        d["hidden_size"], d["num_hidden_layers"], d["num_attention_heads"], d["num_key_value_heads"], d["intermediate_size"], d["vocab_size"])
    assert ec.head_layers == ref["diffusion_head_config"]["head_layers"]
    # the architecture dict bench.py builds its synthetic model from says the same as the shipped file wherever it says anything
    mine, theirs = _flat(CONFIGS[tag]), _flat(ref)
    assert {k: (v, theirs.get(k)) for k, v in mine.items() if k in theirs and theirs[k] != v} == {}
    assert set(mine) <= set(theirs)


@pytest.mark.parametrize("tag", sorted(CASES))
def test_reference_model_exposes_the_parameter_table_the_loader_reads(tag):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import refshim
    Ref = refshim.install_generate_shims()
    from vibevoice.modular.configuration_vibevoice import VibeVoiceConfig
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.engine import map_param_name
    raw = json.load(open(os.path.join(REF, "vibevoice", "configs", CASES[tag])))
    raw["decoder_config"]["num_hidden_layers"] = 2             # per-layer names and shapes repeat; 28 layers would be 3-15 GB of RAM
    raw["decoder_config"]["vocab_size"] = 1024
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    # only names and shapes are read: skip every random initialisation (minutes of CPU at these widths)
    import torch.nn.init as init
    saved = {n: getattr(init, n) for n in ("normal_", "uniform_", "trunc_normal_", "kaiming_uniform_", "kaiming_normal_", "xavier_uniform_",
                                           "xavier_normal_", "constant_", "zeros_", "ones_")}
    tsaved = {n: getattr(torch.Tensor, n) for n in ("normal_", "uniform_", "fill_", "zero_")}
    try:
        for n in saved:
            setattr(init, n, lambda t, *a, **k: t)
        for n in tsaved:
            setattr(torch.Tensor, n, lambda self, *a, **k: self)
        cfg = VibeVoiceConfig(**raw)
        refshim.expose_text_config(cfg)
        model = Ref(cfg)
        ref_sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        del model
    finally:
        for n, f in saved.items():
            setattr(init, n, f)
        for n, f in tsaved.items():
            setattr(torch.Tensor, n, f)
        torch.set_default_dtype(old)
    mine = copy.deepcopy(CONFIGS[tag])
    mine["decoder_config"]["num_hidden_layers"] = 2
    mine["decoder_config"]["vocab_size"] = 1024
    syn = {k: tuple(s) for k, s in synthetic.param_shapes(mine).items()}
    # every reference tensor has an engine name, except the two scalar speech factors (from_pretrained -> set_speech_factors)
    not_taken = sorted(k for k in ref_sd if map_param_name(k) is None)
    assert not_taken == ["model.speech_bias_factor", "model.speech_scaling_factor"]
    ref_eng = {map_param_name(k): s for k, s in ref_sd.items() if map_param_name(k)}
    syn_eng = {map_param_name(k): s for k, s in syn.items() if map_param_name(k)}
    assert len(ref_eng) == len(ref_sd) - 2                      # the mapping is injective
    # what the engine is fed in bench.py (synthetic) is, name for name and shape for shape, a subset of what a checkpoint holds;
    # the only tensor a checkpoint may add is an untied lm_head (absent -> the embedding table is used, as tie_word_embeddings says)
    assert set(syn) <= set(ref_sd)
    assert set(ref_eng) - set(syn_eng) <= {map_param_name("lm_head.weight")}
    assert {n: (syn_eng[n], ref_eng[n]) for n in syn_eng if syn_eng[n] != ref_eng[n]} == {}


def test_reference_streaming_model_exposes_the_parameter_table_the_loader_reads():
    """Streaming-0.5B: no config file ships with the reference; the architecture dict of vibevoice_amd.configs goes through the
    reference's own VibeVoiceStreamingConfig / model classes (full depth: 4 + 20 layers at 896 wide) and every tensor the engine is
    fed is there with that shape -- the reference model only adds the acoustic tokenizer's ENCODER, which streaming inference never
    runs (voice prompts arrive as prefilled caches, modeling_vibevoice_streaming_inference.py:412-751)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import refshim
    RefS = refshim.install_streaming_shims()
    from vibevoice.modular.configuration_vibevoice_streaming import VibeVoiceStreamingConfig
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    raw = copy.deepcopy(CONFIGS["0.5b-streaming"])
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    tsaved = {n: getattr(torch.Tensor, n) for n in ("normal_", "uniform_", "fill_", "zero_")}
    try:
        for n in tsaved:
            setattr(torch.Tensor, n, lambda self, *a, **k: self)
        cfg = VibeVoiceStreamingConfig(**raw)
        refshim.expose_text_config(cfg)
        model = RefS(cfg)
        ref_sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        del model
    finally:
        for n, f in tsaved.items():
            setattr(torch.Tensor, n, f)
        torch.set_default_dtype(old)
    syn = {k: tuple(s) for k, s in synthetic.streaming_param_shapes(raw).items()}
    assert set(syn) <= set(ref_sd)
    extra = sorted(k for k in set(ref_sd) - set(syn) if not k.startswith("model.acoustic_tokenizer.encoder."))
    # the two scalar speech factors (set_speech_factors) and the TTS backbone's own embedding table, which the reference itself
    # calls unused (modeling_vibevoice_streaming.py:140)
    assert extra == ["model.speech_bias_factor", "model.speech_scaling_factor", "model.tts_language_model.embed_tokens.weight"]
    assert {k: (syn[k], ref_sd[k]) for k in syn if syn[k] != ref_sd[k]} == {}
