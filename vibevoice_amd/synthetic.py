"""Synthetic weights and inputs at the real model shapes (no checkpoints or
tokenizer files exist offline).  Used by bench.py,
__graft_entry__.smoke() and the full-size GPU tests -- never by generate().

Weights: seeded N(0, std) per tensor with the reference's key names and shapes;
layers the reference zero-initialises (diffusion-head adaLN / final linear) and
the 1e-6 layer scales are re-randomised so every stage carries signal.
Inputs reproduce the processor's prompt layout (vibevoice_processor.py:268-296):
  system-ish prefix, per-speaker voice prompt <start><diffusion x frames><end>,
  text, trailing <speech_start>; token ids otherwise uniform random.
"""
import types
from collections import OrderedDict

import torch

# Qwen2.5 ids the reference's tokenizer maps to speech control tokens
# (modular_vibevoice_text_tokenizer.py:165-181): <|vision_start|>, <|vision_end|>, <|vision_pad|>,
# <|image_pad|> (pad), <|endoftext|> (eos)
TOKENS = types.SimpleNamespace(speech_start_id=151652, speech_end_id=151653, speech_diffusion_id=151654,
                               pad_token_id=151655, eos_token_id=151643, bos_token_id=None)


def _depths(s):
    return [int(x) for x in s.split("-")] if isinstance(s, str) else list(s)


def _block(sh, p, C):
    sh[p + "gamma"] = (C,)
    sh[p + "ffn_gamma"] = (C,)
    sh[p + "norm.weight"] = (C,)
    sh[p + "ffn_norm.weight"] = (C,)
    sh[p + "mixer.conv.conv.conv.weight"] = (C, 1, 7)
    sh[p + "mixer.conv.conv.conv.bias"] = (C,)
    sh[p + "ffn.linear1.weight"] = (4 * C, C)
    sh[p + "ffn.linear1.bias"] = (4 * C,)
    sh[p + "ffn.linear2.weight"] = (C, 4 * C)
    sh[p + "ffn.linear2.bias"] = (C,)


def _encoder(sh, p, tc):
    nf, ratios, depths, vae = tc["encoder_n_filters"], list(reversed(tc["encoder_ratios"])), _depths(tc["encoder_depths"]), tc["vae_dim"]
    sh[p + "downsample_layers.0.0.conv.conv.weight"] = (nf, 1, 7)
    sh[p + "downsample_layers.0.0.conv.conv.bias"] = (nf,)
    for i, r in enumerate(ratios):
        sh[p + f"downsample_layers.{i+1}.0.conv.conv.weight"] = (nf * 2 ** (i + 1), nf * 2 ** i, 2 * r)
        sh[p + f"downsample_layers.{i+1}.0.conv.conv.bias"] = (nf * 2 ** (i + 1),)
    for i, d in enumerate(depths):
        for j in range(d):
            _block(sh, p + f"stages.{i}.{j}.", nf * 2 ** i)
    Cl = nf * 2 ** (len(depths) - 1)
    sh[p + "head.conv.conv.weight"] = (vae, Cl, 7)
    sh[p + "head.conv.conv.bias"] = (vae,)


def _decoder(sh, p, tc):
    nf, ratios, vae = tc.get("decoder_n_filters", 32), tc["encoder_ratios"], tc["vae_dim"]
    depths = list(reversed(_depths(tc["encoder_depths"])))
    nd = len(depths)
    C0 = nf * 2 ** (nd - 1)
    sh[p + "upsample_layers.0.0.conv.conv.weight"] = (C0, vae, 7)
    sh[p + "upsample_layers.0.0.conv.conv.bias"] = (C0,)
    for i, r in enumerate(ratios):
        cin, cout = nf * 2 ** (nd - 1 - i), nf * 2 ** (nd - 2 - i)
        sh[p + f"upsample_layers.{i+1}.0.convtr.convtr.weight"] = (cin, cout, 2 * r)
        sh[p + f"upsample_layers.{i+1}.0.convtr.convtr.bias"] = (cout,)
    for i, d in enumerate(depths):
        for j in range(d):
            _block(sh, p + f"stages.{i}.{j}.", nf * 2 ** (nd - 1 - i))
    sh[p + "head.conv.conv.weight"] = (1, nf, 7)
    sh[p + "head.conv.conv.bias"] = (1,)


def param_shapes(cfg) -> "OrderedDict[str, tuple]":
    d, h = cfg["decoder_config"], cfg["diffusion_head_config"]
    H, I, V = d["hidden_size"], d["intermediate_size"], d["vocab_size"]
    nh, nkv = d["num_attention_heads"], d["num_key_value_heads"]
    hd = H // nh
    sh = OrderedDict()
    p = "model.language_model."
    sh[p + "embed_tokens.weight"] = (V, H)
    for i in range(d["num_hidden_layers"]):
        q = p + f"layers.{i}."
        sh[q + "input_layernorm.weight"] = (H,)
        sh[q + "post_attention_layernorm.weight"] = (H,)
        for n, o in (("q", nh * hd), ("k", nkv * hd), ("v", nkv * hd)):
            sh[q + f"self_attn.{n}_proj.weight"] = (o, H)
            sh[q + f"self_attn.{n}_proj.bias"] = (o,)
        sh[q + "self_attn.o_proj.weight"] = (H, nh * hd)
        sh[q + "mlp.gate_proj.weight"] = (I, H)
        sh[q + "mlp.up_proj.weight"] = (I, H)
        sh[q + "mlp.down_proj.weight"] = (H, I)
    sh[p + "norm.weight"] = (H,)
    if not d.get("tie_word_embeddings", False):
        sh["lm_head.weight"] = (V, H)
    p = "model.prediction_head."
    L, Fd = h.get("latent_size", 64), int(H * h.get("head_ffn_ratio", 3.0))
    sh[p + "noisy_images_proj.weight"] = (H, L)
    sh[p + "cond_proj.weight"] = (H, H)
    sh[p + "t_embedder.mlp.0.weight"] = (H, 256)
    sh[p + "t_embedder.mlp.2.weight"] = (H, H)
    for i in range(h.get("head_layers", 4)):
        q = p + f"layers.{i}."
        sh[q + "norm.weight"] = (H,)
        sh[q + "adaLN_modulation.1.weight"] = (3 * H, H)
        sh[q + "ffn.gate_proj.weight"] = (Fd, H)
        sh[q + "ffn.up_proj.weight"] = (Fd, H)
        sh[q + "ffn.down_proj.weight"] = (H, Fd)
    sh[p + "final_layer.adaLN_modulation.1.weight"] = (2 * H, H)
    sh[p + "final_layer.linear.weight"] = (L, H)
    _encoder(sh, "model.acoustic_tokenizer.encoder.", cfg["acoustic_tokenizer_config"])
    _decoder(sh, "model.acoustic_tokenizer.decoder.", cfg["acoustic_tokenizer_config"])
    if cfg.get("semantic_tokenizer_config"):
        _encoder(sh, "model.semantic_tokenizer.encoder.", cfg["semantic_tokenizer_config"])
    for name, din in (("acoustic_connector", cfg.get("acoustic_vae_dim", 64)), ("semantic_connector", cfg.get("semantic_vae_dim", 128))):
        q = f"model.{name}."
        sh[q + "fc1.weight"] = (H, din)
        sh[q + "fc1.bias"] = (H,)
        sh[q + "norm.weight"] = (H,)
        sh[q + "fc2.weight"] = (H, H)
        sh[q + "fc2.bias"] = (H,)
    return sh


def streaming_param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """Key names/shapes of VibeVoiceStreamingForConditionalGenerationInference's hot-path parameters
    (modeling_vibevoice_streaming.py:108-164, modeling_vibevoice_streaming_inference.py:84-100)."""
    base = param_shapes(dict(cfg, semantic_tokenizer_config=None))
    n_tts = cfg["tts_backbone_num_hidden_layers"]
    n_lm = cfg["decoder_config"]["num_hidden_layers"] - n_tts
    H = cfg["decoder_config"]["hidden_size"]
    sh = OrderedDict()
    for k, v in base.items():
        if k.startswith("model.language_model.layers."):
            i = int(k.split(".")[3])
            rest = ".".join(k.split(".")[4:])
            if i < n_lm:
                sh[k] = v
            else:
                sh[f"model.tts_language_model.layers.{i - n_lm}.{rest}"] = v
        elif k == "model.language_model.norm.weight":
            sh["model.tts_language_model.norm.weight"] = v
        elif k.startswith("model.acoustic_tokenizer.encoder.") or k.startswith("model.semantic_") or k == "lm_head.weight":
            continue
        else:
            sh[k] = v
    sh["model.tts_input_types.weight"] = (2, H)
    sh["tts_eos_classifier.fc1.weight"] = (H, H)
    sh["tts_eos_classifier.fc1.bias"] = (H,)
    sh["tts_eos_classifier.fc2.weight"] = (1, H)
    sh["tts_eos_classifier.fc2.bias"] = (1,)
    return sh


def random_tensor(key, shape, gen, device, dtype):
    """One seeded tensor.  Matrices ~ N(0, 1/sqrt(fan_in)) (keeps activations O(1) through
    28 layers and the codec), norm weights ~ 1, biases small, layer scales 0.5."""
    if key.endswith("norm.weight") or key.endswith("layernorm.weight"):
        t = 1.0 + 0.02 * torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
    elif key.endswith("gamma"):
        t = torch.full(shape, 0.5, device=device, dtype=torch.float32)
    elif key.endswith(".bias"):
        t = 0.02 * torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
    elif key.endswith("embed_tokens.weight"):
        t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32)
    else:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "convtr" in key:
            fan_in = shape[0] * 2
        t = torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * (fan_in ** -0.5)
    return t.to(dtype)


def random_state_dict(cfg, device, seed=0, dtype=torch.bfloat16):
    """yields (reference_key, tensor) one at a time (the 7B bundle is 18.7 GB in bf16)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for k, shape in param_shapes(cfg).items():
        yield k, random_tensor(k, shape, gen, device, dtype)


def synthetic_inputs(cfg, n_speakers=1, text_tokens=220, voice_frames=75, seed=0, hop=3200, batch=1):
    """Processor-shaped inputs for `batch` utterances (all the same length here)."""
    g = torch.Generator().manual_seed(seed)
    T = TOKENS
    rows, sims = [], []
    for _ in range(batch):
        ids, sim = [], []

        def put(tokens, is_speech=False):
            ids.extend(tokens)
            sim.extend([is_speech] * len(tokens))
        put(torch.randint(0, 151000, (24,), generator=g).tolist())                 # system prompt + " Voice input:\n"
        for _s in range(n_speakers):
            put(torch.randint(0, 151000, (3,), generator=g).tolist())              # " Speaker i:"
            put([T.speech_start_id])
            put([T.speech_diffusion_id] * voice_frames, True)
            put([T.speech_end_id])
            put(torch.randint(0, 151000, (1,), generator=g).tolist())              # "\n"
        put(torch.randint(0, 151000, (text_tokens,), generator=g).tolist())        # " Text input:\n Speaker i: ..."
        put(torch.randint(0, 151000, (4,), generator=g).tolist())                  # " Speech output:\n"
        put([T.speech_start_id])
        rows.append(ids)
        sims.append(sim)
    input_ids = torch.tensor(rows, dtype=torch.long)
    attention_mask = torch.ones_like(input_ids)
    speech_input_mask = torch.tensor(sims, dtype=torch.bool)
    n_spk_total = n_speakers * batch
    speech_tensors = (torch.rand(n_spk_total, voice_frames * hop, generator=g) * 0.2 - 0.1)
    speech_masks = torch.ones(n_spk_total, voice_frames, dtype=torch.bool)
    return dict(input_ids=input_ids, attention_mask=attention_mask, speech_input_mask=speech_input_mask,
                speech_tensors=speech_tensors, speech_masks=speech_masks)


def forced_schedule(n_steps, turn=150):
    """Forced token schedule: turns of `turn` <speech_diffusion> separated by
    <speech_end>,<speech_start>; replaces argmax identically in the oracle and the HIP path."""
    T = TOKENS
    out = []
    while len(out) < n_steps:
        out.extend([T.speech_diffusion_id] * turn)
        out.extend([T.speech_end_id, T.speech_start_id])
    return out[:n_steps]
