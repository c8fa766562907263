"""Streaming-0.5B path: engine generate() vs the oracle loop on identical
presets, text windows and noise.  Tolerances as in test_gpu_generate.py (xsplit=3)."""
import types

import pytest
import torch

import synth
from gpu_util import rel_err
from oracle import generate_streaming as ogs
from oracle import lm as olm

pytestmark = pytest.mark.gpu


def build(n_lm=1, n_tts=2, use_graph=False, xsplit=3, lmcfg=None, head_layers=2, model_dtype=torch.float32):
    from vibevoice_amd.engine import Engine, EngineConfig
    from vibevoice_amd.modeling_streaming import VibeVoiceStreamingForConditionalGenerationInference
    cfg = lmcfg or synth.LMCfg(hidden=128, layers=n_lm + n_tts, heads=2, kv_heads=1, inter=256, vocab=320)
    assert cfg.layers == n_lm + n_tts
    H = cfg.hidden
    w = synth.lm_weights(cfg)
    hc = synth.HeadCfg(hidden=H, layers=head_layers)
    cc = synth.CodecCfg()
    head_w = synth.head_weights(hc)
    ac_w = synth.decoder_weights(cc, 3)
    ac_conn = synth.connector_weights(64, H, 4)
    g = synth.Gen(900)
    tts_types = g.normal((2, H), 0.5, mat=False)
    eos = {"fc1.weight": g.linear(H, H), "fc1.bias": g.vec(H, 0.1), "fc2.weight": g.linear(1, H, 0.3), "fc2.bias": g.vec(1, 0.1, -1.5)}
    lm_w = {k: v for k, v in w.items() if k.startswith("embed") or any(k.startswith(f"layers.{i}.") for i in range(n_lm))}
    tts_w = {"norm.weight": w["norm.weight"], "embed_tokens.weight": w["embed_tokens.weight"]}
    for j in range(n_tts):
        for k, v in w.items():
            if k.startswith(f"layers.{n_lm + j}."):
                tts_w[f"layers.{j}." + k[len(f"layers.{n_lm + j}."):]] = v
    mk = lambda ww, L: olm.Qwen2Oracle(ww, L, cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.theta, cfg.eps, kv_round_bf16=True)
    om = ogs.StreamingOracleModel(lm=mk(lm_w, n_lm), tts_lm=mk(tts_w, n_tts), tts_types=tts_types, eos=eos, head_w=head_w,
                                  head_layers=hc.layers, ac_w=ac_w, ac_conn=ac_conn, ratios=cc.ratios,
                                  dec_depths=cc.dec_depths, scaling=0.2, bias=-0.05)
    ecfg = EngineConfig(lm_hidden=H, lm_layers=cfg.layers, lm_heads=cfg.heads, lm_kv_heads=cfg.kv_heads, lm_inter=cfg.inter,
                        lm_vocab=cfg.vocab, head_layers=hc.layers, n_filters=cc.n_filters, enc_depths=cc.enc_depths,
                        sem_dim=0, has_acoustic_encoder=False, n_slots=2, max_ctx=512, xsplit=xsplit,
                        use_graph=use_graph, tts_layers=n_tts)
    eng = Engine(ecfg)
    sd = {"lm." + k: v for k, v in w.items()}
    sd.update({"head." + k: v for k, v in head_w.items()})
    sd.update({"dec." + k[len("decoder."):]: v for k, v in ac_w.items()})
    sd.update({"ac_conn." + k: v for k, v in ac_conn.items()})
    sd["tts_input_types.weight"] = tts_types
    sd.update({"eos." + k: v for k, v in eos.items()})
    eng.load_state_dict(sd, mapped=True, strict=True)
    cfgd = {"decoder_config": {"max_position_embeddings": 512}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "tts_backbone_num_hidden_layers": n_tts}
    model = VibeVoiceStreamingForConditionalGenerationInference(cfgd, eng, model_dtype=model_dtype)
    model.set_speech_factors(0.2, -0.05)
    model.set_ddpm_inference_steps(5)
    return om, model, cfg


def preset_for_engine(p, n_lm, n_tts):
    def branch(cache, last):
        kv = [(cache.k[i][None], cache.v[i][None]) for i in range(len(cache.k))]
        hid = torch.zeros(1, cache.length, last.shape[0])
        hid[0, -1] = last
        return types.SimpleNamespace(past_key_values=kv, last_hidden_state=hid)
    return {"lm": branch(p.lm_cache, torch.zeros_like(p.tts_last)), "tts_lm": branch(p.tts_cache, p.tts_last),
            "neg_lm": None, "neg_tts_lm": branch(p.neg_tts_cache, p.neg_tts_last)}


def run(om, model, cfg, n_text, max_new, seed=5):
    g = synth.Gen(seed)
    prompt = torch.from_numpy(g.rng.integers(0, 300, (23,)))
    text = torch.from_numpy(g.rng.integers(0, 300, (n_text,)))
    bank = {}

    def noise_fn(frame, n2):
        if frame not in bank:
            bank[frame] = synth.Gen(seed * 100 + frame).normal((n2, 64), 1.0, mat=False)
        return bank[frame]
    pre_o = ogs.make_preset(om, prompt, 305)
    pre_e = preset_for_engine(ogs.make_preset(om, prompt, 305), len(om.lm.w), 0)
    max_length = pre_o.tts_cache.length + max_new
    otr, htr = [], []
    n_tok, audio, reach, fin = ogs.oracle_generate_streaming(om, pre_o, text, 1.5, 5, noise_fn, max_length, otr)
    out = model.generate(tts_text_ids=text[None], all_prefilled_outputs=pre_e, cfg_scale=1.5, max_new_tokens=max_new,
                         _noise_fn=noise_fn, _trace=htr)
    return (n_tok, audio, reach, fin, otr), (out, htr)


def check(o, h):
    (n_tok, audio, reach, fin, otr), (out, htr) = o, h
    assert len(otr) == len(htr)
    for a, b in zip(htr, otr):
        assert rel_err(a["latent"], b["latent"]) <= 5e-3
        assert rel_err(a["tts_last"], b["tts_last"]) <= 5e-3
        assert abs(a["eos"] - b["eos"]) <= 5e-3 * max(1.0, abs(b["eos"]))
    assert bool(out.reach_max_step_sample[0]) == reach
    assert out.speech_outputs[0].shape[-1] == audio.shape[-1]
    assert rel_err(out.speech_outputs[0][0], audio[0]) <= 1e-2


def test_streaming_text_windows_then_max_length():
    om, model, cfg = build()
    try:
        # 12 text tokens = windows of 5,5,2, then speech-only windows until the length cap
        o, h = run(om, model, cfg, n_text=12, max_new=40)
        check(o, h)
        assert o[2] or o[3]
    finally:
        model.engine.close()


def test_streaming_with_graphs_short_text():
    om, model, cfg = build(use_graph=True)
    try:
        o, h = run(om, model, cfg, n_text=3, max_new=17, seed=9)
        check(o, h)
    finally:
        model.engine.close()
