"""Pin the oracle (oracle/*.py) to the golden vectors recorded from the
reference's own classes by tests/golden/make_golden.py."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import codec, connector, dpm, head, lm

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    d = np.load(os.path.join(G, name))
    return {k: (torch.from_numpy(d[k]) if d[k].ndim else d[k].item()) for k in d.files}


def check_w(w, g):
    assert np.allclose(synth.checksum(w), g["wsum"].numpy(), rtol=1e-12), \
        "synthetic weights differ from the ones the golden file was generated with"


def close(a, b, tol=2e-5):
    err = (a - b).abs().max().item()
    ref = b.abs().max().item()
    assert err <= tol * max(1.0, ref), (err, ref)


@pytest.mark.parametrize("n", [5, 10, 20])
def test_schedule(n):
    g = load("dpm_schedule.npz")
    s = dpm.Schedule(n)
    assert torch.equal(s.timesteps, g[f"timesteps_{n}"])
    assert torch.equal(s.sigmas, g[f"sigmas_{n}"])


def test_head_forward():
    g = load("head_forward.npz")
    hc = synth.HeadCfg()
    w = synth.head_weights(hc)
    check_w(w, g)
    for t in g["ts"].tolist():
        out = head.head_forward(w, g["noisy"], torch.full((4,), float(t)), g["cond"], hc.layers, hc.eps)
        close(out, g[f"out_{t}"])


@pytest.mark.parametrize("n", [5, 10])
def test_sampler(n):
    g = load(f"sampler_{n}.npz")
    hc = synth.HeadCfg()
    w = synth.head_weights(hc)
    check_w(w, g)
    lat = dpm.sample_speech_tokens(
        lambda x, t, c: head.head_forward(w, x, t, c, hc.layers, hc.eps),
        g["pos"], g["neg"], g["cfg_scale"], n, g["noise"])
    close(lat, g["latent"], 1e-4)


def test_codec_decode_stream_and_reset():
    g = load("codec_decode.npz")
    cc = synth.CodecCfg()
    w = {**synth.encoder_weights(cc, 2), **synth.decoder_weights(cc, 3)}
    check_w(w, g)
    lat = g["latents"]
    st = {}
    chunks = []
    for t in range(lat.shape[1]):
        chunks.append(codec.decoder_forward(w, lat[:, t:t + 1].permute(0, 2, 1), cc.ratios, cc.dec_depths, st, cc.eps))
        if t == g["reset_after"]:
            codec.zero_state(st)
    close(torch.cat(chunks, -1), g["stream"])
    ra = g["reset_after"] + 1
    close(codec.decoder_forward(w, lat[:, :ra].permute(0, 2, 1), cc.ratios, cc.dec_depths, None, cc.eps), g["nonstream_a"])
    close(codec.decoder_forward(w, lat[:, ra:].permute(0, 2, 1), cc.ratios, cc.dec_depths, None, cc.eps), g["nonstream_b"])
    # invariant the reference implies: streaming == non-streaming per segment
    close(g["stream"][..., : ra * 3200], g["nonstream_a"], 1e-4)


def test_acoustic_encode_nonstream():
    g = load("acoustic_encode.npz")
    cc = synth.CodecCfg()
    w = {**synth.encoder_weights(cc, 2), **synth.decoder_weights(cc, 3)}
    check_w(w, g)
    mean = codec.encoder_forward(w, g["wav"].unsqueeze(1), cc.ratios, cc.enc_depths, None, cc.eps).permute(0, 2, 1)
    close(mean, g["mean"])


def test_semantic_encode_stream():
    g = load("semantic_encode.npz")
    sc = synth.CodecCfg(vae_dim=128)
    w = synth.encoder_weights(sc, 7)
    check_w(w, g)
    st = {}
    outs = []
    for t in range(g["stream"].shape[1]):
        o = codec.encoder_forward(w, g["audio"][:, :, t * 3200:(t + 1) * 3200], sc.ratios, sc.enc_depths, st, sc.eps)
        outs.append(o.permute(0, 2, 1))
        if t == g["reset_after"]:
            codec.zero_state(st)
    close(torch.cat(outs, 1), g["stream"])
    full = codec.encoder_forward(w, g["audio"][:, :, :3 * 3200], sc.ratios, sc.enc_depths, None, sc.eps).permute(0, 2, 1)
    close(full, g["nonstream_a"])


@pytest.mark.parametrize("name,din,seed", [("ac", 64, 4), ("sem", 128, 8)])
def test_connector(name, din, seed):
    g = load(f"connector_{name}.npz")
    w = synth.connector_weights(din, 96, seed)
    check_w(w, g)
    close(connector.connector_forward(w, g["x"]), g["y"])


LM_CFGS = {"d64": synth.LMCfg(), "d128": synth.LMCfg(hidden=256, heads=2, kv_heads=1, inter=384),
           "gqa": synth.LMCfg(hidden=256, heads=4, kv_heads=2, inter=320, layers=3)}


@pytest.mark.parametrize("tag", ["d64", "d128", "gqa"])
def test_lm(tag):
    g = load(f"lm_{tag}.npz")
    cfg = LM_CFGS[tag]
    w = synth.lm_weights(cfg)
    check_w(w, g)
    m = lm.Qwen2Oracle(w, cfg.layers, cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.theta, cfg.eps)
    c = m.new_cache()
    h = m.forward(m.embed(g["ids"][0]), c)
    close(h, g["prefill_hidden"], 1e-4)
    outs = [m.forward(g["dec_in"][i][None], c)[0] for i in range(g["dec_in"].shape[0])]
    close(torch.stack(outs), g["decode_hidden"], 1e-4)
