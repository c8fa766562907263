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


@pytest.mark.parametrize("n", [5, 10])
def test_sampler_sde(n):
    """The stochastic solver demo/gradio_demo.py:142-146 installs (`noise_scheduler.from_config(..., algorithm_type=
    'sde-dpmsolver++', beta_schedule='squaredcos_cap_v2')`): golden = the reference scheduler built exactly that way, its own
    variance-noise draws recorded (they are plain torch.randn([2n, 64]) calls on the global generator, one per solver step)."""
    g = load(f"sampler_sde_{n}.npz")
    hc = synth.HeadCfg()
    w = synth.head_weights(hc)
    check_w(w, g)
    lat = dpm.sample_speech_tokens(
        lambda x, t, c: head.head_forward(w, x, t, c, hc.layers, hc.eps),
        g["pos"], g["neg"], g["cfg_scale"], n, g["noise"], algorithm_type="sde-dpmsolver++", step_noise=g["step_noise"])
    close(lat, g["latent"], 1e-4)
    # the draws are the global generator's stream: seeded the same way, N x randn(2n, 64) reproduces them
    torch.manual_seed(int(g["seed"]))
    for i in range(n):
        assert torch.equal(torch.randn(4, hc.latent), g["step_noise"][i])
    # and the product's coefficient table (vibevoice_amd/schedule.py) restates the same update: x' = cs x + c0 x0 + c1 dx0 + cn eps
    from vibevoice_amd import schedule
    tv, coef = schedule.make_table(n, False, "sde-dpmsolver++")
    assert coef.shape == (n, 6) and coef[-1, 5] == 0.0 and (coef[:-1, 5] > 0).all()
    x = g["noise"][:2].clone().float()
    x0p = torch.zeros_like(x)
    cond = torch.cat([g["pos"], g["neg"]])
    for i in range(n):
        eps = head.head_forward(w, torch.cat([x, x]), torch.full((4,), float(tv[i])), cond, hc.layers, hc.eps)
        v = eps[2:] + float(g["cfg_scale"]) * (eps[:2] - eps[2:])
        a, s_, cs, c0, c1, cn = [float(c) for c in coef[i]]
        x0 = a * x - s_ * v
        x = cs * x + c0 * x0 + c1 * (x0 - x0p) + cn * g["step_noise"][i][:2]
        x0p = x0
        close(x, g["per_step"][i], 2e-4)


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


# ---------------------------------------------------------------- row G: the generate() loop itself
def _oracle_small(kv_round_bf16=False):
    """The tiny model of tests/gpu_util.build_small, oracle side only (no engine)."""
    import gpu_util
    lmcfg = synth.LMCfg()
    hc = synth.HeadCfg(hidden=lmcfg.hidden, layers=2)
    cc, sc = synth.CodecCfg(), synth.CodecCfg(vae_dim=128)
    s = gpu_util.Small(eng=None, lmcfg=lmcfg, hc=hc, cc=cc, sc=sc, lm_w=synth.lm_weights(lmcfg), lm_head=synth.lm_head_weight(lmcfg),
                       head_w=synth.head_weights(hc), ac_w={**synth.encoder_weights(cc, 2), **synth.decoder_weights(cc, 3)},
                       sem_w=synth.encoder_weights(sc, 7), ac_conn=synth.connector_weights(64, lmcfg.hidden, 4),
                       sem_conn=synth.connector_weights(128, lmcfg.hidden, 8))
    return s.oracle_model(kv_round_bf16=kv_round_bf16)


@pytest.mark.parametrize("name", ["generate_forced_b1", "generate_forced_b2", "generate_greedy_b1", "generate_cap_b1", "generate_ragged_voice_b1",
                                  "generate_norefresh_b1", "generate_norefresh_b2", "generate_times_b2", "generate_late_start_b2", "generate_late_start_b2r", "generate_single_entry_b2", "generate_multivoice_b2", "generate_ragged_voice_full_b2"])
def test_generate_loop_matches_the_reference_generate(name):
    """Golden = the REFERENCE's own generate() (modeling_vibevoice_inference.py:326-710) run on the tiny seeded model
    (tests/golden/make_golden.py::gen_generate, through oracle/refshim.install_generate_shims).  The oracle loop gets
    the same inputs, the same forced token plan and the recorded noise draws: token sequences identical, waveform
    rel-L2 <= 1e-4 (fp32 both sides; different summation order only).  Covers voice-prompt prefill, the negative branch's
    reset on <speech_start>, the cache fix-ups for non-diffusing rows of a desynchronised batch, the codec cache reset
    on <speech_end>, EOS / max-length bookkeeping; the two generate_norefresh files are refresh_negative=False runs (negative pass at
    every step, never reset, :503-516); generate_times_b2 runs with max_length_times=0.4 on a left-padded batch of two (the loop length
    follows the padded width, each row's cap its own length: the shorter row is stopped by reach_max_step_sample, :421-422,523-539);
    the two generate_late_start files start one row's first frame later than the other's, where the reference's tokenizer cache drops the
    streaming row's conv history for that frame (modular_vibevoice_tokenizer.py:198-207); generate_single_entry_b2 has a one-frame segment
    in one row while the other diffuses: the correction's mask shift happens and its K/V shift does not (:603 vs :613), the entry of
    THAT step stays and the older one is masked (oracle.generate.NegativeRow restates the arrays literally); generate_multivoice_b2 carries
    several voice samples per prompt (a multi-speaker script: three samples for two rows)."""
    from oracle import generate as ogen
    z = np.load(os.path.join(G, name + ".npz"))
    tok = ogen.TokenIds(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                        bos_token_id=None, pad_token_id=305)
    ids = torch.from_numpy(z["input_ids"])
    B = ids.shape[0]
    draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
    N = z["speech_tensors"].shape[0]                   # voice samples of the whole batch (generate_multivoice_b2: 3 for 2 rows)
    pre = (draws[0].reshape(N), draws[1].reshape(N, 3, 64))
    it = iter(draws[2:])

    def noise_fn(step, n2):
        d = next(it)
        assert d.numel() == n2 * 64, (d.numel(), n2)
        return d.reshape(n2, 64)
    forced = None
    if z["forced"].size:
        forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    mnt = {"generate_greedy_b1": 10, "generate_cap_b1": 6}.get(name)
    seq, audio, reach = ogen.oracle_generate(_oracle_small(), tok, ids, torch.from_numpy(z["attention_mask"]),
                                             torch.from_numpy(z["speech_tensors"]), torch.from_numpy(z["speech_masks"]),
                                             torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, num_steps=5,
                                             max_new_tokens=mnt, noise_fn=noise_fn, prefill_noise=pre, forced_tokens=forced,
                                             refresh_negative="norefresh" not in name,
                                             max_length_times={"generate_times_b2": 0.4}.get(name, 2))
    assert torch.equal(seq, torch.from_numpy(z["sequences"]))
    assert torch.equal(reach, torch.from_numpy(z["reach_max"]))
    assert next(it, None) is None                      # every recorded draw was consumed, in order
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        if ref.numel() == 0:
            assert audio[b] is None
            continue
        got = audio[b].reshape(-1)
        assert got.shape == ref.shape
        err = float((got - ref).norm() / ref.norm())
        assert err <= 1e-4, err


@pytest.mark.parametrize("name", ["generate_forced_b1_bf16", "generate_forced_b2_bf16"])
def test_the_oracle_in_bf16_follows_the_reference_in_bf16(name):
    """Pins what bench.py's parity blocks call `reference_bf16_vs_fp32` (the reference path's own rounding noise) to the REFERENCE:
    golden = the reference's own classes cast to bf16 -- its GPU dtype (demo/inference_from_file.py:284-292), run here on the CPU -- through
    its own generate() on the tiny seeded model with a forced plan, every draw and every frame's latents recorded; plus the fp32 reference
    replaying THE SAME draws (tests/golden/make_golden.py: run(..., dtype=torch.bfloat16)).  Both runs are free-running, as the reference runs.
    Held here: (1) the oracle in fp32 on those draws IS the fp32 reference (<= 1e-4); (2) the oracle with bf16 weights and activations picks
    the same tokens and stays within 3e-2 (latents, per frame) / 3e-2 (waveform) of the reference in bf16 -- several times closer than
    either is to fp32; (3) the two noise floors -- reference bf16 vs reference fp32, oracle bf16 vs oracle fp32 -- agree per frame to
    within 10 % (+1e-2).  So a floor measured with the oracle is the reference's floor."""
    from oracle import generate as ogen
    z = np.load(os.path.join(G, name + ".npz"))
    tok = ogen.TokenIds(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304, bos_token_id=None, pad_token_id=305)
    ids = torch.from_numpy(z["input_ids"])
    B = ids.shape[0]
    draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
    runs = {}
    for dt in (torch.bfloat16, torch.float32):
        it = iter(draws[2:])
        tr = ogen.Trace()
        seq, audio, reach = ogen.oracle_generate(
            ogen.cast_model(_oracle_small(), dt), tok, ids, torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["speech_tensors"]).to(dt),
            torch.from_numpy(z["speech_masks"]), torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, num_steps=5,
            noise_fn=lambda step, n2: next(it).reshape(n2, 64).to(dt), prefill_noise=(draws[0].reshape(B).to(dt), draws[1].reshape(B, 3, 64).to(dt)),
            forced_tokens=forced, trace=tr)
        assert torch.equal(seq, torch.from_numpy(z["sequences"])) and next(it, None) is None
        runs[dt] = (audio, tr.latents)
    n = int(z["n_latents"])
    assert len(runs[torch.bfloat16][1]) == n
    for i in range(n):
        r16, r32 = torch.from_numpy(z[f"latent_{i}"]), torch.from_numpy(z[f"fp32_latent_{i}"])
        o16, o32 = runs[torch.bfloat16][1][i], runs[torch.float32][1][i]
        assert rel(o32, r32) <= 1e-4, (i, rel(o32, r32))
        assert rel(o16, r16) <= 3e-2, (i, rel(o16, r16))
        f_ref, f_or = rel(r16, r32), rel(o16, o32)
        assert abs(f_or - f_ref) <= 0.10 * f_ref + 1e-2, (i, f_ref, f_or)
        assert rel(o16, r16) <= 0.5 * f_ref, (i, rel(o16, r16), f_ref)          # the two bf16 runs are closer to each other than to fp32
    for b in range(B):
        r16, r32 = torch.from_numpy(z[f"audio_{b}"]), torch.from_numpy(z[f"fp32_audio_{b}"])
        o16, o32 = runs[torch.bfloat16][0][b].reshape(-1), runs[torch.float32][0][b].reshape(-1)
        assert rel(o32, r32) <= 1e-4 and rel(o16, r16) <= 3e-2, (b, rel(o32, r32), rel(o16, r16))
        assert abs(rel(o16, o32) - rel(r16, r32)) <= 0.10 * rel(r16, r32) + 1e-2


@pytest.mark.parametrize("name", ["generate_sde_b1", "generate_sde_b2"])
def test_generate_loop_under_the_gradio_scheduler_matches_the_reference(name):
    """Golden = the REFERENCE's own generate() after demo/gradio_demo.py:142-146's scheduler swap
    (`model.model.noise_scheduler = noise_scheduler.from_config(config, algorithm_type='sde-dpmsolver++', ...)`): per frame the
    recorded stream holds the initial randn(2n, 64) and then one randn(2n, 64) per solver step (scheduler.step()'s variance
    noise).  The oracle loop, fed the same draws in the same order, reproduces sequences and waveforms."""
    from oracle import generate as ogen
    z = np.load(os.path.join(G, name + ".npz"))
    tok = ogen.TokenIds(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                        bos_token_id=None, pad_token_id=305)
    ids = torch.from_numpy(z["input_ids"])
    B = ids.shape[0]
    draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
    pre = (draws[0].reshape(B), draws[1].reshape(B, 3, 64))
    it = iter(draws[2:])
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    seq, audio, reach = ogen.oracle_generate(
        _oracle_small(), tok, ids, torch.from_numpy(z["attention_mask"]), torch.from_numpy(z["speech_tensors"]),
        torch.from_numpy(z["speech_masks"]), torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, num_steps=5,
        noise_fn=lambda step, n2: next(it).reshape(n2, 64), prefill_noise=pre, forced_tokens=forced,
        algorithm_type="sde-dpmsolver++", sde_noise_fn=lambda step, N, n2: torch.stack([next(it).reshape(n2, 64) for _ in range(N)]))
    assert torch.equal(seq, torch.from_numpy(z["sequences"]))
    assert torch.equal(reach, torch.from_numpy(z["reach_max"]))
    assert next(it, None) is None                      # every recorded draw was consumed, in order
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        got = audio[b].reshape(-1)
        assert got.shape == ref.shape
        err = float((got - ref).norm() / ref.norm())
        assert err <= 1e-4, err


def _oracle_streaming_small(n_lm=1, n_tts=2, eos_bias=None):
    """The tiny split model of tests/test_gpu_streaming.py::build, oracle side only."""
    from oracle import generate_streaming as ogs
    from oracle import lm as olm
    cfg = synth.LMCfg(hidden=128, layers=n_lm + n_tts, heads=2, kv_heads=1, inter=256, vocab=320)
    H = cfg.hidden
    w = synth.lm_weights(cfg)
    hc = synth.HeadCfg(hidden=H, layers=2)
    cc = synth.CodecCfg()
    g = synth.Gen(900)
    tts_types = g.normal((2, H), 0.5, mat=False)
    eos = {"fc1.weight": g.linear(H, H), "fc1.bias": g.vec(H, 0.1), "fc2.weight": g.linear(1, H, 0.3), "fc2.bias": g.vec(1, 0.1, -1.5)}
    if eos_bias is not None:
        eos["fc2.bias"] = torch.full((1,), float(eos_bias))
    lm_w = {k: v for k, v in w.items() if k.startswith("embed") or any(k.startswith(f"layers.{i}.") for i in range(n_lm))}
    tts_w = {"norm.weight": w["norm.weight"], "embed_tokens.weight": w["embed_tokens.weight"]}
    for j in range(n_tts):
        for k, v in w.items():
            if k.startswith(f"layers.{n_lm + j}."):
                tts_w[f"layers.{j}." + k[len(f"layers.{n_lm + j}."):]] = v
    mk = lambda ww, L: olm.Qwen2Oracle(ww, L, cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.theta, cfg.eps, kv_round_bf16=False)
    return ogs.StreamingOracleModel(lm=mk(lm_w, n_lm), tts_lm=mk(tts_w, n_tts), tts_types=tts_types, eos=eos,
                                    head_w=synth.head_weights(hc), head_layers=hc.layers, ac_w=synth.decoder_weights(cc, 3),
                                    ac_conn=synth.connector_weights(64, H, 4), ratios=cc.ratios, dec_depths=cc.dec_depths,
                                    scaling=0.2, bias=-0.05)


# ---------------------------------------------------------------- row Z: the Streaming-0.5B loop
@pytest.mark.parametrize("name", ["streaming_text12_cap40", "streaming_text3_cap20", "streaming_eos", "streaming_cap_on_text"])
def test_streaming_loop_matches_the_reference_generate(name):
    """Golden = the reference's VibeVoiceStreamingForConditionalGenerationInference.generate()
    (modeling_vibevoice_streaming_inference.py:412-751) on the tiny seeded split model, started from prefilled branches
    produced by its own forward_lm / forward_tts_lm (stored in the golden).  The oracle loop starts from the same caches
    and gets the recorded noise: same number of tokens, same stop reason, waveform rel-L2 <= 1e-4."""
    from oracle import generate_streaming as ogs
    from oracle import lm as olm
    z = np.load(os.path.join(G, name + ".npz"))
    om = _oracle_streaming_small(eos_bias=float(z["eos_bias"]) if name in ("streaming_eos", "streaming_cap_on_text") else None)

    def cache(tag, oracle_lm):
        c = oracle_lm.new_cache()
        n = int(z[f"{tag}_layers"])
        for li in range(n):
            c.k[li] = torch.from_numpy(z[f"{tag}_k{li}"]).clone()
            c.v[li] = torch.from_numpy(z[f"{tag}_v{li}"]).clone()
        c.length = c.k[0].shape[1]
        return c
    preset = ogs.Preset(cache("lm", om.lm), cache("tts", om.tts_lm), cache("neg_tts", om.tts_lm),
                        torch.from_numpy(z["tts_last"]), torch.from_numpy(z["neg_tts_last"]))
    draws = iter([torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))])
    max_length = preset.tts_cache.length + int(z["max_new"])
    n_tok, audio, reach, fin = ogs.oracle_generate_streaming(om, preset, torch.from_numpy(z["text"]), 1.5, 5,
                                                             lambda frame, n2: next(draws).reshape(n2, 64), max_length)
    assert next(draws, None) is None
    assert n_tok == int(z["n_tokens"])
    assert bool(reach) == bool(z["reach_max"][0])
    ref = torch.from_numpy(z["audio"])
    got = audio.reshape(-1)
    assert got.shape == ref.shape
    err = float((got - ref).norm() / ref.norm())
    assert err <= 1e-4, err
