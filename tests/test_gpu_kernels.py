"""GPU parity: every HIP stage (through the C ABI) against the oracle on seeded inputs.

Tolerances (stated, fp32-class): with xsplit=3 the engine multiplies exact bf16
weights by exact fp32 activations and accumulates in fp32, so the only
differences from the fp32 oracle are summation order and libm ulps:
rel-L2 <= 2e-5 per GEMM, <= 1e-4 per composed stage; the LM additionally
keeps a bf16 KV cache, which the oracle mirrors with kv_round_bf16=True.
xsplit=1 (bf16 activations inside the MFMA, the reference's GPU numerics):
rel-L2 <= 2e-2.
"""
import numpy as np
import pytest
import torch

import synth
from gpu_util import build_small, max_err, rel_err
from oracle import codec, connector, dpm, head
from oracle import lm as olm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sm():
    s = build_small(synth.LMCfg(), xsplit=3, max_rows=160)
    yield s
    s.eng.close()


def dev(t, eng):
    out = t.to(eng.device, torch.float32).contiguous()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("T,N,K", [(1, 64, 64), (2, 48, 96), (2, 4608, 3584), (16, 200, 72), (2, 64, 18944),
                                   (37, 130, 40), (200, 128, 32), (5, 20, 7)])
@pytest.mark.parametrize("xs,tol", [(3, 2e-5), (2, 2e-4), (1, 2e-2)])
def test_gemm_plain(sm, T, N, K, xs, tol):
    eng = sm.eng
    g = synth.Gen(T * 1000 + N + K)
    w = g.normal((N, K), 1.0 / np.sqrt(K))
    x = g.normal((T, K), 1.0, mat=False)
    ref = x @ w.t()
    wp = eng.pack_matrix(w)
    xd = dev(x, eng)
    for ks in (0, 1, 2, 4):
        y = eng.new(T, N)
        with torch.cuda.stream(eng.stream):
            eng.gemm_raw(wp, xd, y, N, K, xsplit=xs, ksplit=ks, nontemporal=ks & 1)
        eng.sync()
        assert rel_err(y, ref) <= tol, (ks, rel_err(y, ref))


def test_gemm_transpose_detecting(sm):
    # A = identity-like weights, asymmetric x: catches row/col swaps in the MFMA layout
    eng = sm.eng
    N = K = 64
    w = torch.eye(N)
    x = torch.arange(3 * K, dtype=torch.float32).reshape(3, K) / 7.0
    wp = eng.pack_matrix(w)
    y = eng.new(3, N)
    with torch.cuda.stream(eng.stream):
        eng.gemm_raw(wp, dev(x, eng), y, N, K, xsplit=3)
    eng.sync()
    assert max_err(y, x) < 1e-6


@pytest.mark.parametrize("pro,epi", [(1, 1), (1, 2), (0, 4), (1, 3), (0, 0), (0, 1), (1, 0)])
@pytest.mark.parametrize("xs,tol", [(2, 2e-4), (1, 2e-2)])
def test_gemm_tile_prefill_shapes(sm, pro, epi, xs, tol):
    """MFMA tile GEMM (tile.hip): taken for launches with >= 48 workgroups -- prompt-prefill sized problems.
    Ragged T (not a multiple of 64), K not a multiple of the 256-k chunk, N not a multiple of the 128-feature tile."""
    eng = sm.eng
    T, N, K = 333, 4112, 416
    g = synth.Gen(991 + pro * 10 + epi)
    w = g.normal((N, K), 1.0 / np.sqrt(K))
    w2 = g.normal((N, K), 1.0 / np.sqrt(K))
    x = g.normal((T, K), 1.0, mat=False)
    nw = g.vec(K, 0.1, 1.0)
    bias = g.vec(N, 0.3)
    nscale = g.uniform((N,), 0.5, 1.5)
    y0 = g.normal((T, N), 1.0, mat=False)
    xin = x
    if pro == 1:
        xin = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * nw
    acc = xin @ w.t()
    if epi == 1:
        ref = acc + bias
    elif epi == 2:
        ref = torch.nn.functional.gelu(acc + bias)
    elif epi == 3:
        ref = torch.nn.functional.silu(acc) * (xin @ w2.t())
    elif epi == 4:
        ref = y0 + nscale * (acc + bias)
    else:
        ref = acc
    y = dev(y0.clone(), eng)
    with torch.cuda.stream(eng.stream):
        eng.gemm_raw(eng.pack_matrix(w), dev(x, eng), y, N, K, pro=pro, epi=epi,
                     w2p=eng.pack_matrix(w2) if epi == 3 else None, nw=dev(nw, eng), eps=1e-5,
                     bias=dev(bias, eng), nscale=dev(nscale, eng), xsplit=xs)
    eng.sync()
    assert rel_err(y, ref) <= tol, rel_err(y, ref)


@pytest.mark.parametrize("epi", [0, 1, 4, 3])
@pytest.mark.parametrize("T,N,K", [(333, 4112, 416), (128, 128, 64), (2048, 1024, 1536), (70, 200, 96), (500, 36, 3584),
                                   (4100, 4112, 416), (4100, 4100, 2080)])
def test_prefill_gemm3(sm, epi, T, N, K):
    """prefill.hip: activations packed to bf16 MFMA fragments (+ RMSNorm in the packing kernel for the BIAS case), 128 x 128
    LDS-staged MFMA GEMM with the store / bias / residual / SwiGLU epilogues; ragged T (not a multiple of 128 / 16), N not a
    multiple of the 128-feature block, an odd number of 32-wide k-tiles (K = 416 -> 13), one-block problems.  The last two
    shapes fill the chip with 256 x 256 workgroups and therefore run the double-buffered kernel (vv_gemm4_kernel): ragged T
    (4100 = 16 row blocks + 4 rows), 257 feature tiles, odd k-tile counts (13, 65), every epilogue.  The reference multiplies the
    same bf16-ROUNDED activations (the packing kernel rounds once, after the norm) in fp32, so what is left is summation order:
    rel-L2 <= 2e-4 for the fp32-out epilogues (a wrong k-tile in a hundred is ~1e-1), <= 3e-3 for SwiGLU, whose output is stored
    as packed bf16 (one more rounding); the fp32-activation reference stays as the accuracy bound (<= 2e-2)."""
    eng = sm.eng
    g = synth.Gen(7000 + T + N + K + epi)
    w = g.normal((N, K), 1.0 / np.sqrt(K))
    w2 = g.normal((N, K), 1.0 / np.sqrt(K))
    x = g.normal((T, K), 1.0, mat=False)
    nw = g.vec(K, 0.1, 1.0)
    bias = g.vec(N, 0.3)
    y0 = g.normal((T, N), 1.0, mat=False)
    norm = epi == 1
    xin = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * nw if norm else x
    acc = xin @ w.t()
    ref = {0: acc, 1: acc + bias, 4: y0 + acc, 3: torch.nn.functional.silu(acc) * (xin @ w2.t())}[epi]
    x16 = synth.bf16_round(xin)
    a16 = (x16.double() @ w.double().t()).float()
    ref16 = {0: a16, 1: a16 + bias, 4: y0 + a16,
             3: torch.nn.functional.silu(a16) * (x16.double() @ w2.double().t()).float()}[epi]
    y = dev(y0.clone(), eng)
    with torch.cuda.stream(eng.stream):
        eng.gemm3_raw(eng.pack_matrix(w), dev(x, eng), y, N, K, epi=epi, w2p=eng.pack_matrix(w2) if epi == 3 else None,
                      nw=dev(nw, eng) if norm else None, eps=1e-5, bias=dev(bias, eng) if epi == 1 else None)
    eng.sync()
    assert rel_err(y, ref) <= 2e-2, rel_err(y, ref)
    assert max_err(y, ref) <= 6e-2, max_err(y, ref)              # no tile is missing or misplaced
    tight = 3e-3 if epi == 3 else 2e-4
    assert rel_err(y, ref16) <= tight, (rel_err(y, ref16), tight)
    assert max_err(y, ref16) <= 4 * tight, (max_err(y, ref16), tight)


def test_lm_prefill_one_pass_and_ragged_chunks_match_the_oracle(sm):
    """A 150-token prompt prefilled (a) in one launch (prefill attention kernel: 16 query rows per workgroup) and (b) in ragged
    7-row chunks (rope/append + split + merge kernels, below 8 rows): both against the oracle's causal forward."""
    eng = sm.eng
    H = sm.lmcfg.hidden
    g = synth.Gen(4242)
    n = min(150, eng.cfg.max_rows)
    x = g.normal((n, H), 1.0, mat=False)
    m = sm.oracle_lm(kv_round_bf16=True)
    ref = m.forward(x, m.new_cache())
    xd = dev(x, eng)
    hid_a = eng.new(n, H)
    hid_b = eng.new(n, H)
    with torch.cuda.stream(eng.stream):
        eng.lm_forward([(0, j) for j in range(n)], xd, hid_a)                # cache 0: one pass
        for i0 in range(0, n, 7):                                            # cache 1: ragged 7-row chunks
            k = min(7, n - i0)
            eng.lm_forward([(1, i0 + j) for j in range(k)], xd[i0:i0 + k], hid_b[i0:i0 + k])
    eng.sync()
    assert rel_err(hid_a, ref) <= 5e-4, rel_err(hid_a, ref)
    assert rel_err(hid_b, ref) <= 5e-4, rel_err(hid_b, ref)


def test_pcm16_matches_the_gradio_conversion(sm):
    """vv_audio_to_pcm16 == demo/gradio_demo.py:1058-1073 (convert_to_16_bit_wav) applied per chunk, bit for bit: a loud
    chunk is peak-normalised, a quiet one is not; truncation toward zero."""
    eng = sm.eng
    rng = np.random.default_rng(3)
    chunks = np.stack([(rng.standard_normal(3200) * sc).astype(np.float32) for sc in (0.2, 3.0, 0.999, 1.7)])
    chunks[2] = np.clip(chunks[2], -1.0, 1.0)
    ref = []
    for data in chunks:
        d = data.copy()
        if np.max(np.abs(d)) > 1.0:
            d = d / np.max(np.abs(d))
        ref.append((d * 32767).astype(np.int16))
    ref = np.stack(ref)
    out = torch.empty(4, 3200, dtype=torch.int16, device=eng.device)
    with torch.cuda.stream(eng.stream):
        eng.audio_to_pcm16(dev(torch.from_numpy(chunks), eng), out)
    eng.sync()
    assert np.array_equal(out.cpu().numpy(), ref)


def test_streamer_pcm16_mode(sm):
    """AudioStreamer(pcm16=engine): the consumer receives int16 chunks converted on the device."""
    from vibevoice_amd.streamer import AudioStreamer
    eng = sm.eng
    st = AudioStreamer(batch_size=2, timeout=20.0, pcm16=eng)
    x = (torch.randn(2, 1, 3200, generator=torch.Generator().manual_seed(1)) * 2.0).to(eng.device)
    st.put(x.to(torch.bfloat16), torch.tensor([0, 1]))
    st.end()
    for b in range(2):
        (c,) = list(st.get_stream(b))
        assert c.dtype == torch.int16 and c.shape == (1, 3200)
        d = x[b, 0].to(torch.bfloat16).float().cpu().numpy()
        d = d / np.max(np.abs(d)) if np.max(np.abs(d)) > 1.0 else d
        assert np.array_equal(c[0].numpy(), (d * 32767).astype(np.int16))
    st._thread.join(timeout=5.0)
    assert not st._thread.is_alive()


def test_create_rejects_unsupported_gqa_groups():
    from vibevoice_amd.engine import Engine, EngineConfig, EngineError
    with pytest.raises(EngineError, match="GQA group"):
        Engine(EngineConfig(lm_hidden=128 * 34, lm_layers=1, lm_heads=34, lm_kv_heads=2, lm_inter=256, lm_vocab=64, lm_head_dim=128))
    with pytest.raises(EngineError, match="GQA group"):
        Engine(EngineConfig(lm_hidden=128 * 6, lm_layers=1, lm_heads=6, lm_kv_heads=4, lm_inter=256, lm_vocab=64, lm_head_dim=128))


@pytest.mark.parametrize("T", [8, 40, 203])
@pytest.mark.parametrize("pro,epi", [(1, 1), (1, 2), (0, 4), (1, 3), (0, 0), (0, 1)])
@pytest.mark.parametrize("xs,tol", [(2, 2e-4), (1, 2e-2)])
def test_gemm_fused_tall(sm, T, pro, epi, xs, tol):
    """16-row GEMV form walking row tiles (grid.y): tokenizer stages / prefill chunks in bench and bf16-split modes"""
    eng = sm.eng
    N, K = 1040, 160          # 65 tiles x 13 row tiles > 512 workgroups at T = 203: also exercises the 4-wave form
    g = synth.Gen(177 + pro * 10 + epi + T)
    w = g.normal((N, K), 1.0 / np.sqrt(K))
    w2 = g.normal((N, K), 1.0 / np.sqrt(K))
    x = g.normal((T, K), 1.0, mat=False)
    nw = g.vec(K, 0.1, 1.0)
    bias = g.vec(N, 0.3)
    nscale = g.uniform((N,), 0.5, 1.5)
    y0 = g.normal((T, N), 1.0, mat=False)
    xin = x
    if pro == 1:
        xin = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * nw
    acc = xin @ w.t()
    if epi == 1:
        ref = acc + bias
    elif epi == 2:
        ref = torch.nn.functional.gelu(acc + bias)
    elif epi == 3:
        ref = torch.nn.functional.silu(acc) * (xin @ w2.t())
    elif epi == 4:
        ref = y0 + nscale * (acc + bias)
    else:
        ref = acc
    y = dev(y0.clone(), eng)
    with torch.cuda.stream(eng.stream):
        eng.gemm_raw(eng.pack_matrix(w), dev(x, eng), y, N, K, pro=pro, epi=epi,
                     w2p=eng.pack_matrix(w2) if epi == 3 else None, nw=dev(nw, eng), eps=1e-5,
                     bias=dev(bias, eng), nscale=dev(nscale, eng), xsplit=xs)
    eng.sync()
    assert rel_err(y, ref) <= tol, rel_err(y, ref)


@pytest.mark.parametrize("pro,epi", [(1, 1), (1, 2), (0, 4), (1, 3), (0, 0)])
def test_gemm_fused(sm, pro, epi):
    eng = sm.eng
    T, N, K = 3, 96, 160
    g = synth.Gen(77 + pro * 10 + epi)
    w = g.normal((N, K), 1.0 / np.sqrt(K))
    w2 = g.normal((N, K), 1.0 / np.sqrt(K))
    x = g.normal((T, K), 1.0, mat=False)
    nw = g.vec(K, 0.1, 1.0)
    bias = g.vec(N, 0.3)
    nscale = g.uniform((N,), 0.5, 1.5)
    y0 = g.normal((T, N), 1.0, mat=False)
    xin = x
    if pro == 1:
        xin = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-5) * nw
    acc = xin @ w.t()
    if epi == 1:
        ref = acc + bias
    elif epi == 2:
        ref = torch.nn.functional.gelu(acc + bias)
    elif epi == 3:
        ref = torch.nn.functional.silu(acc) * (xin @ w2.t())
    elif epi == 4:
        ref = y0 + nscale * (acc + bias)
    else:
        ref = acc
    y = dev(y0.clone(), eng)
    with torch.cuda.stream(eng.stream):
        eng.gemm_raw(eng.pack_matrix(w), dev(x, eng), y, N, K, pro=pro, epi=epi,
                     w2p=eng.pack_matrix(w2) if epi == 3 else None, nw=dev(nw, eng), eps=1e-5,
                     bias=dev(bias, eng), nscale=dev(nscale, eng), xsplit=3)
    eng.sync()
    assert rel_err(y, ref) <= 2e-5, rel_err(y, ref)


def test_head_forward(sm):
    eng = sm.eng
    g = synth.Gen(100)
    noisy = g.normal((4, 64), 1.0, mat=False)
    cond = g.normal((4, sm.hc.hidden), 1.0, mat=False)
    for t in (999.0, 500.0, 3.0):
        ref = head.head_forward(sm.head_w, noisy, torch.full((4,), t), cond, sm.hc.layers, sm.hc.eps)
        out = eng.new(4, 64)
        with torch.cuda.stream(eng.stream):
            eng.head_forward(dev(noisy, eng), t, dev(cond, eng), out)
        eng.sync()
        assert rel_err(out, ref) <= 1e-4, (t, rel_err(out, ref))


@pytest.mark.parametrize("n_steps", [5, 10, 20])
@pytest.mark.parametrize("n", [1, 2])
def test_sampler(sm, n_steps, n):
    eng = sm.eng
    g = synth.Gen(200 + n_steps + n)
    pos = g.normal((n, sm.hc.hidden), 1.0, mat=False)
    neg = g.normal((n, sm.hc.hidden), 1.0, mat=False)
    noise = g.normal((2 * n, 64), 1.0, mat=False)
    ref = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(sm.head_w, x, t, c, sm.hc.layers, sm.hc.eps),
                                   pos, neg, 1.3, n_steps, noise)
    eng.set_num_steps(n_steps)
    out = eng.new(n, 64)
    with torch.cuda.stream(eng.stream):
        eng.diffusion_sample(n, dev(torch.cat([pos, neg]), eng), dev(noise[:n], eng), 1.3, out)
    eng.sync()
    assert rel_err(out, ref) <= 5e-4, rel_err(out, ref)


@pytest.mark.parametrize("n_steps", [5, 20])
@pytest.mark.parametrize("n", [1, 2])
def test_sampler_sde(sm, n_steps, n):
    """vv_set_schedule_sde + vv_diffusion_sample_sde: the stochastic sde-dpmsolver++ update demo/gradio_demo.py:142-146 installs
    (fused CFG + solver epilogue with the per-step variance-noise term) against the oracle sampler, which is pinned to the
    reference scheduler built that way (tests/golden/sampler_sde_*.npz).  Then the ABI's guard rails: the deterministic entry
    point refuses a stochastic table and vice versa, and switching back restores the deterministic result."""
    from vibevoice_amd.engine import EngineError
    eng = sm.eng
    g = synth.Gen(700 + n_steps + n)
    pos = g.normal((n, sm.hc.hidden), 1.0, mat=False)
    neg = g.normal((n, sm.hc.hidden), 1.0, mat=False)
    noise = g.normal((2 * n, 64), 1.0, mat=False)
    step_noise = g.normal((n_steps, 2 * n, 64), 1.0, mat=False)
    hf = lambda x, t, c: head.head_forward(sm.head_w, x, t, c, sm.hc.layers, sm.hc.eps)
    ref = dpm.sample_speech_tokens(hf, pos, neg, 1.3, n_steps, noise, algorithm_type="sde-dpmsolver++", step_noise=step_noise)
    ref_det = dpm.sample_speech_tokens(hf, pos, neg, 1.3, n_steps, noise)
    assert rel_err(ref, ref_det) > 1e-2                                   # the noise term matters at this size
    out = eng.new(n, 64)
    cond, nz = dev(torch.cat([pos, neg]), eng), dev(noise[:n], eng)
    sn = dev(step_noise[:, :n].contiguous(), eng)
    try:
        eng.set_num_steps(n_steps, algorithm_type="sde-dpmsolver++")
        with torch.cuda.stream(eng.stream):
            eng.diffusion_sample(n, cond, nz, 1.3, out, step_noise=sn)
            eng.diffusion_sample(n, cond, nz, 1.3, out, step_noise=sn)     # second call: the captured graph replays
        eng.sync()
        assert rel_err(out, ref) <= 5e-4, rel_err(out, ref)
        with pytest.raises(EngineError, match="stochastic"):
            eng.diffusion_sample(n, cond, nz, 1.3, out)
    finally:
        eng.set_num_steps(n_steps)                                         # back to the model's own scheduler for the other tests
    with pytest.raises(EngineError, match="deterministic"):
        eng.diffusion_sample(n, cond, nz, 1.3, out, step_noise=sn)
    with torch.cuda.stream(eng.stream):
        eng.diffusion_sample(n, cond, nz, 1.3, out)
    eng.sync()
    assert rel_err(out, ref_det) <= 5e-4, rel_err(out, ref_det)


def test_connectors(sm):
    eng = sm.eng
    g = synth.Gen(400)
    lat = g.normal((3, 64), 1.0, mat=False)
    sem = g.normal((3, 128), 1.0, mat=False)
    ref = connector.connector_forward(sm.ac_conn, lat) + connector.connector_forward(sm.sem_conn, sem)
    out = eng.new(3, sm.lmcfg.hidden)
    with torch.cuda.stream(eng.stream):
        eng.connect(3, dev(lat, eng), dev(sem, eng), out)
    eng.sync()
    assert rel_err(out, ref) <= 5e-5
    ref2 = connector.connector_forward(sm.ac_conn, lat)
    with torch.cuda.stream(eng.stream):
        eng.connect(3, dev(lat, eng), None, out)
    eng.sync()
    assert rel_err(out, ref2) <= 5e-5


def test_codec_decode_stream_reset(sm):
    eng = sm.eng
    cc = sm.cc
    g = synth.Gen(300)
    nfr = 7
    lat = g.normal((nfr, 64), 1.0, mat=False)
    st = {}
    slot = 1
    with torch.cuda.stream(eng.stream):
        eng.codec_reset(slot)
    for t in range(nfr):
        scaled = lat[t] / sm.scaling - sm.bias
        ref = codec.decoder_forward(sm.ac_w, scaled[None, :, None], cc.ratios, cc.dec_depths, st, cc.eps)[0, 0]
        out = eng.new(3200)
        with torch.cuda.stream(eng.stream):
            eng.codec_decode(slot, dev(lat[t:t + 1], eng), out)
        eng.sync()
        assert rel_err(out, ref) <= 2e-4, (t, rel_err(out, ref))
        if t == 3:
            codec.zero_state(st)
            with torch.cuda.stream(eng.stream):
                eng.codec_reset(slot)


def test_semantic_encode_stream(sm):
    eng = sm.eng
    sc = sm.sc
    g = synth.Gen(301)
    nfr = 5
    audio = g.uniform((nfr, 3200), -0.5, 0.5)
    st = {}
    for t in range(nfr):
        ref = codec.encoder_forward(sm.sem_w, audio[t][None, None], sc.ratios, sc.enc_depths, st, sc.eps)[0, :, 0]
        out = eng.new(128)
        with torch.cuda.stream(eng.stream):
            eng.semantic_encode(0, dev(audio[t], eng), out)
        eng.sync()
        assert rel_err(out, ref) <= 2e-4, (t, rel_err(out, ref))
        if t == 2:
            codec.zero_state(st)
            with torch.cuda.stream(eng.stream):
                eng.codec_reset(0)


def test_acoustic_encode_nonstream(sm):
    eng = sm.eng
    cc = sm.cc
    g = synth.Gen(302)
    nfr = 5                                  # enc_frames=2 -> chunks of 2,2,1
    wav = g.uniform((nfr * 3200,), -0.5, 0.5)
    ref = codec.encoder_forward(sm.ac_w, wav[None, None], cc.ratios, cc.enc_depths, None, cc.eps)[0].t()
    out = eng.new(nfr, 64)
    with torch.cuda.stream(eng.stream):
        eng.acoustic_encode(nfr, dev(wav, eng), out)
    eng.sync()
    assert rel_err(out, ref) <= 2e-4, rel_err(out, ref)


@pytest.mark.parametrize("samples", [8000, 6500, 12801, 15999, 3201])
def test_acoustic_encode_ragged_tail(sm, samples):
    """A signal that does not fill its last frame: the reference right-pads PER strided conv layer (SConv1d.forward ->
    get_extra_padding_for_conv1d), i.e. past the end of the signal every strided conv reads zeros; vv_acoustic_encode_ragged zeroes
    those rows in front of each strided conv of the last pass.  Against the oracle encoder on the UNPADDED signal (which pads per
    layer, as the reference); enc_frames = 2, so the partial frame falls into a pass of one or two frames.  The whole-frame entry point
    on the zero-padded waveform differs in the last frame only (that is the deviation this entry point removes)."""
    eng, cc = sm.eng, sm.cc
    g = synth.Gen(303 + samples)
    wav = g.uniform((samples,), -0.5, 0.5)
    nfr = -(-samples // 3200)
    ref = codec.encoder_forward(sm.ac_w, wav[None, None], cc.ratios, cc.enc_depths, None, cc.eps)[0].t()
    assert ref.shape[0] == nfr
    padded = torch.zeros(nfr * 3200)
    padded[:samples] = wav
    out, out_whole = eng.new(nfr, 64), eng.new(nfr, 64)
    with torch.cuda.stream(eng.stream):
        eng.acoustic_encode(nfr, dev(padded, eng), out, valid_samples=samples)
        eng.acoustic_encode(nfr, dev(padded, eng), out_whole)
    eng.sync()
    assert rel_err(out, ref) <= 2e-4, rel_err(out, ref)
    for f in range(nfr):
        assert rel_err(out[f], ref[f]) <= 5e-4, (f, rel_err(out[f], ref[f]))
    if nfr > 1:
        assert rel_err(out_whole[:nfr - 1], ref[:nfr - 1]) <= 2e-4            # causal: the frames before the partial one agree either way
    if nfr * 3200 - samples >= 64:                                             # (one missing sample moves the last latent by 4e-4)
        assert rel_err(out_whole[nfr - 1], ref[nfr - 1]) > 1e-3                # ... and the partial frame is where the two paddings part


def test_kv_move_keeps_the_rotation(sm):
    """vv_kv_move: cached position src copied onto dst in every layer, keys keeping the rotation they were computed with -- the
    reference's single-entry negative correction leaves the entry of step 1 (rotated for position 1) as the row's only entry.  Three
    tokens through one cache of the engine and of the oracle; entry 1 moved onto 0 on both sides; the next token at position 1 must
    see the same context."""
    s = build_small(LM_CASES["gqa"], xsplit=3, max_ctx=256)
    eng = s.eng
    try:
        H = s.lmcfg.hidden
        m = s.oracle_lm(kv_round_bf16=True)
        g = synth.Gen(611)
        x = g.normal((3, H), 1.0, mat=False)
        oc = m.new_cache()
        m.forward(x[0:1], oc)
        m.forward(x[1:2], oc)
        for l in range(len(oc.k)):
            oc.k[l][:, 0] = oc.k[l][:, 1].clone()
            oc.v[l][:, 0] = oc.v[l][:, 1].clone()
        oc.truncate(1)
        ref = m.forward(x[2:3], oc)                       # position 1, context = the moved entry
        hid = eng.new(1, H)
        xd = dev(x, eng)
        with torch.cuda.stream(eng.stream):
            eng.lm_forward([(1, 0)], xd[0:1], hid)
            eng.lm_forward([(1, 1)], xd[1:2], hid)
            eng.kv_move(1, 1, 0)
            eng.lm_forward([(1, 1)], xd[2:3], hid)
        eng.sync()
        assert rel_err(hid, ref) <= 2e-3, rel_err(hid, ref)
        with pytest.raises(Exception):
            eng.kv_move(99, 0, 1)
    finally:
        eng.close()


@pytest.mark.parametrize("case", ["d64", "gqa"])
def test_decode_attention_ignores_cache_slots_past_the_sequence(case):
    """Cache positions >= the row's length carry a softmax weight of exactly 0 -- but the kernel multiplies whole 32-position blocks,
    and 0 x NaN is NaN.  A slot re-used by the next utterance, imported K/V, or a profiling replay (bench.py's roofline chains append
    whatever their stale inputs give) leaves arbitrary bits there: poison every position past the sequence with NaN / Inf in every
    layer and decode on -- the result must be the clean run's, bit for bit."""
    s = build_small(LM_CASES[case], xsplit=3, max_ctx=256)
    eng = s.eng
    try:
        cfg = s.lmcfg
        H, kvh, hd = cfg.hidden, cfg.kv_heads, cfg.hidden // cfg.heads
        g = synth.Gen(707)
        x = g.normal((40, H), 1.0, mat=False)
        xd = dev(x, eng)

        def run(poison):
            hid = eng.new(1, H)
            outs = []
            with torch.cuda.stream(eng.stream):
                for t in range(40):
                    if poison and t in (1, 5, 31, 33):
                        bad = torch.full((kvh, 96, hd), float("nan"), device=eng.device)
                        bad[:, ::3] = float("inf")
                        for layer in range(cfg.layers):
                            eng.kv_import_at(0, layer, t, bad, bad)          # positions t .. t+95: everything past the sequence
                    eng.lm_forward([(0, t)], xd[t:t + 1], hid)
                    outs.append(hid.clone())
            eng.sync()
            return torch.cat(outs)
        clean = run(False)
        dirty = run(True)
        assert bool(torch.isfinite(dirty).all())
        assert torch.equal(clean, dirty)
    finally:
        eng.close()


def test_prefill_attention_ignores_cache_slots_past_the_prompt():
    """The same for the prompt pass of the bf16 mode (vv_attn_prefill4 multiplies whole 64-position stages): NaN / Inf in the cache
    slots past the prompt's end, in every layer -- a re-used slot, imported K/V -- must not reach the prompt's hidden states.  The pass
    zeroes the V tail of its last stage first (vv_kv_zero_v_tail_kernel); prompt lengths off and on the 64-position grid."""
    s = build_small(LM_CASES["gqa"], xsplit=1, max_ctx=512, max_rows=256)
    eng = s.eng
    try:
        cfg = s.lmcfg
        H, kvh, hd = cfg.hidden, cfg.kv_heads, cfg.hidden // cfg.heads
        g = synth.Gen(708)
        for L0 in (70, 100, 128, 191):
            x = dev(g.normal((L0, H), 1.0, mat=False), eng)

            def run(poison):
                hid = eng.new(L0, H)
                with torch.cuda.stream(eng.stream):
                    if poison:
                        bad = torch.full((kvh, 512 - L0, hd), float("nan"), device=eng.device)
                        bad[:, ::3] = float("inf")
                    else:
                        bad = torch.zeros((kvh, 512 - L0, hd), device=eng.device)
                    for layer in range(cfg.layers):
                        eng.kv_import_at(0, layer, L0, bad, bad)              # everything past the prompt
                    eng.lm_forward_span(0, 0, L0, x, hid)
                eng.sync()
                return hid.clone()
            clean, dirty = run(False), run(True)
            assert bool(torch.isfinite(dirty).all()), L0
            assert torch.equal(clean, dirty), L0
    finally:
        eng.close()


LM_CASES = {"d64": synth.LMCfg(), "d128": synth.LMCfg(hidden=256, heads=2, kv_heads=1, inter=384),
            "gqa": synth.LMCfg(hidden=256, heads=4, kv_heads=2, inter=320, layers=3)}


@pytest.mark.parametrize("tag", ["d64", "d128", "gqa"])
def test_lm_prefill_decode_and_logits(tag):
    s = build_small(LM_CASES[tag], xsplit=3, max_ctx=1024)
    eng = s.eng
    try:
        H = s.lmcfg.hidden
        m = s.oracle_lm(kv_round_bf16=True)
        g = synth.Gen(500)
        L0 = 37
        ids = torch.from_numpy(g.rng.integers(0, s.lmcfg.vocab, (L0,)))
        oc = m.new_cache()
        ref_pre = m.forward(m.embed(ids), oc)
        emb = eng.new(L0, H)
        for i0 in range(0, L0, 16):
            with torch.cuda.stream(eng.stream):
                eng.embed(ids[i0:i0 + 16].tolist(), emb[i0:])
        eng.sync()
        assert max_err(emb, m.embed(ids)) == 0.0
        hid = eng.new(L0, H)
        for i0 in range(0, L0, 16):
            n = min(16, L0 - i0)
            with torch.cuda.stream(eng.stream):
                eng.lm_forward([(2, i0 + j) for j in range(n)], emb[i0:i0 + n], hid[i0:i0 + n])
        eng.sync()
        assert rel_err(hid, ref_pre) <= 3e-4, rel_err(hid, ref_pre)
        # decode steps on cache 2 plus an independent short cache 1 in the same pass
        oc2 = m.new_cache()
        x = g.normal((6, 2, H), 1.0, mat=False)
        for i in range(6):
            r1 = m.forward(x[i, 0:1], oc)
            r2 = m.forward(x[i, 1:2], oc2)
            out = eng.new(2, H)
            with torch.cuda.stream(eng.stream):
                eng.lm_forward([(2, L0 + i), (1, i)], dev(x[i], eng), out)
            eng.sync()
            assert rel_err(out[0], r1[0]) <= 3e-4, (i, rel_err(out[0], r1[0]))
            assert rel_err(out[1], r2[0]) <= 3e-4, (i, rel_err(out[1], r2[0]))
        # restricted logits
        valid = [5, 17, 300 % s.lmcfg.vocab, 2]
        eng.set_valid_tokens(valid)
        lg = eng.new(2, len(valid))
        with torch.cuda.stream(eng.stream):
            eng.lm_logits(2, out, lg)
        eng.sync()
        ref_lg = torch.nn.functional.linear(torch.stack([r1[0], r2[0]]), s.lm_head)[:, valid]
        assert rel_err(lg, ref_lg) <= 5e-4
    finally:
        eng.close()


def test_lm_long_context_split_attention():
    # many positions -> every flash-decoding split and the tail masking are exercised
    s = build_small(LM_CASES["gqa"], xsplit=3, max_ctx=2048)
    eng = s.eng
    try:
        H = s.lmcfg.hidden
        m = s.oracle_lm(kv_round_bf16=True)
        g = synth.Gen(501)
        L0 = 700
        x = g.normal((L0, H), 1.0, mat=False)
        oc = m.new_cache()
        ref = m.forward(x, oc)
        hid = eng.new(L0, H)
        xd = dev(x, eng)
        for i0 in range(0, L0, 16):
            n = min(16, L0 - i0)
            with torch.cuda.stream(eng.stream):
                eng.lm_forward([(0, i0 + j) for j in range(n)], xd[i0:i0 + n], hid[i0:i0 + n])
        eng.sync()
        assert rel_err(hid[-50:], ref[-50:]) <= 5e-4, rel_err(hid[-50:], ref[-50:])
        assert rel_err(hid, ref) <= 5e-4
    finally:
        eng.close()


def test_bf16_activation_mode_tolerance():
    s = build_small(synth.LMCfg(), xsplit=1)
    eng = s.eng
    try:
        g = synth.Gen(600)
        pos = g.normal((1, s.hc.hidden), 1.0, mat=False)
        neg = g.normal((1, s.hc.hidden), 1.0, mat=False)
        noise = g.normal((2, 64), 1.0, mat=False)
        ref = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(s.head_w, x, t, c, s.hc.layers, s.hc.eps),
                                       pos, neg, 1.3, 10, noise)
        eng.set_num_steps(10)
        out = eng.new(1, 64)
        with torch.cuda.stream(eng.stream):
            eng.diffusion_sample(1, dev(torch.cat([pos, neg]), eng), dev(noise[:1], eng), 1.3, out)
        eng.sync()
        assert rel_err(out, ref) <= 5e-2, rel_err(out, ref)
    finally:
        eng.close()


@pytest.mark.parametrize("n", [1, 3, 8])
def test_bf16_mode_batched_sampler_rows(n):
    """8 utterances diffusing together = 16 head rows: the 16-row adaLN / gated-residual / CFG+DPM GEMV forms (bench mode
    only) against the oracle sampler, SURVEY 8d tolerance for bf16 activations.  n = 1 (two rows): the decode forms of the bf16 mode,
    i.e. the folded shift operand and the solver-step seam launch (headtail.hip) over its double-buffered state, deterministic and
    stochastic solver."""
    s = build_small(synth.LMCfg(), xsplit=1, n_slots=8)
    eng = s.eng
    try:
        g = synth.Gen(610 + n)
        pos = g.normal((n, s.hc.hidden), 1.0, mat=False)
        neg = g.normal((n, s.hc.hidden), 1.0, mat=False)
        noise = g.normal((2 * n, 64), 1.0, mat=False)
        ref = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(s.head_w, x, t, c, s.hc.layers, s.hc.eps),
                                       pos, neg, 1.3, 10, noise)
        eng.set_num_steps(10)
        out = eng.new(n, 64)
        with torch.cuda.stream(eng.stream):
            eng.diffusion_sample(n, dev(torch.cat([pos, neg]), eng), dev(noise[:n], eng), 1.3, out)
        eng.sync()
        assert rel_err(out, ref) <= 5e-2, rel_err(out, ref)
        # the stochastic solver through the 16-row forms (gradio scheduler, batched rows)
        sn = g.normal((10, 2 * n, 64), 1.0, mat=False)
        ref_s = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(s.head_w, x, t, c, s.hc.layers, s.hc.eps),
                                         pos, neg, 1.3, 10, noise, algorithm_type="sde-dpmsolver++", step_noise=sn)
        eng.set_num_steps(10, algorithm_type="sde-dpmsolver++")
        with torch.cuda.stream(eng.stream):
            eng.diffusion_sample(n, dev(torch.cat([pos, neg]), eng), dev(noise[:n], eng), 1.3, out, step_noise=dev(sn[:, :n].contiguous(), eng))
        eng.sync()
        assert rel_err(out, ref_s) <= 5e-2, rel_err(out, ref_s)
    finally:
        eng.close()


def test_lora_merge_reaches_the_engine(tmp_path):
    """load_lora_assets(): a peft-format LM adapter is folded into the packed weights; the LM then matches the oracle
    run on W + (alpha/r) B A."""
    import json as _json
    import types as _types
    from safetensors.torch import save_file
    from vibevoice_amd import lora
    s = build_small(synth.LMCfg(), xsplit=3)
    eng = s.eng
    try:
        g = synth.Gen(8080)
        r = 4
        targets = ["layers.0.self_attn.q_proj", "layers.1.mlp.down_proj", "layers.0.self_attn.o_proj"]
        sd, merged = {}, dict(s.lm_w)
        for t in targets:
            w = s.lm_w[t + ".weight"]
            a = g.normal((r, w.shape[1]), 0.3)
            b = g.normal((w.shape[0], r), 0.3)
            sd[f"base_model.model.{t}.lora_A.weight"] = a.contiguous()
            sd[f"base_model.model.{t}.lora_B.weight"] = b.contiguous()
            merged[t + ".weight"] = (w + (8 / r) * (b @ a)).to(torch.bfloat16).to(torch.float32)   # engine stores bf16
        root = tmp_path / "ft" / "lora"
        root.mkdir(parents=True)
        save_file(sd, str(root / "adapter_model.safetensors"))
        (root / "adapter_config.json").write_text(_json.dumps({"r": r, "lora_alpha": 8}))
        model = _types.SimpleNamespace(engine=eng)
        rep = lora.load_lora_assets(model, str(tmp_path / "ft"), base_state=lambda k: s.lm_w[k[len(lora.LM_PREFIX):]])
        assert rep.language_model and rep.merged_tensors == 3
        c = s.lmcfg
        orc = olm.Qwen2Oracle(merged, c.layers, c.heads, c.kv_heads, c.head_dim, c.theta, c.eps, kv_round_bf16=True)
        x = g.normal((6, c.hidden), 1.0, mat=False)
        ref = orc.forward(x, olm.KVCache(c.layers))
        hid = eng.new(6, c.hidden)
        with torch.cuda.stream(eng.stream):
            eng.lm_forward([(0, j) for j in range(6)], dev(x, eng), hid)
        eng.sync()
        assert rel_err(hid, ref) <= 5e-4, rel_err(hid, ref)
    finally:
        eng.close()


@pytest.mark.parametrize("n", [1, 5, 16])
def test_lm_logits_over_the_whole_vocabulary_and_the_logits_processors(sm, n):
    """vv_lm_logits_full (one wave per vocabulary row, fp32 products of the bf16 table) against the fp32 matmul, and the product's
    full-vocabulary processors (vibevoice_amd/modeling.py::_full_vocab_scores) against HF's own classes -- the ones the reference
    gets from GenerationMixin._get_logits_processor (modeling_vibevoice_inference.py:310-319): repetition penalty, temperature,
    top-k, top-p, min-p, in that order."""
    import types as _types
    from transformers.generation.logits_process import (MinPLogitsWarper, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    eng = sm.eng
    H, V = sm.lmcfg.hidden, sm.lmcfg.vocab
    g = synth.Gen(9100 + n)
    hid = g.normal((n, H), 2.0, mat=False)
    table = sm.lm_head if getattr(sm, "lm_head", None) is not None else sm.lm_w["embed_tokens.weight"]
    ref = hid @ table.float().t()
    out = eng.new(16 * V)
    with torch.cuda.stream(eng.stream):
        eng.lm_logits_full(n, dev(hid, eng), out)
    eng.sync()
    got = out[:n * V].view(n, V).cpu()
    assert rel_err(got, ref) <= 1e-5, rel_err(got, ref)
    # processors
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
    order = []
    for i in range(n):
        ids = [int(t) for t in g.rng.integers(0, V, (9 + i,))]
        order.append(_types.SimpleNamespace(idx=i, ids=ids, tokens=[int(t) for t in g.rng.integers(0, V, (3,))], seq_len0=30, init_len=len(ids)))
    warp = dict(top_k=200, top_p=0.93, min_p=0.002, repetition_penalty=1.3)
    S = dict(warp=warp, do_sample=True, temperature=0.7, pad_id=V - 1)
    with torch.cuda.stream(eng.stream):
        sc = m._full_vocab_scores(dev(hid, eng), order, S)
    eng.sync()
    sc = sc.cpu()
    want = ref.clone()
    for i, u in enumerate(order):
        row_ids = torch.tensor([[V - 1] + u.ids + u.tokens])
        want[i:i + 1] = RepetitionPenaltyLogitsProcessor(1.3)(row_ids, want[i:i + 1])
    for proc in (TemperatureLogitsWarper(0.7), TopKLogitsWarper(200), TopPLogitsWarper(0.93), MinPLogitsWarper(0.002)):
        want = proc(None, want)
    keep_w, keep_g = torch.isfinite(want), torch.isfinite(sc)
    assert int((keep_w != keep_g).sum()) <= n            # a token exactly at a filter's boundary may fall on either side
    both = keep_w & keep_g
    assert float((sc[both] - want[both]).abs().max()) <= 1e-4 * float(want[both].abs().max())
