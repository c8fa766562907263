"""CPU-only checks: C-ABI surface, host-side schedule/config logic, synthetic shapes,
and the multi-process (gloo, world_size=2) utterance-sharding logic."""
import os
import re
import socket
import ctypes

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from vibevoice_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "vvhip.h")).read()
    declared = set(re.findall(r"\b(vv_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 25
    so = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared:
        assert hasattr(so, s), s
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    _lib.load()


def test_binary_carries_the_hash_of_its_sources(tmp_path):
    """libvvhip.so embeds sha256[:16] of the sources + flags it was compiled from (vv_build_id); build.stale() and the loader
    compare it with the sources beside the binary -- content, not mtime -- so a stale in-tree .so is rebuilt or refused."""
    import shutil
    from vibevoice_amd import _lib, build
    build.build()
    sid = build.source_id()
    assert build.binary_id() == sid and not build.stale()
    assert _lib.load().vv_build_id() == b"VVHIP_BUILD_ID=" + sid.encode()
    # an edited source changes the id: checked on a scratch copy of the tree (nothing is rebuilt here)
    src = os.path.join(tmp_path, "csrc")
    shutil.copytree(build.CSRC, src)
    old_csrc, old_hdr = build.CSRC, build.HEADERS
    try:
        build.CSRC = src
        build.HEADERS = [os.path.join(src, "vv_common.h"), old_hdr[1]]
        assert build.source_id() == sid
        with open(os.path.join(src, "misc.hip"), "a") as f:
            f.write("\n// edited\n")
        assert build.source_id() != sid and build.stale()
    finally:
        build.CSRC, build.HEADERS = old_csrc, old_hdr


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vibevoice_amd.engine import Engine, EngineConfig, EngineError
    with pytest.raises(EngineError):
        Engine(EngineConfig(lm_hidden=128, lm_layers=1, lm_heads=2, lm_kv_heads=1, lm_inter=256, lm_vocab=100))


@pytest.mark.parametrize("n", [5, 10, 20])
def test_schedule_table_matches_oracle_stepper(n):
    from oracle import dpm
    from vibevoice_amd import schedule
    tv, coef = schedule.make_table(n)
    s = dpm.Schedule(n)
    assert np.array_equal(tv, s.timesteps.float().numpy())
    assert np.isfinite(coef).all()
    torch.manual_seed(n)
    x = torch.randn(3, 64)
    xo = x.clone()
    st = dpm.DPMState(s)
    x0p = torch.zeros_like(x)
    for i in range(n):
        v = torch.randn(3, 64)
        xo = st.step(v, xo)
        a, sg, cs, c0, c1 = [torch.tensor(c) for c in coef[i]]
        x0 = a * x - sg * v
        x = cs * x + c0 * x0 + c1 * (x0 - x0p)
        x0p = x0
    assert (x - xo).abs().max() < 5e-6
    tvb, _ = schedule.make_table(10, t_cast_bf16=True)
    assert tvb[0] == 1000.0        # 999 -> 1000 under the reference's bf16 cast (modeling_vibevoice_inference.py:705)


def test_param_shapes_match_survey_counts():
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.synthetic import param_shapes

    def count(cfg, prefix):
        return sum(int(np.prod(s)) for k, s in param_shapes(cfg).items() if k.startswith(prefix))
    c15, c7 = CONFIGS["1.5b"], CONFIGS["7b"]
    assert count(c15, "model.language_model.layers.") == 1_310_339_072
    assert count(c7, "model.language_model.layers.") == 6_525_618_176
    assert count(c15, "model.prediction_head.") == 123_279_360
    assert count(c7, "model.prediction_head.") == 669_333_504
    assert count(c15, "model.acoustic_tokenizer.decoder.") == 343_695_969
    assert count(c15, "model.acoustic_tokenizer.encoder.") == 343_696_032
    assert count(c15, "model.semantic_tokenizer.encoder.") == 344_613_600
    assert "lm_head.weight" in param_shapes(c7) and "lm_head.weight" not in param_shapes(c15)


def test_engine_config_from_reference_json():
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import engine_config_from_reference
    e = engine_config_from_reference(CONFIGS["7b"], n_slots=2)
    assert (e.lm_hidden, e.lm_heads, e.lm_kv_heads, e.lm_head_dim, e.lm_inter) == (3584, 28, 4, 128, 18944)
    assert e.head_ffn == 3 * 3584 and e.hop == 3200 and e.max_ctx == 32768 and e.n_slots == 2
    bad = dict(CONFIGS["7b"])
    bad["acoustic_tokenizer_config"] = dict(bad["acoustic_tokenizer_config"], mixer_layer="conv")
    with pytest.raises(ValueError):
        engine_config_from_reference(bad)


def test_param_name_mapping():
    from vibevoice_amd.engine import map_param_name
    assert map_param_name("model.language_model.layers.3.mlp.up_proj.weight") == "lm.layers.3.mlp.up_proj.weight"
    assert map_param_name("model.acoustic_tokenizer.decoder.head.conv.conv.bias") == "dec.head.conv.conv.bias"
    assert map_param_name("model.semantic_tokenizer.encoder.stages.0.0.gamma") == "senc.stages.0.0.gamma"
    assert map_param_name("lm_head.weight") == "lm_head.weight"
    assert map_param_name("model.speech_scaling_factor") is None


def test_forced_schedule_and_inputs_layout():
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    T = synthetic.TOKENS
    f = synthetic.forced_schedule(310, turn=150)
    assert f[:150] == [T.speech_diffusion_id] * 150 and f[150:152] == [T.speech_end_id, T.speech_start_id]
    inp = synthetic.synthetic_inputs(CONFIGS["1.5b"], n_speakers=2, text_tokens=50, voice_frames=7)
    assert inp["input_ids"][0, -1] == T.speech_start_id
    assert int(inp["speech_input_mask"].sum()) == 14 == int(inp["speech_masks"].sum())
    assert inp["speech_tensors"].shape == (2, 7 * 3200)


def test_shard_utterances_partition():
    from vibevoice_amd.parallel import shard_utterances
    costs = [5, 9, 1, 7, 3, 8, 2, 6, 4, 10, 11]
    for world in (1, 2, 3, 8):
        sh = shard_utterances(costs, world)
        assert sorted(i for s in sh for i in s) == list(range(len(costs)))
        loads = [sum(costs[i] for i in s) for s in sh]
        assert max(loads) - min(loads) <= max(costs)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, q):
    import torch.distributed as dist
    from vibevoice_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    shapes = [("a.weight", (4, 6)), ("b.bias", (5,)), ("c.weight", (3, 2, 2))]

    def make(name, shape):       # only ever called on rank 0
        assert dist.get_rank() == 0
        g = torch.Generator().manual_seed(len(name))
        return torch.randn(shape, generator=g)
    got = {k: v.clone() for k, v in parallel.broadcast_params(shapes, make, "cpu", torch.float32)}
    checksum = float(sum(v.double().sum() for v in got.values()))
    # each rank "decodes" its own shard of utterances: frames proportional to the cost
    costs = [4, 2, 6, 8]
    mine = parallel.shard_utterances(costs, world)[rank]
    units = float(sum(costs[i] for i in mine))
    total, wall = parallel.aggregate_throughput(units, 1.0 + rank, "cpu")
    q.put((rank, checksum, mine, total, wall))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_aggregate_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, m0, t0, w0), (r1, c1, m1, t1, w1) = res
    assert c0 == c1                       # identical weights on both ranks after the broadcast
    assert sorted(m0 + m1) == [0, 1, 2, 3] and not set(m0) & set(m1)
    assert t0 == t1 == 20.0 and w0 == w1 == 2.0    # sum of units, max of walls


def test_streaming_param_mapping_and_shapes():
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling_streaming import map_streaming_param_name
    from vibevoice_amd.synthetic import streaming_param_shapes
    cfg = CONFIGS["0.5b-streaming"]
    sh = streaming_param_shapes(cfg)
    n_tts = cfg["tts_backbone_num_hidden_layers"]
    n_lm = cfg["decoder_config"]["num_hidden_layers"] - n_tts
    assert (n_lm, n_tts) == (4, 20)
    assert sum(int(np.prod(v)) for k, v in sh.items() if "language_model.layers" in k) == 357_897_216
    names = {map_streaming_param_name(k, n_lm) for k in sh}
    assert "lm.layers.23.mlp.down_proj.weight" in names and "lm.layers.3.self_attn.q_proj.bias" in names
    assert {"lm.norm.weight", "tts_input_types.weight", "eos.fc1.weight", "eos.fc2.bias", "dec.head.conv.conv.weight"} <= names
    assert map_streaming_param_name("model.semantic_connector.fc1.weight", n_lm) is None


def test_streaming_oracle_windows_cpu():
    """The streaming oracle loop on a tiny CPU model: 7 text tokens -> windows of 5 and 2, six frames after each,
    then speech-only windows until the length cap."""
    import synth
    from oracle import generate_streaming as ogs
    from oracle import lm as olm
    cfg = synth.LMCfg(hidden=128, layers=2, heads=2, kv_heads=1, inter=256, vocab=320)
    w = synth.lm_weights(cfg)
    lm_w = {k: v for k, v in w.items() if k.startswith("embed") or k.startswith("layers.0.")}
    tts_w = {"norm.weight": w["norm.weight"], "embed_tokens.weight": w["embed_tokens.weight"]}
    tts_w.update({"layers.0." + k[len("layers.1."):]: v for k, v in w.items() if k.startswith("layers.1.")})
    mk = lambda ww: olm.Qwen2Oracle(ww, 1, cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.theta, cfg.eps)
    hc, cc = synth.HeadCfg(hidden=128, layers=1), synth.CodecCfg()
    g = synth.Gen(1)
    eos = {"fc1.weight": g.linear(128, 128), "fc1.bias": g.vec(128), "fc2.weight": g.linear(1, 128), "fc2.bias": g.vec(1, 0.1, -30.0)}
    m = ogs.StreamingOracleModel(lm=mk(lm_w), tts_lm=mk(tts_w), tts_types=g.normal((2, 128), 0.5, mat=False), eos=eos,
                                 head_w=synth.head_weights(hc), head_layers=1, ac_w=synth.decoder_weights(cc, 3),
                                 ac_conn=synth.connector_weights(64, 128, 4), ratios=cc.ratios, dec_depths=cc.dec_depths,
                                 scaling=0.2, bias=-0.05)
    pre = ogs.make_preset(m, torch.arange(9), 300)
    calls = []
    noise = lambda f, n2: (calls.append(f), torch.zeros(n2, 64))[1]
    n_tok, audio, reach, fin = ogs.oracle_generate_streaming(m, pre, torch.arange(7), 1.5, 3, noise, max_length=9 + 7 + 14)
    assert reach and not fin
    assert calls == list(range(len(calls))) and len(calls) == 15         # 6 + 6 + 3: the cap hits inside the third speech window
    assert audio.shape[-1] == 15 * 3200


# ---------------------------------------------------------------- LoRA merge at snapshot time
def test_lora_merge_math_and_key_mapping(tmp_path):
    import json as _json
    from safetensors.torch import save_file
    from vibevoice_amd import lora
    torch.manual_seed(3)
    base = {"model.language_model.layers.0.self_attn.q_proj.weight": torch.randn(12, 8),
            "model.language_model.layers.1.mlp.down_proj.weight": torch.randn(8, 20),
            "model.prediction_head.layers.0.ffn.gate_proj.weight": torch.randn(24, 8)}
    r = 2
    lm_sd, hd_sd = {}, {}
    for k, w in base.items():
        a, b = torch.randn(r, w.shape[1]), torch.randn(w.shape[0], r)
        if k.startswith(lora.LM_PREFIX):
            mod = k[len(lora.LM_PREFIX):-len(".weight")]
            lm_sd[f"base_model.model.{mod}.lora_A.weight"] = a
            lm_sd[f"base_model.model.{mod}.lora_B.weight"] = b
        else:
            mod = k[len(lora.HEAD_PREFIX):-len(".weight")]                 # peft saw the head through the shim's `base`
            hd_sd[f"base_model.model.base.{mod}.lora_A.default.weight"] = a
            hd_sd[f"base_model.model.base.{mod}.lora_B.default.weight"] = b
    root = tmp_path / "ckpt" / "lora"
    (root / "diffusion_head").mkdir(parents=True)
    (root / "acoustic_connector").mkdir()
    save_file(lm_sd, str(root / "adapter_model.safetensors"))
    (root / "adapter_config.json").write_text(_json.dumps({"r": r, "lora_alpha": 32}))
    torch.save(hd_sd, str(root / "diffusion_head" / "adapter_model.bin"))
    (root / "diffusion_head" / "adapter_config.json").write_text(_json.dumps({"r": r, "lora_alpha": 4, "use_rslora": True}))
    torch.save({"fc1.weight": torch.ones(8, 4)}, str(root / "acoustic_connector" / "pytorch_model.bin"))
    assert lora.resolve_adapter_root(str(tmp_path / "ckpt")) == str(root)
    ups = {k: (t, kind) for k, t, kind in lora.planned_updates(str(root), lambda k: base[k])}
    assert set(ups) == set(base) | {"model.acoustic_connector.fc1.weight"}
    for k, w in base.items():
        sd = lm_sd if k.startswith(lora.LM_PREFIX) else hd_sd
        scale = 32 / r if k.startswith(lora.LM_PREFIX) else 4 / r ** 0.5
        pairs = lora.lora_pairs(sd, lora.LM_PREFIX if k.startswith(lora.LM_PREFIX) else lora.HEAD_PREFIX,
                                strip="" if k.startswith(lora.LM_PREFIX) else "base.")
        A, B = pairs[k]
        assert torch.allclose(ups[k][0], w + scale * (B @ A), atol=1e-5)
    assert ups["model.acoustic_connector.fc1.weight"][1] == "acoustic_connector"
    with pytest.raises(ValueError):
        lora.merge_lora(torch.zeros(3, 3), torch.zeros(2, 4), torch.zeros(3, 2), 1.0)


def test_lora_merge_rounding_equals_the_peft_operations_on_a_bf16_model():
    """peft is not installed here; its merge is three tensor operations (peft/tuners/lora/layer.py: Linear.get_delta_weight, then
    `base_layer.weight.data += delta_weight`), replayed literally on a bf16 weight: with fp32 adapter matrices (peft's loader upcasts
    them, the reference's trainer saves fp32) the in-place add computes in fp32 and rounds once = merge_dtype "float32" bit for bit;
    with bf16 adapter matrices on the CPU get_delta_weight upcasts, multiplies, scales and rounds the DELTA to bf16 first = "bfloat16"."""
    from vibevoice_amd import lora
    g = torch.Generator().manual_seed(11)
    W = (torch.randn(96, 64, generator=g) * 0.02).to(torch.bfloat16)
    A, B = torch.randn(8, 64, generator=g) * 0.05, torch.randn(96, 8, generator=g) * 0.05
    scale = 32 / 8
    # (1) fp32 adapters: delta fp32, `+=` on the bf16 parameter
    w1 = W.clone()
    w1 += (B @ A) * scale
    assert w1.dtype == torch.bfloat16
    assert torch.equal(lora.merge_lora(W, A, B, scale, merge_dtype="float32"), w1)
    # (2) bf16 adapters on the CPU: cast_to_fp32 branch of get_delta_weight
    Ab, Bb = A.to(torch.bfloat16), B.to(torch.bfloat16)
    delta = ((Bb.float() @ Ab.float()) * scale).to(torch.bfloat16)
    w2 = W.clone()
    w2 += delta
    assert torch.equal(lora.merge_lora(W, Ab, Bb, scale, merge_dtype="bfloat16"), w2)
    assert not torch.equal(w1, w2)                                   # the two roundings do differ somewhere
    assert float((w1.float() - w2.float()).abs().max()) <= 2.0 ** -8 * float(w1.float().abs().max())       # by about one ulp
    with pytest.raises(ValueError):
        lora.merge_lora(W, A, B, scale, merge_dtype="fp16")


def test_streamer_surface_and_ordering_cpu():
    from vibevoice_amd.streamer import AudioStreamer
    s = AudioStreamer(batch_size=2, stop_signal=None, timeout=5.0)
    s.put(torch.full((2, 1, 4), 1.0), torch.tensor([0, 1]))
    s.put(torch.full((1, 1, 4), 2.0), torch.tensor([1]))
    s.end(torch.tensor([0]))
    s.put(torch.full((2, 1, 4), 3.0), torch.tensor([0, 1]))          # sample 0 already ended: dropped (streamer.py:52)
    s.end()
    assert s.finished_flags == [True, True]
    got0 = [c.flatten().tolist() for c in s.get_stream(0)]
    got1 = [c.flatten().tolist() for c in s.get_stream(1)]
    assert got0 == [[1.0] * 4]
    assert got1 == [[1.0] * 4, [2.0] * 4, [3.0] * 4]
    with pytest.raises(ValueError):
        s.get_stream(2)


# ---------------------------------------------------------------- the product's host loop vs the reference's generate()
@pytest.mark.parametrize("name", ["generate_forced_b1", "generate_forced_b2", "generate_greedy_b1", "generate_cap_b1", "generate_ragged_voice_b1"])
@pytest.mark.parametrize("speculate", [True, False])
def test_host_generate_loop_matches_reference_goldens(monkeypatch, name, speculate):
    """vibevoice_amd/modeling.py::generate -- the code that ships -- driven on CPU through tests/fake_engine.FakeEngine
    (every numeric stage delegated to the pinned oracle), against the goldens recorded from the reference's own
    generate(): identical token sequences / stop flags, waveform rel-L2 <= 1e-4.  Pins the host orchestration: row tables,
    negative-branch reset and fix-ups for desynchronised rows, speculative sampler (on and off), codec resets."""
    import types as _types
    import fake_engine
    from test_oracle_golden import G as GOLD, _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ids = torch.from_numpy(z["input_ids"])
    B = ids.shape[0]
    draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
    pre = (draws[0].reshape(B), draws[1].reshape(B, 3, 64))
    # the reference drew noise only on steps where some row emitted <speech_diffusion>: map those steps to the recorded draws
    seqs = z["sequences"]
    L0 = ids.shape[1]
    per_step, di = {}, 2
    for step in range(seqs.shape[1] - L0):
        if (seqs[:, L0 + step] == 303).any():
            per_step[step] = draws[di].reshape(-1, 64)
            di += 1
    assert di == len(draws)

    def noise_fn(step, n2):                       # a wrong speculative guess asks for a step that never diffused: discarded
        if step not in per_step:
            return torch.zeros(n2, 64)
        d = per_step[step]
        return d if d.shape[0] == n2 else torch.zeros(n2, 64)     # speculation assumes every active row diffuses
    forced = None
    if z["forced"].size:
        forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeEngine(_oracle_small(), n_slots=2)
        cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.speculate_sampling = speculate
        m.concurrent_codecs = False
        tok = _types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                                     bos_token_id=None, pad_token_id=305)
        class RecStreamer:                  # same recorder the golden run used (make_golden.py)
            def __init__(self, batch):
                self.finished_flags = [False] * batch
                self.log = []

            def put(self, chunk, idx):
                self.log.append([0, int(chunk.shape[0])] + [int(i) for i in idx.tolist()])

            def end(self, idx=None):
                ids_ = list(range(len(self.finished_flags))) if idx is None else [int(i) for i in idx.tolist()]
                self.log.append([1, len(ids_)] + ids_)
        rec = RecStreamer(B) if "streamer_log" in z.files else None
        out = m.generate(input_ids=ids, attention_mask=torch.from_numpy(z["attention_mask"]),
                         speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                         speech_input_mask=torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, tokenizer=tok,
                         max_new_tokens={"generate_greedy_b1": 10, "generate_cap_b1": 6}.get(name), generation_config={"do_sample": False},
                         _forced_tokens=forced, _noise_fn=noise_fn, _prefill_noise=pre, show_progress_bar=False, audio_streamer=rec)
    if rec is not None:                       # the AudioStreamer sees the same put / end calls, in the same order, as from the reference
        ref_log = [[int(v) for v in row if v >= 0] for row in z["streamer_log"]]
        assert rec.log == ref_log, (rec.log, ref_log)
    assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
    assert torch.equal(out.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        if ref.numel() == 0:
            assert out.speech_outputs[b] is None
            continue
        got = out.speech_outputs[b].reshape(-1)
        assert got.shape == ref.shape
        err = float((got - ref).norm() / ref.norm())
        assert err <= 1e-4, err


@pytest.mark.parametrize("name", ["streaming_text12_cap40", "streaming_text3_cap20", "streaming_eos", "streaming_cap_on_text"])
def test_host_streaming_loop_matches_reference_goldens(monkeypatch, name):
    """vibevoice_amd/modeling_streaming.py::generate on CPU through FakeStreamingEngine (oracle arithmetic), started from
    the prefilled branches the reference produced, against the goldens recorded from the reference's streaming
    generate(): token count, stop flag, waveform rel-L2 <= 1e-4."""
    import types as _types
    import fake_engine
    from test_oracle_golden import G as GOLD, _oracle_streaming_small
    from vibevoice_amd.modeling_streaming import VibeVoiceStreamingForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, name + ".npz"))
    draws = [torch.from_numpy(z[f"draw_{i}"]).reshape(2, 64) for i in range(int(z["n_draws"]))]

    def branch(tag):
        n = int(z[f"{tag}_layers"])
        kv = [(torch.from_numpy(z[f"{tag}_k{li}"])[None], torch.from_numpy(z[f"{tag}_v{li}"])[None]) for li in range(n)]
        L = kv[0][0].shape[2]
        hid = torch.zeros(1, L, 128)
        hid[0, -1] = torch.from_numpy(z[f"{tag}_last"])
        return _types.SimpleNamespace(past_key_values=kv, last_hidden_state=hid)
    pre = {"lm": branch("lm"), "tts_lm": branch("tts"), "neg_lm": None, "neg_tts_lm": branch("neg_tts")}
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeStreamingEngine(_oracle_streaming_small(eos_bias=float(z["eos_bias"]) if name in ("streaming_eos", "streaming_cap_on_text") else None), 1, 2)
        cfgd = {"decoder_config": {"max_position_embeddings": 512}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "tts_backbone_num_hidden_layers": 2}
        m = VibeVoiceStreamingForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        out = m.generate(tts_text_ids=torch.from_numpy(z["text"])[None], all_prefilled_outputs=pre, cfg_scale=1.5,
                         max_new_tokens=int(z["max_new"]), _noise_fn=lambda frame, n2: draws[frame])
        # the same call as the reference demo makes it: the processor's prompt ids go in, an AudioStreamer listens
        log = []

        class RecStreamer:
            finished_flags = [False]

            def put(self, chunk, idx):
                log.append([0, int(chunk.shape[0]), int(chunk.shape[-1])] + [int(i) for i in idx.tolist()])

            def end(self, idx=None):
                ids = [0] if idx is None else [int(i) for i in idx.tolist()]
                log.append([1, len(ids), -1 if idx is None else 0] + ids)
        eng.caches = {0: eng.om.lm.new_cache(), 1: eng.om.tts_lm.new_cache(), 2: eng.om.tts_lm.new_cache()}
        out2 = m.generate(tts_text_ids=torch.from_numpy(z["text"])[None], tts_lm_input_ids=torch.from_numpy(z["prompt"])[None],
                          all_prefilled_outputs=pre, cfg_scale=1.5, max_new_tokens=int(z["max_new"]), audio_streamer=RecStreamer(),
                          _noise_fn=lambda frame, n2: draws[frame])
    # sequences = prompt + consumed text + one id per speech token, exactly the reference's tts_lm_input_ids (:722); the
    # streamer sees the reference's put / end calls in the reference's order (a chunk per frame -- also after EOS inside a
    # window -- end(idx) whenever the EOS head fires, a final end())
    assert torch.equal(out2.sequences[0], torch.from_numpy(z["sequences"]))
    assert log == z["streamer_log"].tolist(), (log, z["streamer_log"].tolist())
    assert float((out2.speech_outputs[0] - out.speech_outputs[0]).norm() / out.speech_outputs[0].norm()) <= 1e-5
    assert int(z["prompt"].shape[0]) + out.sequences.shape[1] == int(z["n_tokens"])
    assert bool(out.reach_max_step_sample[0]) == bool(z["reach_max"][0])
    ref = torch.from_numpy(z["audio"])
    got = out.speech_outputs[0].reshape(-1)
    assert got.shape == ref.shape
    err = float((got - ref).norm() / ref.norm())
    assert err <= 1e-4, err


def test_bench_algorithmic_bytes_match_the_survey_figures():
    """bench.py's roofline numerator = SURVEY 8(d)'s per-frame formula: 1.5B N=10 L~400 -> 6.44 GB, 7B -> 27.7 GB,
    7B N=20 L=32K -> 42.65 GB."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from vibevoice_amd.configs import CONFIGS
    gb = lambda m, n, lp, ln: bench.algorithmic_bytes_per_frame(CONFIGS[m], n, lp, ln) / 1e9
    assert abs(gb("1.5b", 10, 400, 75) - 6.44) < 0.03
    assert abs(gb("7b", 10, 400, 75) - 27.7) < 0.1
    assert abs(gb("7b", 20, 32000, 75) - 42.65) < 0.3


def test_host_generate_sampling_keeps_the_reference_rng_stream(monkeypatch):
    """do_sample=True, nothing injected: tokens come from torch.multinomial, prefill and diffusion noise from torch.randn,
    all on the global generator.  Seeded like the reference run that produced generate_sampled_b1.npz, the product's host
    loop must consume the generator in the same order and shapes -> identical tokens and waveform."""
    import types as _types
    import fake_engine
    from test_oracle_golden import G as GOLD, _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, "generate_sampled_b1.npz"))
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeEngine(_oracle_small(), n_slots=1)
        cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        tok = _types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                                     bos_token_id=None, pad_token_id=305)
        torch.manual_seed(int(z["seed"]))
        out = m.generate(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                         speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                         speech_input_mask=torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, tokenizer=tok,
                         max_new_tokens=14, generation_config={"do_sample": True, "top_k": 0}, show_progress_bar=False)
    assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
    ref = torch.from_numpy(z["audio_0"])
    got = out.speech_outputs[0].reshape(-1)
    assert got.shape == ref.shape
    assert float((got - ref).norm() / ref.norm()) <= 1e-4


@pytest.mark.parametrize("name,gen_cfg,max_new", [
    ("generate_sampled_warped_b1.npz", {"do_sample": True, "top_k": 300, "top_p": 0.995, "min_p": 0.0001, "temperature": 0.8,
                                        "repetition_penalty": 1.15}, 14),
    ("generate_sampled_warped_b2.npz", {"do_sample": True, "top_k": 310, "top_p": 0.998, "temperature": 1.3, "repetition_penalty": 1.05}, 12),
    ("generate_greedy_reppen_b1.npz", {"do_sample": False, "repetition_penalty": 4.0}, 10),
])
def test_host_generate_full_vocabulary_processors_match_the_reference(monkeypatch, name, gen_cfg, max_new):
    """The reference's generate() with HF's full-vocabulary logits processors in front of its valid-token constraint (repetition
    penalty; with do_sample: temperature, top-k, top-p, min-p -- modeling_vibevoice_inference.py:310-319,416-419,488-496), seeded
    and recorded by tests/golden/make_golden.py.  The product routes such calls through vv_lm_logits_full (here: the fake engine's
    fp32 lm_head) and its own implementation of the processors: identical token sequences (the multinomial draws see the same
    distributions on the same generator) and waveforms, for one utterance and for a left-padded batch of two."""
    import types as _types
    import fake_engine
    from test_oracle_golden import G as GOLD, _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, name))
    B = z["input_ids"].shape[0]
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeEngine(_oracle_small(), n_slots=B)
        cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        tok = _types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                                     bos_token_id=None, pad_token_id=305)
        torch.manual_seed(int(z["seed"]))
        out = m.generate(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                         speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                         speech_input_mask=torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, tokenizer=tok,
                         max_new_tokens=max_new, generation_config=gen_cfg, show_progress_bar=False)
    assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"])), (out.sequences.tolist(), z["sequences"].tolist())
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        if ref.numel() == 0:
            assert out.speech_outputs[b] is None or out.speech_outputs[b].numel() == 0
            continue
        got = out.speech_outputs[b].reshape(-1)
        assert got.shape == ref.shape
        assert float((got - ref).norm() / ref.norm()) <= 1e-4


def test_full_vocabulary_processors_equal_the_transformers_classes(monkeypatch):
    """vibevoice_amd/modeling.py::_full_vocab_scores against the classes the reference gets from
    GenerationMixin._get_logits_processor (repetition penalty, temperature, top-k, top-p, min-p, in that order), on the fake
    engine's fp32 lm_head: the same kept / filtered pattern and the same surviving scores for 7 rows with different histories."""
    import types as _types
    import fake_engine
    from test_oracle_golden import _oracle_small
    from transformers.generation.logits_process import (MinPLogitsWarper, RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    om = _oracle_small()
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeEngine(om, n_slots=1)
        cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        V, H = om.lm_head.shape
        g = torch.Generator().manual_seed(77)
        n = 7
        hid = torch.randn(n, H, generator=g) * 2.0
        order = []
        for i in range(n):
            ids = torch.randint(0, V, (8 + i,), generator=g).tolist()
            order.append(_types.SimpleNamespace(idx=i, ids=ids, tokens=torch.randint(0, V, (4,), generator=g).tolist(),
                                                seq_len0=40 if i % 2 else len(ids), init_len=len(ids)))
        warp = dict(top_k=180, top_p=0.9, min_p=0.003, repetition_penalty=1.25)
        got = m._full_vocab_scores(hid, order, dict(warp=warp, do_sample=True, temperature=0.6, pad_id=V - 2))
    want = torch.nn.functional.linear(hid, om.lm_head)
    for i, u in enumerate(order):
        row = ([V - 2] if u.seq_len0 > u.init_len else []) + u.ids + u.tokens
        want[i:i + 1] = RepetitionPenaltyLogitsProcessor(1.25)(torch.tensor([row]), want[i:i + 1])
    for proc in (TemperatureLogitsWarper(0.6), TopKLogitsWarper(180), TopPLogitsWarper(0.9), MinPLogitsWarper(0.003)):
        want = proc(None, want)
    assert torch.equal(torch.isfinite(got), torch.isfinite(want))
    keep = torch.isfinite(want)
    assert torch.allclose(got[keep], want[keep], rtol=1e-6, atol=1e-6)
    assert int(keep.sum(dim=-1).min()) >= 1 and int(keep.sum()) < n * V // 2       # the filters bite, and keep at least one id per row


def test_host_generate_reports_a_row_whose_valid_tokens_were_all_filtered(monkeypatch):
    """top_k = 1 over the toy vocabulary keeps one id per row -- with random weights not a speech token: the reference dies in
    torch.multinomial on NaN probabilities (tests/golden/make_golden.py walks its seed list for that reason); the product says why."""
    import types as _types
    import fake_engine
    from test_oracle_golden import G as GOLD, _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, "generate_sampled_b1.npz"))
    with fake_engine.cpu_cuda_shims(monkeypatch):
        eng = fake_engine.FakeEngine(_oracle_small(), n_slots=1)
        cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        tok = _types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                                     bos_token_id=None, pad_token_id=305)
        with pytest.raises(RuntimeError, match="removed every valid speech token"):
            m.generate(input_ids=torch.from_numpy(z["input_ids"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                       speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                       speech_input_mask=torch.from_numpy(z["speech_input_mask"]), cfg_scale=1.3, tokenizer=tok,
                       max_new_tokens=14, generation_config={"do_sample": True, "top_k": 1}, show_progress_bar=False)


def test_recorded_bench_line_keeps_the_driver_contract():
    """profiles/r06_bench_default.json is the stdout of `python bench.py` on an MI355X: the one JSON line the driver parses.  Its fields
    are the contract (metric / unit = BASELINE.json's, whole-job value, workload named in config, `roofline` with algorithmic bytes over
    measured launch time against the 8 TB/s HBM peak, `cpu_baseline` on a bounded sample AT THE TIMED KV LENGTH), the north-star target
    configuration rides under extra.configs with its own roofline and batch parity, and the numbers must be self-consistent."""
    import json
    with open(os.path.join(ROOT, "profiles", "r06_bench_default.json")) as f:
        d = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert d["metric"] in base["metric"] and d["unit"] == "audio-s/wall-s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and base["published"] == {}            # nothing published for this metric
    assert d["dtype"] == "bf16" and "7B" in d["config"]["workload"].upper() and "model" in d["config"]
    frames_per_s = d["value"] / (3200 / 24000)
    assert abs(frames_per_s * d["ms_per_step"] / 1e3 - 1.0) < 0.03            # 19 of 20 timed steps are frames (one control token)... within 3 %
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "vv_gemv_kernel"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert abs(r["achieved"] - r["bytes_per_launch"] / 1e9 / (r["avg_launch_us"] / 1e6)) / r["achieved"] < 0.01
    if r["traffic"] is not None:
        assert 0.9 < r["traffic"] / r["bytes_per_launch"] < 1.1              # PMC bytes ~ algorithmic bytes: no wasted re-reads
    assert r["launches_per_step"] * r["avg_launch_us"] / 1e3 < d["ms_per_step"]      # the chain fits inside the step
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == d["unit"] and c["value"] > 0 and "sample" in c
    assert abs(c["kv_length"] - d["config"]["kv_len_timed"]) <= 8 and "KV LENGTH" in c["sample"]      # the CPU window runs where the GPU leg is timed
    g = d["gpu_eager_baseline"]
    assert g["value"] > c["value"] and abs(g["speedup_of_this_path"] - d["value"] / g["value"]) < 0.05 and g["kv_length"] == c["kv_length"]
    assert d["parity"]["within_bounds"] and d["parity_long"]["within_bounds"]
    assert len(d["extra"]["libvvhip_build_id"]) == 16
    cf = d["extra"]["configs"]
    assert {"configs[1]", "configs[4]", "configs[3] per GPU"} <= set(cf)
    c3 = cf["configs[3] per GPU"]
    # the north-star target configuration: 8 utterances in lock-step on the GPU, its own roofline (batch kernels) and a batch-8 parity block
    assert c3["utterances_per_gpu"] == 8 and c3["n_gpus"] == 1 and "4 speaker" in c3["workload"] and "8 utterances per GPU" in c3["workload"]
    assert 0.94 <= c3["value"] * (c3["ms_per_step"] / 1e3) / (8 * 3200 / 24000) <= 1.01      # 8 frames per step (19 or 20 of the 20 timed steps are frames)
    assert c3["value"] > 5.0 * 8                                             # north_star: >= 5 x real time for every one of the 8 utterances
    assert c3["roofline"]["kernel"] == "vv_gemv16p_kernel" and c3["roofline"]["attention"]["frac"] > 0.5
    p3 = c3["parity"]
    assert p3["rows"] == 8 and p3["within_bounds"] and p3["vs_fp32"]["distinct_requests"] == 2 and p3["vs_fp32"]["latent"] <= 5e-2
    assert p3["vs_fp32"]["rows_of_one_request_latent_spread"] == 0.0


def test_bench_checkpoint_hook_resolves_model_directories(tmp_path, monkeypatch):
    """"if $VIBEVOICE_MODEL_DIR holds real checkpoints they are used instead" -- bench.find_checkpoint() resolves the
    released models' directory names (config.json + *.safetensors) or the root itself when its config has the model's decoder
    geometry; anything else falls back to synthetic weights (None)."""
    import importlib.util
    import json
    import os
    from safetensors.torch import save_file
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_hook_cpu", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from vibevoice_amd.configs import CONFIGS
    monkeypatch.delenv("VIBEVOICE_MODEL_DIR", raising=False)
    assert bench.find_checkpoint("7b") is None
    monkeypatch.setenv("VIBEVOICE_MODEL_DIR", str(tmp_path))
    assert bench.find_checkpoint("7b") is None                              # empty root
    d = tmp_path / "VibeVoice-Large"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(CONFIGS["7b"]))
    assert bench.find_checkpoint("7b") is None                              # no shards yet
    save_file({"model.speech_scaling_factor": torch.tensor(0.19)}, str(d / "model-00001-of-00001.safetensors"))
    assert bench.find_checkpoint("7b") == str(d) and bench.find_checkpoint("1.5b") is None
    got = dict(bench.checkpoint_tensors(str(d)))
    assert abs(float(got["model.speech_scaling_factor"]) - 0.19) < 1e-6
    # the root itself is a checkpoint of the model whose geometry its config states
    (tmp_path / "config.json").write_text(json.dumps(CONFIGS["1.5b"]))
    save_file({"x": torch.zeros(1)}, str(tmp_path / "model.safetensors"))
    assert bench.find_checkpoint("1.5b") == str(tmp_path) and bench.find_checkpoint("0.5b-streaming") is None


def test_engine_sync_refuses_captured_graphs_that_hold_memset_nodes(monkeypatch):
    """Round 6: a memset node of a replayed hipGraph filled with stale words on this runtime, so the library captures kernel
    launches only and Engine.sync() -- where every generate() ends -- raises when vv_stat(ctx, 5) says otherwise.  Host logic only: the
    engine object is built around a stub of the C library."""
    import pytest
    from vibevoice_amd.engine import Engine

    class Lib:
        def __init__(self, foreign):
            self.foreign = foreign

        def vv_check(self, ctx, stream):
            return 0

        def vv_stat(self, ctx, what):
            return {4: 0, 5: self.foreign}.get(what, 0)

        def vv_destroy(self, ctx):
            return None

    class Stream:
        cuda_stream = 0

        def synchronize(self):
            return None

    def make(foreign):
        e = object.__new__(Engine)
        e.stream, e.lib, e._ctx, e._fallback_warned = Stream(), Lib(foreign), 1, False
        return e

    make(0).sync()
    with pytest.raises(RuntimeError, match="memset / memcpy node"):
        make(2).sync()
    monkeypatch.setenv("VVHIP_ALLOW_FOREIGN_NODES", "1")          # the escape for A/B runs against library builds from before round 6
    make(2).sync()
