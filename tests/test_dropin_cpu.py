"""CPU tests of the drop-in surface: the reference demo's call sequence against the product class, the
attribute tree callers read, weight-snapshot handles, continuous admission, the streamers.  The numeric stages run
through tests/fake_engine (oracle arithmetic); what is under test is the product's host code."""
import asyncio
import json
import os
import threading
import time
import types

import numpy as np
import pytest
import torch

import fake_engine
import synth
from test_oracle_golden import G as GOLD

TOK = types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                            bos_token_id=None, pad_token_id=305)


def tiny_reference_config():
    """The tiny model of tests/gpu_util.build_small in the reference's config.json schema (vibevoice/configs/*.json)."""
    lm, cc = synth.LMCfg(), synth.CodecCfg()
    tok = {"causal": True, "channels": 1, "conv_bias": True, "conv_norm": "none", "encoder_depths": cc.depth_str,
           "encoder_n_filters": cc.n_filters, "encoder_ratios": cc.ratios, "layernorm": "RMSNorm", "layernorm_eps": cc.eps,
           "mixer_layer": "depthwise_conv", "pad_mode": "constant"}
    return {
        "acoustic_vae_dim": 64, "semantic_vae_dim": 128,
        "acoustic_tokenizer_config": dict(tok, decoder_depths=None, decoder_n_filters=cc.n_filters, decoder_ratios=cc.ratios,
                                          fix_std=0.5, std_dist_type="gaussian", vae_dim=64),
        "semantic_tokenizer_config": dict(tok, fix_std=0, std_dist_type="none", vae_dim=128),
        "diffusion_head_config": {"ddpm_num_inference_steps": 5, "head_ffn_ratio": 3.0, "head_layers": 2, "latent_size": 64,
                                  "rms_norm_eps": 1e-5, "hidden_size": lm.hidden},
        "decoder_config": {"hidden_size": lm.hidden, "intermediate_size": lm.inter, "num_attention_heads": lm.heads,
                           "num_key_value_heads": lm.kv_heads, "num_hidden_layers": lm.layers, "vocab_size": lm.vocab,
                           "max_position_embeddings": lm.max_pos, "rms_norm_eps": lm.eps, "rope_theta": lm.theta,
                           "model_type": "qwen2", "tie_word_embeddings": False},
    }


def tiny_reference_state_dict():
    """Same seeded weights as test_oracle_golden._oracle_small, under the reference checkpoint's key names."""
    lm = synth.LMCfg()
    hc = synth.HeadCfg(hidden=lm.hidden, layers=2)
    cc, sc = synth.CodecCfg(), synth.CodecCfg(vae_dim=128)
    sd = {"model.language_model." + k: v for k, v in synth.lm_weights(lm).items()}
    sd["lm_head.weight"] = synth.lm_head_weight(lm)
    sd.update({"model.prediction_head." + k: v for k, v in synth.head_weights(hc).items()})
    sd.update({"model.acoustic_tokenizer." + k: v for k, v in {**synth.encoder_weights(cc, 2), **synth.decoder_weights(cc, 3)}.items()})
    sd.update({"model.semantic_tokenizer." + k: v for k, v in synth.encoder_weights(sc, 7).items()})
    sd.update({"model.acoustic_connector." + k: v for k, v in synth.connector_weights(64, lm.hidden, 4).items()})
    sd.update({"model.semantic_connector." + k: v for k, v in synth.connector_weights(128, lm.hidden, 8).items()})
    sd["model.speech_scaling_factor"] = torch.tensor(0.2)
    sd["model.speech_bias_factor"] = torch.tensor(-0.05)
    return sd


@pytest.fixture()
def checkpoint_dir(tmp_path):
    from safetensors.torch import save_file
    d = tmp_path / "tiny-vibevoice"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(tiny_reference_config()))
    sd = {k: v.contiguous() for k, v in tiny_reference_state_dict().items()}
    keys = sorted(sd)
    save_file({k: sd[k] for k in keys[::2]}, str(d / "model-00001-of-00002.safetensors"))       # two shards, like the converter
    save_file({k: sd[k] for k in keys[1::2]}, str(d / "model-00002-of-00002.safetensors"))
    return str(d)


@pytest.fixture()
def product(monkeypatch, checkpoint_dir):
    """the product class loaded from a checkpoint directory, its Engine replaced by the oracle-backed fake"""
    from vibevoice_amd import modeling
    with fake_engine.cpu_cuda_shims(monkeypatch):
        monkeypatch.setattr(modeling, "Engine", fake_engine.LoadableFakeEngine)
        yield modeling, checkpoint_dir


def test_reference_demo_call_sequence(product, capsys):
    """demo/inference_from_file.py:297-431 replayed line by line: from_pretrained(path, torch_dtype, device_map,
    attn_implementation) -> eval() -> set_ddpm_inference_steps(num_steps=10) -> the attribute read at :367-368 ->
    processor-shaped inputs moved with .to(device) -> generate(**inputs, max_new_tokens=None, cfg_scale, tokenizer,
    generation_config={'do_sample': False}, verbose=True, is_prefill=True) -> the fields the demo reads (:401-431).
    Then the same call seeded with do_sample=True must land on the golden recorded from the reference's own generate()."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(
        path, torch_dtype=torch.float32, device_map="cuda", attn_implementation="flash_attention_2")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=10)
    assert model.ddpm_inference_steps == 10
    assert hasattr(model.model, "language_model")
    print(f"Language model attention: {model.model.language_model.config._attn_implementation}")     # :367-368
    assert "Language model attention: vvhip" in capsys.readouterr().out
    assert model.requested_attn_implementation == "flash_attention_2"
    # reference properties (:87-117) and what lora_loading.py touches (:88-131,163-169)
    assert model.prediction_head is model.model.prediction_head and model.acoustic_connector is model.model.acoustic_connector
    assert model.semantic_connector is model.model.semantic_connector and model.acoustic_tokenizer is model.model.acoustic_tokenizer
    assert abs(float(model.speech_scaling_factor) - 0.2) < 1e-7 and abs(float(model.model.speech_bias_factor) + 0.05) < 1e-7
    assert next(model.parameters()).device == model.device
    assert model.config.decoder_config.hidden_size == 128 and model.config.diffusion_head_config.ddpm_num_inference_steps == 5
    assert list(model.noise_scheduler.timesteps[:2]) == [999, 899]

    z = np.load(os.path.join(GOLD, "generate_sampled_b1.npz"))
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"]),
              "parsed_scripts": [[(0, "hello")]], "all_speakers_list": [[0]]}            # what BatchEncoding carries (:374-404)
    for k, v in inputs.items():
        if torch.is_tensor(v):
            inputs[k] = v.to("cpu")                                                       # the demo's .to(target_device)
    model.set_ddpm_inference_steps(num_steps=5)
    outputs = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK,
                             generation_config={"do_sample": False}, verbose=True, is_prefill=True)
    assert outputs.speech_outputs and outputs.speech_outputs[0] is not None
    audio_samples = outputs.speech_outputs[0].shape[-1]
    assert audio_samples % 3200 == 0 and audio_samples > 0
    input_tokens = inputs["input_ids"].shape[1]
    assert outputs.sequences.shape[1] - input_tokens > 0
    assert outputs.reach_max_step_sample.shape == (1,)
    # seeded sampling run == the reference's own generate() (tests/golden/generate_sampled_b1.npz)
    torch.manual_seed(int(z["seed"]))
    out2 = model.generate(**inputs, max_new_tokens=14, cfg_scale=1.3, tokenizer=TOK, generation_config={"do_sample": True, "top_k": 0},
                          verbose=False, is_prefill=True, show_progress_bar=False)
    assert torch.equal(out2.sequences.cpu(), torch.from_numpy(z["sequences"]))
    ref = torch.from_numpy(z["audio_0"])
    got = out2.speech_outputs[0].reshape(-1)
    assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4


def test_generation_config_forms(product):
    modeling, path = product
    m = modeling.VibeVoiceForConditionalGenerationInference
    assert m._generation_options(None) == (False, 1.0, None)
    assert m._generation_options({"do_sample": True, "temperature": 0.7, "top_k": 0}) == (True, 0.7, None)
    # transformers==4.51.3 (the reference's pin): do_sample alone means top-50 over the whole vocabulary
    assert m._generation_options({"do_sample": True}) == (True, 1.0, dict(top_k=50, top_p=1.0, min_p=0.0, repetition_penalty=1.0))
    assert m._generation_options({"do_sample": True, "top_k": 10, "top_p": 0.9})[2] == dict(top_k=10, top_p=0.9, min_p=0.0, repetition_penalty=1.0)
    assert m._generation_options({"do_sample": False, "repetition_penalty": 1.2})[2] == dict(top_k=0, top_p=1.0, min_p=0.0, repetition_penalty=1.2)
    assert m._generation_options({"do_sample": False, "top_k": 10}) == (False, 1.0, None)        # warpers only act with sampling
    from transformers import GenerationConfig
    assert m._generation_options(GenerationConfig(do_sample=True, temperature=0.5, top_k=0)) == (True, 0.5, None)      # an object, not a dict
    for bad in ({"do_sample": True, "typical_p": 0.5}, {"num_beams": 4}, {"no_repeat_ngram_size": 3}):
        with pytest.raises(NotImplementedError):
            m._generation_options(bad)
    with pytest.raises(ValueError):
        m._generation_options({"do_sample": True, "top_p": 1.5})
    with pytest.raises(TypeError):
        m._generation_options(3)


def test_weight_handles_take_the_reference_lora_loader_calls(product, tmp_path):
    """What lora_loading.py:71-84,112-131 does with `model.model.<component>`: load_state_dict(strict=False) + .to(device)
    must reach the engine's weight snapshot; vibevoice_amd.load_lora_assets (merge at snapshot time) goes through the same."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
    eng = model.engine
    n0 = len(eng.uploads)
    new_fc1 = torch.full((128, 64), 0.25)
    res = model.model.acoustic_connector.load_state_dict({"fc1.weight": new_fc1, "bogus": torch.zeros(1)}, strict=False)
    model.model.acoustic_connector.to(torch.device("cpu"))
    assert eng.uploads[n0:] == ["ac_conn.fc1.weight"] and torch.equal(eng._w["ac_conn.fc1.weight"], new_fc1)
    assert res.unexpected_keys == ["bogus"]
    with pytest.raises(RuntimeError):
        model.model.prediction_head.load_state_dict({"nope": torch.zeros(1)}, strict=True)
    # LoRA merged into the packed weights from the checkpoint's own base tensors (model.base_tensor)
    from safetensors.torch import save_file
    from vibevoice_amd import load_lora_assets
    root = tmp_path / "ft" / "lora"
    root.mkdir(parents=True)
    key = "model.language_model.layers.0.self_attn.q_proj.weight"
    base = model.base_tensor(key)
    a, b = torch.randn(2, base.shape[1]), torch.randn(base.shape[0], 2)
    save_file({"base_model.model.layers.0.self_attn.q_proj.lora_A.weight": a, "base_model.model.layers.0.self_attn.q_proj.lora_B.weight": b},
              str(root / "adapter_model.safetensors"))
    (root / "adapter_config.json").write_text(json.dumps({"r": 2, "lora_alpha": 4}))
    rep = load_lora_assets(model, str(tmp_path / "ft"))
    assert rep.language_model and rep.merged_tensors == 1
    assert torch.allclose(eng._w["lm.layers.0.self_attn.q_proj.weight"], base + 2.0 * (b @ a), atol=1e-5)


def _requests(n, seed):
    """n single-utterance requests with different prompt lengths, forced token plans that end at different steps"""
    g = synth.Gen(seed)
    D, E, S, X = TOK.speech_diffusion_id, TOK.speech_end_id, TOK.speech_start_id, TOK.eos_token_id
    plans = [[D, D, D, X], [D, E, S, D, D, D, D, X], [D, D, X], [D, D, D, D, D, E, X], [D, X], [D, D, D, E, S, D, X]]
    reqs = []
    for i in range(n):
        L = 9 + 3 * (i % 4)
        ids = torch.from_numpy(g.rng.integers(0, 300, (1, L)))
        ids[0, -1] = S
        noise = {s: synth.Gen(1000 * (seed + i) + s).normal((2, 64), 1.0, mat=False) for s in range(16)}
        reqs.append({"input_ids": ids, "attention_mask": torch.ones_like(ids), "_forced_tokens": plans[i % len(plans)],
                     "_noise_fn": (lambda nz: (lambda step, n2: nz[step]))(noise)})
    return reqs


def test_continuous_admission_equals_one_by_one(monkeypatch):
    """generate_continuous(): 6 queued utterances over 2 slots.  A slot freed by EOS is refilled on the next iteration while
    the other utterance keeps decoding (the batch is never drained), and every utterance comes out exactly as generate()
    produces it alone (tokens identical, waveform rel-L2 <= 1e-5 through the oracle-backed engine)."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    reqs = _requests(6, 3)
    with fake_engine.cpu_cuda_shims(monkeypatch):
        solo = []
        for r in reqs:
            m1 = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=1), model_dtype=torch.float32)
            m1.set_speech_factors(0.2, -0.05)
            m1.set_ddpm_inference_steps(5)
            solo.append(m1.generate(input_ids=r["input_ids"], attention_mask=r["attention_mask"], cfg_scale=1.3, tokenizer=TOK,
                                    generation_config={"do_sample": False}, _forced_tokens=[r["_forced_tokens"]],
                                    _noise_fn=r["_noise_fn"], show_progress_bar=False))
        eng = fake_engine.FakeEngine(_oracle_small(), n_slots=2)
        m = VibeVoiceForConditionalGenerationInference(cfgd, eng, model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        outs = m.generate_continuous(reqs, tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
    assert len(outs) == 6
    for a, b in zip(outs, solo):
        assert torch.equal(a.sequences.cpu(), b.sequences.cpu())
        assert torch.equal(a.reach_max_step_sample.cpu(), b.reach_max_step_sample.cpu())
        assert (a.speech_outputs[0] is None) == (b.speech_outputs[0] is None)
        if b.speech_outputs[0] is not None:
            assert a.speech_outputs[0].shape == b.speech_outputs[0].shape
            d = (a.speech_outputs[0] - b.speech_outputs[0]).norm() / b.speech_outputs[0].norm()
            assert float(d) <= 1e-5, float(d)         # the CPU BLAS blocks a 2-row and a 1-row matmul differently; nothing else differs
    st = m.last_stats
    assert st["max_in_flight"] == 2
    adm = st["admissions"]
    assert [a[1] for a in adm] == list(range(6))                    # queue order
    assert adm[0][0] == 0 and adm[1][0] == 0 and all(a[0] > 0 for a in adm[2:])
    # no drain: the total number of iterations is far below the sum of the utterances' own lengths
    own = [len(r["_forced_tokens"]) for r in reqs]
    assert st["iterations"] < sum(own) and st["iterations"] >= max(own)


def test_interleaved_lanes_draw_from_their_own_generators(monkeypatch):
    """Random draws under generate_interleaved(): every lane owns a CPU / device generator seeded from the global CPU generator on the
    caller's thread.  A seeded call is reproducible although the lanes run on racing host threads, a wrong speculative guess in one lane
    (plans with <speech_end> right after a frame: the speculative noise draw is undone) rewinds that lane's generator only, and the
    process-global generator moves by exactly the seed draw -- nothing a lane does reaches it."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    reqs = [{k: v for k, v in r.items() if k != "_noise_fn"} for r in _requests(6, 11)]     # no injected noise: torch.randn draws
    reqs[1]["input_ids"] = reqs[0]["input_ids"].clone()                 # two identical requests (they land in different lanes)
    reqs[1]["attention_mask"] = reqs[0]["attention_mask"].clone()
    reqs[1]["_forced_tokens"] = list(reqs[0]["_forced_tokens"])
    with fake_engine.cpu_cuda_shims(monkeypatch):
        monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=2), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        m.speculate_sampling = True
        kw = dict(tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
        runs, states = [], []
        for _ in range(3):
            torch.manual_seed(77)
            runs.append(m.generate_interleaved(reqs, lanes=2, **kw))
            states.append(torch.get_rng_state())
        torch.manual_seed(77)
        torch.randint(0, 2 ** 62, (2, 2), dtype=torch.int64)
        only_the_seed_draw = torch.get_rng_state()
        shards = m.last_stats["shards"]
        m.close_lanes()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a.sequences, b.sequences)
            # same draws -> the same waveform up to the CPU BLAS's thread-dependent summation order (another stream would be O(1) away)
            assert float((a.speech_outputs[0] - b.speech_outputs[0]).norm() / b.speech_outputs[0].norm()) <= 1e-5
    assert all(torch.equal(s, only_the_seed_draw) for s in states)
    # identical requests in different lanes hear different noise (each lane has its own stream)
    lane_of = {i: k for k, sh in enumerate(shards) for i in sh}
    if lane_of[0] != lane_of[1]:
        a, b = runs[0][0].speech_outputs[0], runs[0][1].speech_outputs[0]
        assert float((a - b).norm() / b.norm()) > 1e-2


def test_interleaved_lanes_host_logic(monkeypatch):
    """generate_interleaved(): the queue over two engine contexts sharing one weight copy, one host thread per lane (here: two
    oracle-backed CPU engines over one oracle model).  Host logic under test: longest-prompt-first split, request order of the result,
    the caller's streamer seeing every request under its own sample index with one end per request, the lanes kept for the next call,
    an exception inside a lane surfacing in the caller after the streamer has been closed."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    reqs = _requests(5, 9)

    class Rec:
        def __init__(self, n):
            self.finished_flags, self.puts, self.ends = [False] * n, [], []

        def put(self, chunk, idx):
            self.puts.append(idx.tolist())

        def end(self, idx=None):
            self.ends.append(None if idx is None else idx.tolist())
    with fake_engine.cpu_cuda_shims(monkeypatch):
        monkeypatch.setattr(torch.cuda, "set_device", lambda *a, **k: None)
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=2), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        kw = dict(tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
        solo = m.generate_continuous(reqs, **kw)
        rec = Rec(5)
        outs = m.generate_interleaved(reqs, lanes=2, audio_streamer=rec, **kw)
        assert m.last_stats["lanes"] == 2 and sorted(i for sh in m.last_stats["shards"] for i in sh) == [0, 1, 2, 3, 4]
        assert m.last_stats["frames"] == sum(o.speech_outputs[0].shape[-1] // 3200 for o in solo)
        for a, b in zip(outs, solo):
            assert torch.equal(a.sequences, b.sequences)
            assert float((a.speech_outputs[0] - b.speech_outputs[0]).norm() / b.speech_outputs[0].norm()) <= 1e-5
        n_put = [sum(p.count(i) for p in rec.puts) for i in range(5)]
        assert [n * 3200 for n in n_put] == [o.speech_outputs[0].shape[-1] for o in solo]
        assert sorted(i for e in rec.ends if e is not None for i in e) == [0, 1, 2, 3, 4] and rec.ends[-1] is None
        lane = m._lanes[0]
        assert lane.engine.shared_from is m.engine and lane._scaling == m._scaling and lane.ddpm_inference_steps == 5
        m.generate_interleaved(reqs[:2], lanes=2, **kw)
        assert m._lanes[0] is lane
        # one lane fails: the exception reaches the caller, the streamer is closed first
        bad = [dict(r) for r in reqs]
        bad[3]["input_ids"] = torch.cat([reqs[3]["input_ids"], reqs[3]["input_ids"]], 0)       # two rows in one request: refused
        rec2 = Rec(5)
        with pytest.raises(ValueError, match="exactly one utterance"):
            m.generate_interleaved(bad, lanes=2, audio_streamer=rec2, **kw)
        assert rec2.ends and rec2.ends[-1] is None
        # the multi-GPU entry point with lanes on each rank (no process group here: one rank, its shard = the whole queue)
        from vibevoice_amd import parallel
        st = {}
        sharded = parallel.generate_sharded(m, reqs, stats=st, lanes=2, **kw)
        assert st["utterances_per_rank"] == [5] and m.last_stats["lanes"] == 2
        for a, b in zip(sharded, solo):
            assert torch.equal(a.sequences, b.sequences.cpu())
            assert float((a.speech_outputs[0] - b.speech_outputs[0].float().cpu()).norm() / b.speech_outputs[0].norm()) <= 1e-5
        m.close_lanes()
        assert lane.engine.closed and m._lanes == []


def test_continuous_frame_store_follows_the_audio_in_flight(monkeypatch):
    """the frame store used to grow by one row per diffusion iteration of the whole queue and was never released.
    Now a finished utterance's frames leave it at once and blocks no live utterance points into are dropped.  With 4-row blocks the
    dropping happens several times inside this 6-utterance queue: every result must stay identical to generate() alone, the store
    must end with at most one block, and it must never hold more than the blocks the two utterances in flight span."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import BenchHooks, VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    reqs = _requests(6, 5)
    with fake_engine.cpu_cuda_shims(monkeypatch):
        solo = []
        for r in reqs:
            m1 = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=1), model_dtype=torch.float32)
            m1.set_speech_factors(0.2, -0.05)
            m1.set_ddpm_inference_steps(5)
            solo.append(m1.generate(input_ids=r["input_ids"], attention_mask=r["attention_mask"], cfg_scale=1.3, tokenizer=TOK,
                                    generation_config={"do_sample": False}, _forced_tokens=[r["_forced_tokens"]],
                                    _noise_fn=r["_noise_fn"], show_progress_bar=False))
            assert len(m1._audio_blocks) <= 1
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=2), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        m.frame_block = 4
        peak = []
        outs = m.generate_continuous(reqs, tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3,
                                     _bench_hooks=BenchHooks(step_callback=lambda it: peak.append(sum(b is not None for b in m._audio_blocks))))
    frames_total = sum(o.speech_outputs[0].shape[-1] // 3200 for o in outs if o.speech_outputs[0] is not None)
    assert frames_total > 4 * 4                                    # the queue spans several blocks ...
    assert max(peak) <= 3, peak                                    # ... but the store never holds more than the live span
    assert len(m._audio_blocks) <= 1
    for a, b in zip(outs, solo):
        assert torch.equal(a.sequences.cpu(), b.sequences.cpu())
        if b.speech_outputs[0] is not None:
            d = (a.speech_outputs[0] - b.speech_outputs[0]).norm() / b.speech_outputs[0].norm()
            assert float(d) <= 1e-5, float(d)


def test_continuous_length_capped_utterance_leaves_the_others_intact(monkeypatch):
    """an utterance retired by the LOOP-level conditions of generate() (range(max_steps) exhausted / max_length
    reached -- not by EOS) while others stay in flight.  The survivors' next-step embeddings were packed in the old order; they must
    follow the survivors.  3 requests on 2 slots, per-request max_new_tokens 3 / 8 / 6, forced plans that never reach EOS before
    the cap: every request must still come out exactly as generate() produces it alone."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    D, E, S = TOK.speech_diffusion_id, TOK.speech_end_id, TOK.speech_start_id
    reqs = _requests(3, 11)
    caps = [3, 8, 6]
    plans = [[D] * 12, [D, D, E, S] + [D] * 10, [D] * 12]
    for r, c, p in zip(reqs, caps, plans):
        r["max_new_tokens"] = c
        r["_forced_tokens"] = p
    with fake_engine.cpu_cuda_shims(monkeypatch):
        solo = []
        for r in reqs:
            m1 = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=1), model_dtype=torch.float32)
            m1.set_speech_factors(0.2, -0.05)
            m1.set_ddpm_inference_steps(5)
            solo.append(m1.generate(input_ids=r["input_ids"], attention_mask=r["attention_mask"], cfg_scale=1.3, tokenizer=TOK,
                                    generation_config={"do_sample": False}, _forced_tokens=[r["_forced_tokens"]],
                                    _noise_fn=r["_noise_fn"], max_new_tokens=r["max_new_tokens"], show_progress_bar=False))
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=2), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        outs = m.generate_continuous(reqs, tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
    assert m.last_stats["max_in_flight"] == 2
    for a, b, c, r in zip(outs, solo, caps, reqs):
        assert a.sequences.shape[1] - r["input_ids"].shape[1] == c          # really ended by the cap
        assert torch.equal(a.sequences.cpu(), b.sequences.cpu()[:, :a.sequences.shape[1]])
        assert a.speech_outputs[0].shape == b.speech_outputs[0].shape
        d = (a.speech_outputs[0] - b.speech_outputs[0]).norm() / b.speech_outputs[0].norm()
        assert float(d) <= 1e-5, float(d)


def test_batch_above_eight_rows_goes_through_the_queue(monkeypatch):
    """The reference's batch is unbounded (modeling_vibevoice_inference.py:393-394); one engine pass carries 8 utterances.  A
    10-row generate() is decoded through the continuous-admission queue (2 slots here) and comes back in the batched call's own
    form: every row's tokens and waveform are what generate() gives that row alone (left-padded to the batch's width), the
    sequences are eos-padded to the longest row, reach_max_step_sample has one entry per row."""
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    reqs = _requests(10, 5)
    L0 = max(r["input_ids"].shape[1] for r in reqs)
    ids = torch.full((10, L0), TOK.pad_token_id, dtype=torch.long)
    mask = torch.zeros((10, L0), dtype=torch.long)
    for b, r in enumerate(reqs):                                   # left padding, as the processor pads (vibevoice_processor.py:349-353)
        n = r["input_ids"].shape[1]
        ids[b, L0 - n:] = r["input_ids"][0]
        mask[b, L0 - n:] = 1
    forced = [r["_forced_tokens"] for r in reqs]
    bank = {}

    def noise_fn(step, n2):                                        # the same draw for every row at its own step: rows are comparable
        return bank.setdefault(step, synth.Gen(7000 + step).normal((2, 64), 1.0, mat=False))[:n2]

    def new_model(n_slots):
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=n_slots), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        return m
    with fake_engine.cpu_cuda_shims(monkeypatch):
        solo = [new_model(1).generate(input_ids=ids[b:b + 1], attention_mask=mask[b:b + 1], cfg_scale=1.3, tokenizer=TOK,
                                      generation_config={"do_sample": False}, _forced_tokens=[forced[b]], _noise_fn=noise_fn,
                                      show_progress_bar=False) for b in range(10)]
        out = new_model(2).generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=TOK, generation_config={"do_sample": False},
                                    _forced_tokens=forced, _noise_fn=noise_fn, show_progress_bar=False)
    assert out.sequences.shape[0] == 10 and len(out.speech_outputs) == 10 and out.reach_max_step_sample.shape == (10,)
    width = out.sequences.shape[1]
    assert width == max(o.sequences.shape[1] for o in solo)
    for b, o in enumerate(solo):
        n = o.sequences.shape[1]
        assert torch.equal(out.sequences[b, :n].cpu(), o.sequences[0].cpu())
        assert bool((out.sequences[b, n:] == TOK.eos_token_id).all())
        a, r = out.speech_outputs[b], o.speech_outputs[0]
        assert (a is None) == (r is None)
        if a is not None:
            assert a.shape == r.shape and float((a.float() - r.float()).norm() / (r.float().norm() + 1e-30)) <= 1e-5
        assert bool(out.reach_max_step_sample[b]) == bool(o.reach_max_step_sample[0])


def test_greedy_batch2_dense_logits_layout(monkeypatch):
    """Free-running greedy decoding of a desynchronised batch of two through the C ABI's logits layout (a dense [n][n_valid]
    block): every row must argmax over ITS OWN logits (rows >= 1 used to be read with the wrong stride)."""
    from oracle import generate as ogen
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, "generate_forced_b2.npz"))
    ids = torch.from_numpy(z["input_ids"])
    mask = torch.from_numpy(z["attention_mask"])
    noise = {}

    def noise_fn(step, n2):
        return noise.setdefault((step, n2), synth.Gen(77 * step + n2).normal((n2, 64), 1.0, mat=False))
    tok = ogen.TokenIds(301, 302, 303, 304, None, 305)
    oseq, oaud, omax = ogen.oracle_generate(_oracle_small(), tok, ids, mask, cfg_scale=1.3, num_steps=5, max_new_tokens=12, noise_fn=noise_fn)
    cfgd = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    with fake_engine.cpu_cuda_shims(monkeypatch):
        m = VibeVoiceForConditionalGenerationInference(cfgd, fake_engine.FakeEngine(_oracle_small(), n_slots=2), model_dtype=torch.float32)
        m.set_speech_factors(0.2, -0.05)
        m.set_ddpm_inference_steps(5)
        m.concurrent_codecs = False
        out = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=TOK, max_new_tokens=12,
                         generation_config={"do_sample": False}, _noise_fn=noise_fn, show_progress_bar=False)
    assert torch.equal(out.sequences.cpu(), oseq)
    assert not torch.equal(oseq[0, ids.shape[1]:], oseq[1, ids.shape[1]:])       # the two rows really decode differently


# ---------------------------------------------------------------- streamers
def test_streamer_thread_ends_with_the_last_sample():
    from vibevoice_amd.streamer import AudioStreamer
    n0 = threading.active_count()
    streamers = []
    for _ in range(6):
        s = AudioStreamer(batch_size=2, timeout=5.0)
        s.put(torch.ones(2, 1, 8), torch.tensor([0, 1]))
        s.end()
        assert [c.sum().item() for c in s.get_stream(0)] == [8.0]
        streamers.append(s)
    for s in streamers:
        s._thread.join(timeout=5.0)
        assert not s._thread.is_alive() and all(r is None for r in s._ring)
    assert threading.active_count() <= n0
    s = AudioStreamer(batch_size=1)
    s.close()                                           # abandoned before any end(): close() stops the drain thread too
    s._thread.join(timeout=5.0)
    assert not s._thread.is_alive()


def test_async_streamer_matches_the_reference_surface():
    """AsyncAudioStreamer (streamer.py:150-264): put()/end() from a producer thread, `async for` on one sample and on the batch."""
    from vibevoice_amd.streamer import AsyncAudioStreamer

    async def run():
        s = AsyncAudioStreamer(batch_size=2, timeout=5.0)

        def producer():
            for i in range(3):
                s.put(torch.full((2, 1, 4), float(i)), torch.tensor([0, 1]))
                time.sleep(0.005)
            s.end(torch.tensor([1]))
            s.put(torch.full((1, 1, 4), 9.0), torch.tensor([0]))
            s.end()
        t = threading.Thread(target=producer)
        t.start()
        got1 = [c.flatten()[0].item() async for c in s.get_stream(1)]
        got0 = [c.flatten()[0].item() async for c in s.get_stream(0)]
        t.join()
        with pytest.raises(ValueError):
            async for _ in s.get_stream(2):
                pass
        return got0, got1

    got0, got1 = asyncio.run(run())
    assert got1 == [0.0, 1.0, 2.0] and got0 == [0.0, 1.0, 2.0, 9.0]

    async def run_batch():
        s = AsyncAudioStreamer(batch_size=2, timeout=5.0)

        def producer():
            for i in range(4):
                s.put(torch.full((2, 1, 4), float(i)), torch.tensor([0, 1]))
            s.end()
        threading.Thread(target=producer).start()
        seen = {0: [], 1: []}
        async for batch in s:
            for idx, c in batch.items():
                seen[idx].append(c.flatten()[0].item())
        return seen
    seen = asyncio.run(run_batch())
    assert seen == {0: [0.0, 1.0, 2.0, 3.0], 1: [0.0, 1.0, 2.0, 3.0]}           # nothing lost, order kept per sample


def test_pcm16_reference_arithmetic():
    """The rule vv_audio_to_pcm16 implements (checked on the GPU in test_gpu_kernels.py), stated with numpy exactly as
    demo/gradio_demo.py:1058-1073 does it."""
    rng = np.random.default_rng(0)
    for scale in (0.3, 2.5):
        data = (rng.standard_normal(3200) * scale).astype(np.float32)
        ref = data.copy()
        if np.max(np.abs(ref)) > 1.0:
            ref = ref / np.max(np.abs(ref))
        ref = (ref * 32767).astype(np.int16)
        peak = np.float32(np.abs(data).max())
        mine = data / peak if peak > 1.0 else data
        mine = np.trunc(mine * np.float32(32767.0)).astype(np.int16)
        assert np.array_equal(ref, mine)


# ---------------------------------------------------------------- the streaming class (SURVEY 8b "Streaming variant")
def tiny_streaming_checkpoint(tmp_path, eos_bias):
    """The tiny split model of test_oracle_golden._oracle_streaming_small (same seeded weights) as a checkpoint directory
    in the reference's layout: config.json (configuration_vibevoice_streaming.py: decoder_config + tts_backbone_num_hidden_layers)
    and safetensors shards keyed like VibeVoiceStreamingForConditionalGenerationInference.state_dict()."""
    from safetensors.torch import save_file
    n_lm, n_tts = 1, 2
    cfg = synth.LMCfg(hidden=128, layers=n_lm + n_tts, heads=2, kv_heads=1, inter=256, vocab=320)
    H = cfg.hidden
    w = synth.lm_weights(cfg)
    hc = synth.HeadCfg(hidden=H, layers=2)
    cc = synth.CodecCfg()
    g = synth.Gen(900)
    tts_types = g.normal((2, H), 0.5, mat=False)
    eos = {"fc1.weight": g.linear(H, H), "fc1.bias": g.vec(H, 0.1), "fc2.weight": g.linear(1, H, 0.3), "fc2.bias": torch.full((1,), float(eos_bias))}
    sd = {"model.language_model.embed_tokens.weight": w["embed_tokens.weight"], "model.tts_language_model.norm.weight": w["norm.weight"],
          "model.tts_input_types.weight": tts_types, "model.speech_scaling_factor": torch.tensor(0.2), "model.speech_bias_factor": torch.tensor(-0.05)}
    for k, v in w.items():
        if k.startswith("layers."):
            i = int(k.split(".")[1])
            rest = k.split(".", 2)[2]
            sd[(f"model.language_model.layers.{i}." if i < n_lm else f"model.tts_language_model.layers.{i - n_lm}.") + rest] = v
    sd.update({"tts_eos_classifier." + k: v for k, v in eos.items()})
    sd.update({"model.prediction_head." + k: v for k, v in synth.head_weights(hc).items()})
    sd.update({"model.acoustic_tokenizer." + k: v for k, v in synth.decoder_weights(cc, 3).items()})
    sd.update({"model.acoustic_connector." + k: v for k, v in synth.connector_weights(64, H, 4).items()})
    config = tiny_reference_config()
    config.pop("semantic_tokenizer_config")
    config["tts_backbone_num_hidden_layers"] = n_tts
    config["decoder_config"].update(hidden_size=H, intermediate_size=cfg.inter, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                                    num_hidden_layers=cfg.layers, vocab_size=cfg.vocab, max_position_embeddings=512)
    config["diffusion_head_config"]["hidden_size"] = H
    d = tmp_path / "tiny-vibevoice-streaming"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(config))
    sd = {k: v.contiguous() for k, v in sd.items()}
    keys = sorted(sd)
    save_file({k: sd[k] for k in keys[::2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[1::2]}, str(d / "model-00002-of-00002.safetensors"))
    return str(d)


def test_reference_streaming_demo_call_sequence(monkeypatch, tmp_path, capsys):
    """demo/streaming_inference_from_file.py:226-355 replayed line by line against the product's streaming class loaded from a
    checkpoint directory (Engine replaced by the oracle-backed fake): from_pretrained(path, torch_dtype, device_map,
    attn_implementation) -> eval() -> set_ddpm_inference_steps(num_steps=5) -> `hasattr(model.model, 'language_model')` +
    the `_attn_implementation` read (:285-286) -> the voice preset as a dict of BaseModelOutputWithPast (:291), deep-copied
    (:318) -> generate(**processor_inputs, max_new_tokens=None, cfg_scale, tokenizer, generation_config, verbose,
    all_prefilled_outputs) -> the fields the demo reads (:325-341).  Preset, text and noise are the ones of the golden the
    REFERENCE's streaming generate() recorded (streaming_eos.npz): same token count, waveform rel-L2 <= 1e-4."""
    import copy
    from transformers.modeling_outputs import BaseModelOutputWithPast
    from vibevoice_amd import modeling_streaming
    z = np.load(os.path.join(GOLD, "streaming_eos.npz"))
    path = tiny_streaming_checkpoint(tmp_path, float(z["eos_bias"]))
    draws = [torch.from_numpy(z[f"draw_{i}"]).reshape(2, 64) for i in range(int(z["n_draws"]))]

    def branch(tag):
        n = int(z[f"{tag}_layers"])
        kv = [(torch.from_numpy(z[f"{tag}_k{li}"])[None], torch.from_numpy(z[f"{tag}_v{li}"])[None]) for li in range(n)]
        hid = torch.zeros(1, kv[0][0].shape[2], 128)
        hid[0, -1] = torch.from_numpy(z[f"{tag}_last"])
        return BaseModelOutputWithPast(last_hidden_state=hid, past_key_values=kv)
    all_prefilled_outputs = {"lm": branch("lm"), "tts_lm": branch("tts"), "neg_lm": None, "neg_tts_lm": branch("neg_tts")}
    with fake_engine.cpu_cuda_shims(monkeypatch):
        monkeypatch.setattr(modeling_streaming, "Engine", fake_engine.LoadableFakeStreamingEngine)
        cls = modeling_streaming.VibeVoiceStreamingForConditionalGenerationInference
        model = cls.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda", attn_implementation="flash_attention_2")
        model.eval()
        model.set_ddpm_inference_steps(num_steps=5)
        assert hasattr(model.model, "language_model")
        print(f"Language model attention: {model.model.language_model.config._attn_implementation}")      # :285-286
        assert "Language model attention: vvhip" in capsys.readouterr().out
        assert model.requested_attn_implementation == "flash_attention_2"
        # reference properties (modeling_vibevoice_streaming_inference.py:119-141) and the split decoder's two halves
        assert model.prediction_head is model.model.prediction_head and model.acoustic_connector is model.model.acoustic_connector
        assert model.acoustic_tokenizer is model.model.acoustic_tokenizer
        assert model.model.language_model.config.num_hidden_layers == 1 and model.model.tts_language_model.config.num_hidden_layers == 2
        assert abs(float(model.speech_scaling_factor) - 0.2) < 1e-7 and abs(float(model.model.speech_bias_factor) + 0.05) < 1e-7
        assert next(model.parameters()).device == model.device and model.to("cuda") is model
        assert model.config.tts_backbone_num_hidden_layers == 2 and list(model.noise_scheduler.timesteps[:2]) == [999, 799]
        assert sorted(model.tts_eos_classifier.expected_keys()) == ["fc1.bias", "fc1.weight", "fc2.bias", "fc2.weight"]
        assert "layers.1.mlp.up_proj.weight" in model.model.tts_language_model.expected_keys()
        assert "layers.1.mlp.up_proj.weight" not in model.model.language_model.expected_keys()

        prompt = torch.from_numpy(z["prompt"])[None]
        text = torch.from_numpy(z["text"])[None]
        lm_len = all_prefilled_outputs["lm"]["last_hidden_state"].size(1)
        inputs = {"input_ids": torch.zeros(1, lm_len, dtype=torch.long), "attention_mask": torch.ones(1, lm_len, dtype=torch.long),
                  "tts_lm_input_ids": prompt, "tts_lm_attention_mask": torch.ones_like(prompt), "tts_text_ids": text,
                  "speech_input_mask": torch.zeros(1, prompt.shape[1], dtype=torch.bool), "speech_tensors": None, "speech_masks": None}
        for k, v in inputs.items():
            if torch.is_tensor(v):
                inputs[k] = v.to("cpu")                                                   # the demo's .to(target_device)
        outputs = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.5, tokenizer=TOK, generation_config={"do_sample": False},
                                 verbose=True, all_prefilled_outputs=copy.deepcopy(all_prefilled_outputs),
                                 _noise_fn=lambda frame, n2: draws[frame])               # the recorded draws: the only test hook
    assert outputs.speech_outputs and outputs.speech_outputs[0] is not None
    audio_samples = outputs.speech_outputs[0].shape[-1]
    assert audio_samples % 3200 == 0 and audio_samples > 0
    input_tokens = inputs["tts_text_ids"].shape[1]
    output_tokens = outputs.sequences.shape[1]
    generated_tokens = output_tokens - input_tokens - all_prefilled_outputs["tts_lm"]["last_hidden_state"].size(1)     # :337-339
    # (the demo's arithmetic; with an EOS inside the first speech window only 5 of the 12 text ids were consumed and the window's
    # 6 speech tokens are all in `sequences` although one frame of audio was kept -- the reference's own bookkeeping, :646-700)
    assert generated_tokens == int(z["n_tokens"]) - 12 - 23
    assert output_tokens == int(z["n_tokens"]) and bool(outputs.reach_max_step_sample[0]) == bool(z["reach_max"][0])
    ref = torch.from_numpy(z["audio"])
    got = outputs.speech_outputs[0].reshape(-1)
    assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4


# ---------------------------------------------------------------- the gradio demo's scheduler swap (demo/gradio_demo.py:51-150,549-602)
@pytest.mark.parametrize("name", ["generate_sde_b1", "generate_sde_b2"])
def test_gradio_demo_scheduler_swap_and_generate(product, capsys, name):
    """demo/gradio_demo.py:139-150 then :566-602 replayed against the product class: eval() ->
    `model.model.noise_scheduler = model.model.noise_scheduler.from_config(config, algorithm_type='sde-dpmsolver++',
    beta_schedule='squaredcos_cap_v2')` -> set_ddpm_inference_steps -> the attention print -> generate(**inputs,
    max_new_tokens=None, cfg_scale, tokenizer, generation_config, generator=..., audio_streamer, stop_check_fn, verbose=False,
    refresh_negative=True, is_prefill=...).  Nothing is injected but the forced token plan (random weights never emit
    <speech_diffusion>): seeded like the run that recorded the golden with the REFERENCE's own generate() under the same
    swap, the product's draws -- prefill, per-frame initial noise, one randn(2n, 64) per solver step for the stochastic
    solver -- come off the global generator in the reference's order and the result lands on the golden."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(
        path, torch_dtype=torch.float32, device_map="cuda", attn_implementation="flash_attention_2")
    model.eval()
    # Use SDE solver by default (:141-146)
    model.model.noise_scheduler = model.model.noise_scheduler.from_config(
        model.model.noise_scheduler.config,
        algorithm_type='sde-dpmsolver++',
        beta_schedule='squaredcos_cap_v2'
    )
    model.set_ddpm_inference_steps(num_steps=5)
    if hasattr(model.model, 'language_model'):
        print(f"Language model attention: {model.model.language_model.config._attn_implementation}")
    assert "Language model attention: vvhip" in capsys.readouterr().out
    sched = model.model.noise_scheduler
    assert sched.config.algorithm_type == "sde-dpmsolver++" and sched.config.beta_schedule == "squaredcos_cap_v2"
    assert sched.config.prediction_type == "v_prediction" and sched.config.solver_order == 2 and sched.num_inference_steps == 5
    # configurations the HIP sampler does not implement are refused at the assignment, not run as something else
    with pytest.raises(NotImplementedError):
        model.model.noise_scheduler = sched.from_config(sched.config, algorithm_type="dpmsolver")
    with pytest.raises(NotImplementedError):
        model.model.noise_scheduler = sched.from_config(sched.config, use_karras_sigmas=True)
    assert model.model.noise_scheduler.config.algorithm_type == "sde-dpmsolver++"      # a refused swap changes nothing

    z = np.load(os.path.join(GOLD, name + ".npz"))
    B = z["input_ids"].shape[0]
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"])}
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    stop_generation = False
    generator = torch.Generator()                    # :579-586: created, seeded, passed -- and never consumed by generate()
    generator.manual_seed(123)
    model.set_ddpm_inference_steps(num_steps=int(5))                                     # :568
    torch.manual_seed(int(z["seed"]))
    outputs = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                             generator=generator, audio_streamer=None, stop_check_fn=lambda: stop_generation, verbose=False,
                             refresh_negative=True, is_prefill=True, _forced_tokens=forced)
    assert torch.equal(outputs.sequences.cpu(), torch.from_numpy(z["sequences"]))
    assert torch.equal(outputs.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        got = outputs.speech_outputs[b].reshape(-1)
        assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4
    # back to the deterministic solver: the same seeded call now reproduces the plain golden's behaviour class (no step draws)
    model.model.noise_scheduler = sched.from_config(sched.config, algorithm_type="dpmsolver++")
    assert model.model.noise_scheduler.config.algorithm_type == "dpmsolver++"
    torch.manual_seed(int(z["seed"]))
    out2 = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                          verbose=False, is_prefill=True, _forced_tokens=forced)
    assert torch.equal(out2.sequences.cpu(), outputs.sequences.cpu())
    assert not torch.allclose(out2.speech_outputs[0].float(), outputs.speech_outputs[0].float())      # a different solver


# ---------------------------------------------------------------- refresh_negative=False (modeling_vibevoice_inference.py:503-516)
@pytest.mark.parametrize("name", ["generate_norefresh_b1", "generate_norefresh_b2"])
def test_generate_without_negative_refresh_lands_on_the_reference_golden(product, name):
    """generate(..., refresh_negative=False) of the product loop, seeded like the run that recorded the golden with the REFERENCE's own
    generate() in that mode: the negative pass runs at every step on the positive pass's input (the lone <speech_start> at step 0),
    is never reset, and -- batch of two -- a row that does not diffuse while the other does loses the step's entry again
    (:590-624).  Only the forced token plan is injected; sequences identical, waveform rel-L2 <= 1e-4.  The request queue refuses
    the mode (it is a rule over the rows of one lock-step batch)."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=5)
    z = np.load(os.path.join(GOLD, name + ".npz"))
    B = z["input_ids"].shape[0]
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"])}
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    for speculate in (False, True):
        model.speculate_sampling = speculate
        torch.manual_seed(int(z["seed"]))
        outputs = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                                 verbose=False, refresh_negative=False, is_prefill=True, _forced_tokens=forced)
        assert torch.equal(outputs.sequences.cpu(), torch.from_numpy(z["sequences"]))
        assert torch.equal(outputs.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
        for b in range(B):
            ref = torch.from_numpy(z[f"audio_{b}"])
            got = outputs.speech_outputs[b].reshape(-1)
            assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4
    # the refreshed mode on the same seeded call is something else
    torch.manual_seed(int(z["seed"]))
    out2 = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                          verbose=False, is_prefill=True, _forced_tokens=forced)
    assert not torch.allclose(out2.speech_outputs[0].float(), outputs.speech_outputs[0].float())
    with pytest.raises(NotImplementedError):
        model.generate_continuous([{k: v[:1] for k, v in inputs.items()}], tokenizer=TOK, refresh_negative=False)


# ---------------------------------------------------------------- max_length_times (modeling_vibevoice_inference.py:370, :421-422, :523-539)
def test_generate_with_a_length_factor_lands_on_the_reference_golden(product):
    """generate(..., max_length_times=0.4) on a left-padded batch of two, seeded like the run that recorded the golden with the REFERENCE's
    own generate(): the loop runs int(0.4 * padded width) steps, each row's own cap follows its unpadded length -- the shorter row is
    stopped by reach_max_step_sample (its last token is still recorded, its frame is not) while the longer one runs to the end of the
    loop unflagged.  Only the forced token plan is injected."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=5)
    z = np.load(os.path.join(GOLD, "generate_times_b2.npz"))
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"])}
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(2)]
    torch.manual_seed(int(z["seed"]))
    out = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                         verbose=False, is_prefill=True, max_length_times=0.4, _forced_tokens=forced)
    assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
    assert out.reach_max_step_sample.tolist() == [False, True]
    assert torch.equal(out.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
    for b in range(2):
        ref = torch.from_numpy(z[f"audio_{b}"])
        got = out.speech_outputs[b].reshape(-1)
        assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4


# ---------------------------------------------------------------- a row whose first frame comes later than the other row's
@pytest.mark.parametrize("name", ["generate_late_start_b2", "generate_late_start_b2r", "generate_multivoice_b2", "generate_ragged_voice_full_b2"])
def test_late_starting_row_costs_the_streaming_row_its_conv_history_as_in_the_reference(product, name):
    """The reference's VibeVoiceTokenizerStreamingCache.get (modular_vibevoice_tokenizer.py:198-207) returns None for a whole decode /
    encode call as soon as one requested row has no entry yet: in a lock-step batch the row that was already streaming loses its conv
    history for the frame in which another row diffuses for the first time.  generate() reproduces it (golden recorded from the
    reference's own generate(); waveform rel-L2 <= 1e-4 on BOTH rows); the request queue does not -- there each request ends as
    generate() on it alone would, which for the streaming row is a different waveform.
    (generate_multivoice_b2 rides along: several voice samples in one prompt -- three samples for two rows -- seeded, nothing injected
    but the plan.)"""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=5)
    z = np.load(os.path.join(GOLD, name + ".npz"))
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"])}
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(2)]
    errs = {}
    for speculate in (False, True):
        model.speculate_sampling = speculate
        torch.manual_seed(int(z["seed"]))
        out = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                             verbose=False, is_prefill=True, _forced_tokens=forced)
        assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
        for b in range(2):
            ref = torch.from_numpy(z[f"audio_{b}"])
            got = out.speech_outputs[b].reshape(-1)
            assert got.shape == ref.shape
            errs[(speculate, b)] = float((got - ref).norm() / ref.norm())
    assert max(errs.values()) <= 1e-4, errs


# ---------------------------------------------------------------- the correction that keeps THIS step's entry (vv_kv_move)
def test_single_entry_correction_follows_the_reference(product):
    """A one-frame speech segment in one row while the other row diffuses: the reference's correction of the non-diffusing row moves the
    mask and not the K/V (modeling_vibevoice_inference.py:603 vs :613), so it keeps the negative entry appended at THAT step and masks
    the older one.  The product recognises the case from the reference's own bookkeeping (masks and counters, no tensors) and moves the
    step's entry onto the older one (vv_kv_move: position 1 -> 0, rotation kept): both rows land on the golden recorded from the
    reference's generate()."""
    modeling, path = product
    model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
    model.eval()
    model.set_ddpm_inference_steps(num_steps=5)
    z = np.load(os.path.join(GOLD, "generate_single_entry_b2.npz"))
    inputs = {"input_ids": torch.from_numpy(z["input_ids"]), "attention_mask": torch.from_numpy(z["attention_mask"]),
              "speech_tensors": torch.from_numpy(z["speech_tensors"]), "speech_masks": torch.from_numpy(z["speech_masks"]),
              "speech_input_mask": torch.from_numpy(z["speech_input_mask"])}
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(2)]
    for speculate in (False, True):
        model.speculate_sampling = speculate
        torch.manual_seed(int(z["seed"]))
        out = model.generate(**inputs, max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK, generation_config={'do_sample': False},
                             verbose=False, is_prefill=True, _forced_tokens=forced)
        assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
        assert torch.equal(out.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
        for b in range(2):
            ref, got = torch.from_numpy(z[f"audio_{b}"]), out.speech_outputs[b].reshape(-1)
            assert got.shape == ref.shape and float((got - ref).norm() / ref.norm()) <= 1e-4
