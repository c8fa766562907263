"""End-to-end parity of generate() (HIP engine through the C ABI) against the
oracle loop on identical inputs, weights, forced/greedy tokens and noise.

Tolerance: xsplit=3 keeps every GEMM at fp32-class accuracy; what accumulates
through the autoregressive feedback is summation order + the bf16 KV cache
(mirrored by the oracle's kv_round_bf16).  Stated bounds: token decisions
identical; per-frame latent rel-L2 <= 5e-3; waveform rel-L2 <= 1e-2 over
<= 12 frames.
"""
import types

import pytest
import torch

import synth
from gpu_util import build_small, rel_err
from oracle import generate as ogen

pytestmark = pytest.mark.gpu

TOK = ogen.TokenIds(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                    bos_token_id=None, pad_token_id=305)


def make_inputs(s, B, with_speech, seed):
    g = synth.Gen(seed)
    V = s.lmcfg.vocab
    lens = [21, 17][:B]
    L0 = max(lens)
    ids = torch.full((B, L0), TOK.pad_token_id, dtype=torch.long)
    mask = torch.zeros((B, L0), dtype=torch.long)
    sim = torch.zeros((B, L0), dtype=torch.bool)
    speech_tensors = speech_masks = None
    n_fr = [2, 3]
    for b in range(B):
        n = lens[b]
        row = torch.from_numpy(g.rng.integers(0, 300, (n,)))
        row[-1] = TOK.speech_start_id
        ids[b, L0 - n:] = row
        mask[b, L0 - n:] = 1
        if with_speech:
            # voice prompt placeholders: n_fr[b] <speech_diffusion> positions
            st = L0 - n + 3
            ids[b, st:st + n_fr[b]] = TOK.speech_diffusion_id
            sim[b, st:st + n_fr[b]] = True
    if with_speech:
        S = 3 * 3200
        speech_tensors = g.uniform((B, S), -0.5, 0.5)
        speech_masks = torch.zeros((B, 3), dtype=torch.bool)
        for b in range(B):
            speech_masks[b, :n_fr[b]] = True
    return ids, mask, sim, speech_tensors, speech_masks


def run_both(s, B, forced, with_speech, cfg=1.3, steps=5, seed=11, max_new_tokens=None, **mode):
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    ids, mask, sim, st, sm = make_inputs(s, B, with_speech, seed)
    g = synth.Gen(seed + 1)
    noise_bank = {}

    def noise_fn(step, n2):
        if (step, n2) not in noise_bank:
            noise_bank[(step, n2)] = synth.Gen(seed * 1000 + step).normal((n2, 64), 1.0, mat=False)
        return noise_bank[(step, n2)]
    pre = None
    if with_speech:
        pre = (g.normal((B,), 1.0, mat=False), g.normal((B, 3, 64), 1.0, mat=False))
    om = s.oracle_model(kv_round_bf16=True)
    otr = ogen.Trace()
    oseq, oaud, omax = ogen.oracle_generate(om, TOK, ids, mask, st, sm, sim if with_speech else None, cfg_scale=cfg,
                                            num_steps=steps, max_new_tokens=max_new_tokens, noise_fn=noise_fn,
                                            prefill_noise=pre, forced_tokens=forced, trace=otr, **mode)
    cfgd = {"decoder_config": {"max_position_embeddings": s.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": steps},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, s.eng, model_dtype=torch.float32)
    m.set_speech_factors(s.scaling, s.bias)
    m.set_ddpm_inference_steps(steps)
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    htr = ogen.Trace()
    out = m.generate(input_ids=ids, attention_mask=mask, speech_tensors=st, speech_masks=sm,
                     speech_input_mask=sim if with_speech else None, cfg_scale=cfg, tokenizer=tok,
                     max_new_tokens=max_new_tokens, generation_config={"do_sample": False},
                     _forced_tokens=forced, _noise_fn=noise_fn, _prefill_noise=pre, _trace=htr,
                     show_progress_bar=False, **mode)
    return (oseq, oaud, omax, otr), (out, htr)


@pytest.fixture(scope="module")
def sm():
    s = build_small(synth.LMCfg(), xsplit=3, n_slots=2, max_ctx=512)
    yield s
    s.eng.close()


D, E, S, X = TOK.speech_diffusion_id, TOK.speech_end_id, TOK.speech_start_id, TOK.eos_token_id


def check(o, h, lat_tol=5e-3, wav_tol=1e-2):
    (oseq, oaud, omax, otr), (out, htr) = o, h
    assert torch.equal(out.sequences.cpu(), oseq)
    assert torch.equal(out.reach_max_step_sample.cpu(), omax)
    assert len(otr.latents) == len(htr.latents)
    for a, b in zip(htr.latents, otr.latents):
        assert rel_err(a, b) <= lat_tol, rel_err(a, b)
    for a, b in zip(htr.neg_hidden, otr.neg_hidden):
        assert rel_err(a, b) <= lat_tol, rel_err(a, b)
    for a, b in zip(out.speech_outputs, oaud):
        if b is None:
            assert a is None
        else:
            assert a.shape[-1] == b.shape[-1]
            assert rel_err(a[0], b[0]) <= wav_tol, rel_err(a[0], b[0])


def test_generate_forced_single(sm):
    forced = [[D, D, D, D, E, S, D, D, D, X]]
    o, h = run_both(sm, 1, forced, with_speech=True)
    check(o, h)
    assert h[0].speech_outputs[0].shape[-1] == 7 * 3200


def test_generate_forced_batch2_desync(sm):
    # rows desynchronise: row 1 ends its first segment earlier and finishes earlier
    forced = [[D, D, D, E, S, D, D, X], [D, D, E, S, D, X]]
    o, h = run_both(sm, 2, forced, with_speech=True, seed=23)
    check(o, h)


@pytest.mark.parametrize("B,forced,seed", [(1, [[D, D, D, D, E, S, D, D, D, X]], 83), (2, [[D, D, D, E, S, D, D, X], [D, D, E, S, D, X]], 89),
                                           (2, [[S, D, D, E, S, D, X], [D, E, E, S, D, D, D, X]], 97)])
def test_generate_without_negative_refresh(sm, B, forced, seed):
    """refresh_negative=False (modeling_vibevoice_inference.py:503-516): the negative branch consumes every step's input, is never
    reset on <speech_start>, and in a batch the rows that do not diffuse while another does lose the step's entry again (:590-624).
    Engine vs the oracle loop, which tests/test_oracle_golden.py pins to the reference's own generate() in this mode
    (tests/golden/generate_norefresh_b{1,2}.npz); speculation on (the wrong guesses at <speech_end> are discarded)."""
    o, h = run_both(sm, B, forced, with_speech=True, seed=seed, refresh_negative=False)
    check(o, h)
    # the mode is not a no-op on these plans: the negative conditions differ from the refreshed run's after the first <speech_start>
    o2, _ = run_both(sm, B, forced, with_speech=True, seed=seed)
    assert any(rel_err(a, b) > 1e-2 for a, b in zip(o[3].neg_hidden, o2[3].neg_hidden))


@pytest.mark.parametrize("name", ["generate_norefresh_b1", "generate_norefresh_b2", "generate_late_start_b2", "generate_late_start_b2r", "generate_times_b2", "generate_multivoice_b2", "generate_ragged_voice_full_b2",
                                  "generate_single_entry_b2"])
def test_generate_against_the_reference_goldens_of_the_rare_modes(sm, name):
    """the engine directly against what the REFERENCE's generate(refresh_negative=False) produced on the same tiny seeded model, inputs,
    forced plan and recorded noise draws (tests/golden/make_golden.py::gen_generate): sequences identical, waveform rel-L2 <= 1e-2
    (xsplit = 3 against fp32).  Also, in the default mode: the two late-start files (a row whose first frame comes later than the other
    row's costs the streaming row its tokenizer conv history for that frame, modular_vibevoice_tokenizer.py:198-207) and the
    max_length_times=0.4 run (per-row length caps on a left-padded batch), a multi-speaker prompt batch (three voice samples for two rows),
    voice samples that do not fill their last frame (the partial frame is part of the prompt: vv_acoustic_encode_ragged) and the one-frame
    segment whose negative-cache correction keeps the step's own entry (modeling_vibevoice_inference.py:603 vs :613: vv_kv_move).
    No RuntimeWarning may be raised: the product has no known deviation left to warn about."""
    import warnings
    import os
    import numpy as np
    from test_oracle_golden import G as GOLD
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    z = np.load(os.path.join(GOLD, name + ".npz"))
    ids = torch.from_numpy(z["input_ids"])
    B = ids.shape[0]
    draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
    N = z["speech_tensors"].shape[0]                    # voice samples of the whole batch (generate_multivoice_b2: 3 for 2 rows)
    pre = (draws[0].reshape(N), draws[1].reshape(N, 3, 64))
    it = iter(draws[2:])
    forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.set_ddpm_inference_steps(5)
    m.speculate_sampling = False               # the recorded draws are consumed strictly in order
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        out = m.generate(input_ids=ids, attention_mask=torch.from_numpy(z["attention_mask"]), speech_tensors=torch.from_numpy(z["speech_tensors"]),
                         speech_masks=torch.from_numpy(z["speech_masks"]), speech_input_mask=torch.from_numpy(z["speech_input_mask"]),
                         cfg_scale=1.3, tokenizer=tok, generation_config={"do_sample": False}, _forced_tokens=forced, _prefill_noise=pre,
                         _noise_fn=lambda step, n2: next(it).reshape(n2, 64), show_progress_bar=False, refresh_negative="norefresh" not in name,
                         **({"max_length_times": 0.4} if name == "generate_times_b2" else {}))
    assert torch.equal(out.sequences.cpu(), torch.from_numpy(z["sequences"]))
    assert torch.equal(out.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
    assert next(it, None) is None
    for b in range(B):
        ref = torch.from_numpy(z[f"audio_{b}"])
        got = out.speech_outputs[b].reshape(-1).float().cpu()
        assert got.shape == ref.shape
        assert rel_err(got, ref) <= 1e-2, rel_err(got, ref)


def test_generate_greedy_free_running(sm):
    # free-running argmax over the 4 valid ids; random weights pick whatever they pick --
    # both sides must pick the same ids
    o, h = run_both(sm, 1, None, with_speech=False, seed=31, max_new_tokens=10)
    check(o, h)


def test_generate_max_length_cap(sm):
    forced = [[D] * 50]
    o, h = run_both(sm, 1, forced, with_speech=False, seed=41, max_new_tokens=6)
    check(o, h)
    assert h[0].sequences.shape[1] == 21 + 6


def test_generate_with_graphs():
    s = build_small(synth.LMCfg(), xsplit=3, n_slots=1, max_ctx=512, use_graph=True)
    try:
        forced = [[D, D, D, D, E, S, D, D, D, X]]
        o, h = run_both(s, 1, forced, with_speech=True)
        check(o, h)
        assert_kernel_nodes_only(s.eng)
    finally:
        s.eng.close()


def assert_kernel_nodes_only(eng):
    """Round 6: a MEMSET node of a replayed hipGraph was seen to fill its range with stale words (host stack addresses among them) instead of the
    captured zero once other graph executables had come and gone in the process -- the sampler's previous-x0 buffer then held a NaN bit
    pattern on some GPUs of the pool and not on others (profiles/r06_memset_node_ab.txt).  Copies and fills inside captured sequences are
    kernels of the library now; vv_stat(ctx, 5) counts the nodes of every captured graph that are not kernel launches."""
    assert eng.stat(1) > 0, "the run captured no graph at all"
    assert eng.stat(5) == 0, f"{eng.stat(5)} memset / memcpy nodes inside the engine's captured graphs"


def test_speculative_sampler_is_invisible(sm):
    """The sampler is enqueued before the token is known (modeling.py).  With the CPU global RNG as the noise source
    (the reference's, :701) a wrong guess must restore the RNG state: speculation on/off give bit-identical output."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    ids, mask, sim, st, spm = make_inputs(sm, 1, False, 77)
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    forced = [[D, D, D, E, S, D, D, E, S, D, X]]       # every <speech_end> is a wrong guess
    outs = []
    for spec in (True, False):
        m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
        m.set_speech_factors(sm.scaling, sm.bias)
        m.set_ddpm_inference_steps(5)
        m.speculate_sampling = spec
        torch.manual_seed(1234)
        out = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, generation_config={"do_sample": False},
                         _forced_tokens=forced, show_progress_bar=False)
        outs.append((out.sequences.cpu(), out.speech_outputs[0].cpu(), torch.rand(1)))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][1].shape[-1] == 6 * 3200
    assert torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][2], outs[1][2])          # the RNG stream ends in the same state


class FakeStreamer:
    """Records the AudioStreamer calls generate() must make (streamer.py:42-76)."""

    def __init__(self, batch):
        self.finished_flags = [False] * batch
        self.puts, self.ends = [], []

    def put(self, chunk, idx):
        self.puts.append((tuple(chunk.shape), idx.tolist()))

    def end(self, idx=None):
        self.ends.append(None if idx is None else idx.tolist())


def test_streamer_contract_and_stop_check(sm):
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    s = sm
    ids, mask, sim, st, smk = make_inputs(s, 2, False, 51)
    cfgd = {"decoder_config": {"max_position_embeddings": s.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, s.eng, model_dtype=torch.float32)
    m.set_speech_factors(s.scaling, s.bias)
    m.set_ddpm_inference_steps(5)
    tok = types.SimpleNamespace(speech_start_id=S, speech_end_id=E, speech_diffusion_id=D, eos_token_id=X,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    fs = FakeStreamer(2)
    forced = [[D, D, X], [D, X]]
    out = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, audio_streamer=fs,
                     _forced_tokens=forced, show_progress_bar=False)
    # one put per step with the rows that diffused, [n,1,3200] chunks; end(idx) at EOS; a final end()
    assert fs.puts == [((2, 1, 3200), [0, 1]), ((1, 1, 3200), [0])]
    assert fs.ends == [[1], [0], None]
    assert out.speech_outputs[0].shape[-1] == 2 * 3200 and out.speech_outputs[1].shape[-1] == 3200
    # external stop: generation ends before the first step and the streamer is closed
    fs2 = FakeStreamer(2)
    out2 = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, audio_streamer=fs2,
                      stop_check_fn=lambda: True, _forced_tokens=forced, show_progress_bar=False)
    assert fs2.puts == [] and fs2.ends[0] is None
    assert out2.sequences.shape[1] == ids.shape[1] and out2.speech_outputs == [None, None]
    # a consumer that closed its stream stops generation at the next step
    fs3 = FakeStreamer(2)
    fs3.finished_flags[0] = True
    out3 = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, audio_streamer=fs3,
                      _forced_tokens=forced, show_progress_bar=False)
    assert fs3.puts == []
    with pytest.raises(NotImplementedError):       # a rule over the rows of ONE lock-step batch: the request queue refuses it
        m.generate_continuous([dict(input_ids=ids[:1], attention_mask=mask[:1])], tokenizer=tok, refresh_negative=False)
    # a batch the engine cannot take in lock-step (6 rows, 2 slots) goes through the continuous-admission queue, as a batch above 8 does
    big, bigm = torch.cat([ids, ids, ids], 0), torch.cat([mask, mask, mask], 0)
    import vibevoice_amd.modeling as vmod
    vmod._WARNED_QUEUED_RNG = False                # the notice is given once per process
    with pytest.warns(UserWarning, match="continuous-admission queue"):
        out6 = m.generate(input_ids=big, attention_mask=bigm, cfg_scale=1.3, tokenizer=tok, _forced_tokens=forced * 3, show_progress_bar=False)
    assert out6.sequences.shape[0] == 6 and len(out6.speech_outputs) == 6
    for b in range(6):
        assert out6.speech_outputs[b].shape == out.speech_outputs[b % 2].shape
        assert torch.equal(out6.sequences[b, :ids.shape[1] + len(forced[b % 2])].cpu(), out.sequences[b % 2, :ids.shape[1] + len(forced[b % 2])].cpu())


def test_pinned_ring_streamer_delivers_the_generated_audio(sm):
    """vibevoice_amd.AudioStreamer (async D2H into a pinned ring, background hand-off): the chunks a consumer reads are,
    in order, exactly the waveform generate() returns."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    from vibevoice_amd.streamer import AudioStreamer
    ids, mask, sim, st, spm = make_inputs(sm, 2, False, 55)
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.set_ddpm_inference_steps(5)
    streamer = AudioStreamer(batch_size=2, timeout=20.0)
    # both samples end on the same step: like the reference (:443-447) generate() stops as soon as ANY stream has been ended
    forced = [[D, D, D, E, S, D, X], [D, D, D, D, D, D, X]]
    torch.manual_seed(7)
    out = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, generation_config={"do_sample": False},
                     _forced_tokens=forced, show_progress_bar=False, audio_streamer=streamer)
    for b in range(2):
        chunks = [c.flatten() for c in streamer.get_stream(b)]
        assert len(chunks) == (4, 6)[b]
        assert torch.equal(torch.cat(chunks), out.speech_outputs[b].cpu().flatten())


def test_streamer_ring_buffers_are_pooled_across_requests():
    """One AudioStreamer per request: a finished streamer hands its pinned ring buffers to the module's pool and the next one takes
    them (no pinned allocation in front of the next request's first chunk).  Three streamers in a row, different data each: the
    consumer reads exactly what was put, the second and third run on buffers the first one allocated."""
    from vibevoice_amd import streamer as S
    dev = torch.device("cuda")
    with S._POOL_LOCK:
        S._POOL.clear()
    seen = []
    for trial in range(3):
        st = S.AudioStreamer(batch_size=1, timeout=20.0)
        data = [torch.full((1, 1, 3200), float(trial * 10 + i), device=dev) + torch.arange(3200, device=dev) * 1e-3 for i in range(5)]
        for d in data:
            st.put(d, torch.tensor([0]))
        torch.cuda.synchronize()
        seen.append({b.data_ptr() for b in st._ring if b is not None})
        st.end()
        got = list(st.get_stream(0))
        assert len(got) == 5
        for g, d in zip(got, data):
            assert torch.equal(g.flatten(), d.cpu().flatten())
        st._thread.join(timeout=10)
        assert st._ring == [None] * len(st._ring)
        with S._POOL_LOCK:
            pooled = {b.data_ptr() for v in S._POOL.values() for b in v}
        assert seen[-1] <= pooled                      # handed back
    assert seen[1] <= seen[0] or seen[1] & seen[0]     # the second request ran on the first one's buffers
    assert seen[2] & (seen[0] | seen[1])
    # a put() after the stream has closed is dropped, and nothing is written into a buffer that went back to the pool
    st.put(torch.zeros(1, 1, 3200, device=dev), torch.tensor([0]))
    assert st._ring == [None] * len(st._ring)


def test_generate_greedy_batch2_free_running(sm):
    """Free-running argmax on a desynchronised batch of TWO: vv_lm_logits writes a dense [n][n_valid] block
    and every row must pick from its own logits.  Row 1 of a B=2 run used to read stale data; forced-token tests cannot
    see that."""
    o, h = run_both(sm, 2, None, with_speech=True, seed=61, max_new_tokens=10)
    check(o, h)
    oseq = o[0]
    assert not torch.equal(oseq[0, -10:], oseq[1, -10:])      # the rows really decode different tokens


def _mk_requests(s, n, seed):
    g = synth.Gen(seed)
    plans = [[D, D, D, X], [D, E, S, D, D, D, D, X], [D, D, X], [D, D, D, D, D, E, X], [D, X], [D, D, D, E, S, D, X], [D, D, D, D, X]]
    reqs = []
    for i in range(n):
        L = 9 + 3 * (i % 4)
        ids = torch.from_numpy(g.rng.integers(0, 300, (1, L)))
        ids[0, -1] = S
        bank = {st: synth.Gen(1000 * (seed + i) + st).normal((2, 64), 1.0, mat=False) for st in range(16)}
        reqs.append({"input_ids": ids, "attention_mask": torch.ones_like(ids), "_forced_tokens": plans[i % len(plans)],
                     "_noise_fn": (lambda nz: (lambda step, n2: nz[step]))(bank)})
    return reqs


def test_continuous_admission_on_the_engine(sm):
    """generate_continuous(): 5 queued utterances through the engine's 2 slots -- a freed slot (KV caches, codec states) is
    re-used by the next utterance while the other keeps decoding -- each against the oracle loop run on it alone."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    reqs = _mk_requests(sm, 5, 7)
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.set_ddpm_inference_steps(5)
    tok = types.SimpleNamespace(speech_start_id=S, speech_end_id=E, speech_diffusion_id=D, eos_token_id=X,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    outs = m.generate_continuous(reqs, tokenizer=tok, generation_config={"do_sample": False}, cfg_scale=1.3)
    assert m.last_stats["max_in_flight"] == 2 and len(m.last_stats["admissions"]) == 5
    om = sm.oracle_model(kv_round_bf16=True)
    for r, o in zip(reqs, outs):
        oseq, oaud, omax = ogen.oracle_generate(om, TOK, r["input_ids"], r["attention_mask"], cfg_scale=1.3, num_steps=5,
                                                noise_fn=r["_noise_fn"], forced_tokens=[r["_forced_tokens"]])
        assert torch.equal(o.sequences.cpu(), oseq)
        assert rel_err(o.speech_outputs[0][0], oaud[0][0]) <= 1e-2, rel_err(o.speech_outputs[0][0], oaud[0][0])


def test_interleaved_lanes_over_one_weight_copy(sm):
    """generate_interleaved: the request queue split over TWO engine contexts that share one weight upload (vv_create_shared), one host
    thread and one stream per lane.  Every request must end exactly as generate_continuous() gives it on the single context (to 1e-5:
    the lanes share nothing but read-only weights), in request order; the caller's streamer
    sees every request's own sample index; a second call reuses the lanes; the child survives its owner being destroyed first."""
    from vibevoice_amd.engine import EngineError
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    reqs = _mk_requests(sm, 6, 21)
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos},
            "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.set_ddpm_inference_steps(5)
    tok = types.SimpleNamespace(speech_start_id=S, speech_end_id=E, speech_diffusion_id=D, eos_token_id=X,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    kw = dict(tokenizer=tok, generation_config={"do_sample": False}, cfg_scale=1.3)
    solo = m.generate_continuous(reqs, **kw)
    try:
        fs = FakeStreamer(6)
        two = m.generate_interleaved(reqs, lanes=2, audio_streamer=fs, **kw)
        assert m.last_stats["lanes"] == 2 and sorted(i for sh in m.last_stats["shards"] for i in sh) == list(range(6))
        assert all(len(sh) == 3 for sh in m.last_stats["shards"])
        for a, b in zip(two, solo):
            assert torch.equal(a.sequences.cpu(), b.sequences.cpu())
            assert a.speech_outputs[0].shape == b.speech_outputs[0].shape
            assert rel_err(a.speech_outputs[0], b.speech_outputs[0]) <= 1e-5          # (which rows share a weight pass differs)
        # the streamer saw each request's frames under the request's own index, and one end per request (+ the closing end())
        frames = {i: 0 for i in range(6)}
        for shape, idx in fs.puts:
            for i in idx:
                frames[i] += 1
        assert [frames[i] * 3200 for i in range(6)] == [o.speech_outputs[0].shape[-1] for o in solo]
        assert sorted(i for e_ in fs.ends if e_ is not None for i in e_) == list(range(6)) and fs.ends[-1] is None
        lane = m._lanes[0]
        assert lane.engine.shared_from is sm.eng and lane.engine is not sm.eng
        again = m.generate_interleaved(list(reversed(reqs)), lanes=2, **kw)             # the lanes are reused
        assert m._lanes[0] is lane
        for a, b in zip(again, reversed(solo)):
            assert torch.equal(a.sequences.cpu(), b.sequences.cpu()) and rel_err(a.speech_outputs[0], b.speech_outputs[0]) <= 1e-5
        with pytest.raises(EngineError, match="upload through the parent"):
            lane.engine.upload("lm.norm.weight", torch.ones(sm.lmcfg.hidden))
        with pytest.raises(EngineError, match="itself a shared context|model fields"):
            import dataclasses
            from vibevoice_amd.engine import Engine
            Engine(dataclasses.replace(sm.eng.cfg, lm_layers=sm.eng.cfg.lm_layers + 1), sm.eng.device, share_from=sm.eng)
    finally:
        m.close_lanes()


def test_shared_context_outlives_its_owner():
    """vv_destroy on the owner of shared weights while a child still uses them: the storage lives until the last child goes"""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    s = build_small(synth.LMCfg(), xsplit=3, n_slots=1, max_ctx=256, use_graph=True)
    forced = [[D, D, E, S, D, X]]
    o, h = run_both(s, 1, forced, with_speech=True, seed=5)
    child = s.eng.fork()
    s.eng.close()                                   # owner first
    import gc
    gc.collect()
    torch.cuda.synchronize()
    junk = torch.randn(64 << 20, device=child.device)          # would land in freed weight storage if it had been released
    s.eng = child
    try:
        o2, h2 = run_both(s, 1, forced, with_speech=True, seed=5)
        check(o2, h2)
        assert rel_err(h2[0].speech_outputs[0], h[0].speech_outputs[0]) <= 1e-6
    finally:
        child.close()
        del junk


def test_generate_batch8_desynchronised():
    """Eight utterances in one batch (BASELINE config 4's per-GPU batch), every row on its own token plan: 16-row LM passes,
    16-row diffusion-head GEMV forms, per-utterance codec chains on side streams -- against the oracle loop."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    s = build_small(synth.LMCfg(), xsplit=2, n_slots=8, max_ctx=256, max_rows=16, use_graph=True)
    try:
        B = 8
        g = synth.Gen(88)
        L0 = 14
        ids = torch.full((B, L0), TOK.pad_token_id, dtype=torch.long)
        mask = torch.zeros((B, L0), dtype=torch.long)
        for b in range(B):
            n = L0 - (b % 4)
            row = torch.from_numpy(g.rng.integers(0, 300, (n,)))
            row[-1] = S
            ids[b, L0 - n:] = row
            mask[b, L0 - n:] = 1
        forced = [[D, D, D, D, X], [D, E, S, D, D, X], [D, D, X], [D, D, D, E, S, D, X], [D, X], [D, D, D, D, D, X],
                  [D, E, S, D, E, S, D, X], [D, D, D, X]]
        bank = {}

        def noise_fn(step, n2):
            return bank.setdefault((step, n2), synth.Gen(5000 + step * 17 + n2).normal((n2, 64), 1.0, mat=False))
        om = s.oracle_model(kv_round_bf16=True)
        otr = ogen.Trace()
        oseq, oaud, omax = ogen.oracle_generate(om, TOK, ids, mask, cfg_scale=1.3, num_steps=5, noise_fn=noise_fn,
                                                forced_tokens=forced, trace=otr)
        cfgd = {"decoder_config": {"max_position_embeddings": s.lmcfg.max_pos},
                "diffusion_head_config": {"ddpm_num_inference_steps": 5},
                "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
        m = VibeVoiceForConditionalGenerationInference(cfgd, s.eng, model_dtype=torch.float32)
        m.set_speech_factors(s.scaling, s.bias)
        m.set_ddpm_inference_steps(5)
        tok = types.SimpleNamespace(speech_start_id=S, speech_end_id=E, speech_diffusion_id=D, eos_token_id=X,
                                    bos_token_id=None, pad_token_id=TOK.pad_token_id)
        htr = ogen.Trace()
        out = m.generate(input_ids=ids, attention_mask=mask, cfg_scale=1.3, tokenizer=tok, generation_config={"do_sample": False},
                         _forced_tokens=forced, _noise_fn=noise_fn, _trace=htr, show_progress_bar=False)
        # xsplit=2 (the two-term activation mode the 16-row GEMV forms exist in): ~fp24-class arithmetic
        check((oseq, oaud, omax, otr), (out, htr), lat_tol=2e-2, wav_tol=3e-2)
        assert_kernel_nodes_only(s.eng)
    finally:
        s.eng.close()


def test_graph_cache_is_bounded(monkeypatch):
    """the hipGraph cache must not grow without bound.  With a cap of 4 executables a generate() run keeps
    evicting and re-capturing and still matches the oracle."""
    monkeypatch.setenv("VVHIP_GRAPH_CAP", "4")
    s = build_small(synth.LMCfg(), xsplit=3, n_slots=1, max_ctx=512, use_graph=True)
    try:
        forced = [[D, D, D, D, E, S, D, D, D, X]]
        o, h = run_both(s, 1, forced, with_speech=True)
        check(o, h)
        assert 0 < s.eng.stat(1) <= 4
    finally:
        s.eng.close()


@pytest.mark.parametrize("extra", [{"top_k": 0}, {"top_k": 319}], ids=["valid-rows", "full-vocabulary"])
def test_sampling_path_at_vanishing_temperature_equals_greedy_batch2(sm, extra):
    """(ids = "full-vocabulary": the same through vv_lm_logits_full and the full-vocabulary processors -- top-k 319 of the toy
    vocabulary's 320 removes nothing that matters -- then the valid-token constraint.)
    do_sample=True on the engine for a desynchronised batch of TWO: the sampling branch builds full-vocabulary rows from the
    dense [n][n_valid] logits block (vibevoice_amd/modeling.py: the branch of ADVICE r1's row-stride finding that no GPU test
    reached -- the reference draws on the device generator, so a seeded run cannot be compared with the CPU oracle).  At
    temperature 1e-3 the categorical draw is the argmax (the tiny model's logit gaps are ~1e-1), so the sampled run must
    reproduce the greedy run -- which test_generate_greedy_batch2_free_running holds to the oracle -- token by token and
    sample by sample; a row reading another row's logits would not."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    o, h = run_both(sm, 2, None, with_speech=True, seed=61, max_new_tokens=10)
    check(o, h)
    ids, mask, sim, st, smk = make_inputs(sm, 2, True, 61)
    g = synth.Gen(62)
    pre = (g.normal((2,), 1.0, mat=False), g.normal((2, 3, 64), 1.0, mat=False))
    bank = {}

    def noise_fn(step, n2):
        if (step, n2) not in bank:
            bank[(step, n2)] = synth.Gen(61 * 1000 + step).normal((n2, 64), 1.0, mat=False)
        return bank[(step, n2)]
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.set_ddpm_inference_steps(5)
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    torch.manual_seed(5)
    out = m.generate(input_ids=ids, attention_mask=mask, speech_tensors=st, speech_masks=smk, speech_input_mask=sim, cfg_scale=1.3,
                     tokenizer=tok, max_new_tokens=10, generation_config={"do_sample": True, "temperature": 1e-3, **extra},
                     _noise_fn=noise_fn, _prefill_noise=pre, show_progress_bar=False)
    greedy = h[0]
    assert torch.equal(out.sequences.cpu(), greedy.sequences.cpu())
    for a, b in zip(out.speech_outputs, greedy.speech_outputs):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a.cpu(), b.cpu())


def test_abi_refuses_out_of_range_arguments(sm):
    """The C ABI validates what would otherwise be an out-of-bounds device access and returns an error code + text
    (vv_last_error) that the ctypes layer raises as EngineError: positions beyond max_ctx, cache ids beyond 2 * n_slots, more
    rows than max_rows, a KV import past the end of the cache.  (The reference fails with PyTorch index errors at the
    corresponding places; a native engine must not turn them into silent memory corruption.)"""
    from vibevoice_amd.engine import EngineError
    eng = sm.eng
    H = sm.lmcfg.hidden
    x = eng.new(4, H)
    y = eng.new(4, H)
    with pytest.raises(EngineError, match="exceeds max_ctx"):
        eng.lm_forward([(0, eng.max_ctx)], x, y)
    with pytest.raises(EngineError, match="cache id"):
        eng.lm_forward([(2 * eng.cfg.n_slots, 0)], x, y)
    with pytest.raises(EngineError, match="n_rows"):
        big = eng.new(eng.cfg.max_rows + 1, H)
        eng.lm_forward([(0, j) for j in range(eng.cfg.max_rows + 1)], big, big)
    kvh, d = sm.lmcfg.kv_heads, sm.lmcfg.head_dim
    k = torch.zeros(kvh, 8, d, device=eng.device)
    with pytest.raises(EngineError, match="exceed max_ctx"):
        eng.kv_import_at(0, 0, eng.max_ctx - 4, k, k)
    eng.sync()
    # the context is still usable after refused calls
    eng.lm_forward([(0, 0)], x[:1], y[:1])
    eng.sync()
    assert torch.isfinite(y[:1]).all()


def test_generate_under_the_gradio_scheduler(sm):
    """generate() on the engine after demo/gradio_demo.py:142-146's scheduler swap (sde-dpmsolver++): forced desynchronised
    batch of 2 with a voice prompt against the oracle loop fed the same initial noise and the same per-step variance noise
    (the oracle loop itself is pinned to the reference's generate() under that swap: tests/golden/generate_sde_b*.npz)."""
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    forced = [[D, D, D, E, S, D, D, X], [D, D, E, S, D, X]]
    ids, mask, sim, st, smk = make_inputs(sm, 2, True, 83)
    g = synth.Gen(84)
    pre = (g.normal((2,), 1.0, mat=False), g.normal((2, 3, 64), 1.0, mat=False))
    bank, sbank = {}, {}

    def noise_fn(step, n2):
        if (step, n2) not in bank:
            bank[(step, n2)] = synth.Gen(83 * 1000 + step).normal((n2, 64), 1.0, mat=False)
        return bank[(step, n2)]

    def sde_fn(step, N, n2):
        if (step, N, n2) not in sbank:
            sbank[(step, N, n2)] = synth.Gen(83 * 2000 + step).normal((N, n2, 64), 1.0, mat=False)
        return sbank[(step, N, n2)]
    om = sm.oracle_model(kv_round_bf16=True)
    otr = ogen.Trace()
    oseq, oaud, omax = ogen.oracle_generate(om, TOK, ids, mask, st, smk, sim, cfg_scale=1.3, num_steps=5, noise_fn=noise_fn,
                                            prefill_noise=pre, forced_tokens=forced, trace=otr,
                                            algorithm_type="sde-dpmsolver++", sde_noise_fn=sde_fn)
    cfgd = {"decoder_config": {"max_position_embeddings": sm.lmcfg.max_pos}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
            "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}
    m = VibeVoiceForConditionalGenerationInference(cfgd, sm.eng, model_dtype=torch.float32)
    m.set_speech_factors(sm.scaling, sm.bias)
    m.model.noise_scheduler = m.model.noise_scheduler.from_config(m.model.noise_scheduler.config, algorithm_type="sde-dpmsolver++",
                                                                  beta_schedule="squaredcos_cap_v2")
    m.set_ddpm_inference_steps(num_steps=5)
    tok = types.SimpleNamespace(speech_start_id=TOK.speech_start_id, speech_end_id=TOK.speech_end_id,
                                speech_diffusion_id=TOK.speech_diffusion_id, eos_token_id=TOK.eos_token_id,
                                bos_token_id=None, pad_token_id=TOK.pad_token_id)
    htr = ogen.Trace()
    try:
        out = m.generate(input_ids=ids, attention_mask=mask, speech_tensors=st, speech_masks=smk, speech_input_mask=sim, cfg_scale=1.3,
                         tokenizer=tok, generation_config={"do_sample": False}, _forced_tokens=forced, _noise_fn=noise_fn,
                         _sde_noise_fn=sde_fn, _prefill_noise=pre, _trace=htr, show_progress_bar=False)
    finally:
        sm.eng.set_num_steps(5)                      # the module fixture goes back to the deterministic table
    check((oseq, oaud, omax, otr), (out, htr))
