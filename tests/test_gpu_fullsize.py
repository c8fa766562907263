"""Parity at the real layer shapes (BASELINE.json configs[1], VibeVoice-1.5B widths): hidden 1536,
12/2 heads x 128, MLP 8960, diffusion head 1536x4608 x 4 layers, the full 3200x acoustic decoder and
semantic encoder (n_filters 32, depths 3-3-3-3-3-3-8).  The layer COUNT of the LM (4 instead of 28) and the
vocabulary (2048) are reduced so the CPU oracle finishes in about a minute; every kernel runs at its real
per-layer shape.  Weights are bf16-representable, engine in xsplit=3, so the bound is fp32-class."""
import copy
import types

import pytest
import torch

from gpu_util import rel_err
from oracle import generate as ogen
from oracle import lm as olm

pytestmark = pytest.mark.gpu


def test_generate_at_1p5b_layer_shapes():
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg = copy.deepcopy(CONFIGS["1.5b"])
    cfg["decoder_config"]["num_hidden_layers"] = 4
    cfg["decoder_config"]["vocab_size"] = 2048
    cfg["decoder_config"]["max_position_embeddings"] = 512
    gen = torch.Generator().manual_seed(0)
    sd = {k: synthetic.random_tensor(k, shp, gen, "cpu", torch.bfloat16) for k, shp in synthetic.param_shapes(cfg).items()}
    model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.float32, None, n_slots=1, max_ctx=512,
                                                                       xsplit=3, use_graph=True, enc_frames=2)
    model.set_speech_factors(0.2, -0.05)
    model.set_ddpm_inference_steps(5)
    model_bf16 = None
    try:
        d = cfg["decoder_config"]

        def sub(prefix):
            return {k[len(prefix):]: v.float() for k, v in sd.items() if k.startswith(prefix)}
        lm_w = sub("model.language_model.")
        lm = olm.Qwen2Oracle(lm_w, 4, d["num_attention_heads"], d["num_key_value_heads"], 128, d["rope_theta"],
                             d["rms_norm_eps"], kv_round_bf16=True)
        depths = [3, 3, 3, 3, 3, 3, 8]
        om = ogen.OracleModel(lm=lm, lm_head=lm_w["embed_tokens.weight"], head_w=sub("model.prediction_head."), head_layers=4,
                              ac_w=sub("model.acoustic_tokenizer."), sem_w=sub("model.semantic_tokenizer."),
                              ac_conn=sub("model.acoustic_connector."), sem_conn=sub("model.semantic_connector."),
                              ratios=[8, 5, 5, 4, 2, 2], enc_depths=depths, dec_depths=list(reversed(depths)),
                              sem_depths=depths, scaling=0.2, bias=-0.05, max_position_embeddings=512)
        T = types.SimpleNamespace(speech_start_id=2001, speech_end_id=2002, speech_diffusion_id=2003, eos_token_id=2004,
                                  bos_token_id=None, pad_token_id=2005)
        tok = ogen.TokenIds(2001, 2002, 2003, 2004, None, 2005)
        g = torch.Generator().manual_seed(1)
        n_voice = 2
        ids = torch.randint(0, 2000, (1, 24), generator=g)
        sim = torch.zeros(1, 24, dtype=torch.bool)
        ids[0, 5:5 + n_voice] = T.speech_diffusion_id
        sim[0, 5:5 + n_voice] = True
        ids[0, -1] = T.speech_start_id
        mask = torch.ones_like(ids)
        wav = torch.rand(1, n_voice * 3200, generator=g) * 0.2 - 0.1
        smask = torch.ones(1, n_voice, dtype=torch.bool)
        pre = (torch.randn(1, generator=g), torch.randn(1, n_voice, 64, generator=g))
        noise = {}

        def noise_fn(step, n2):
            if step not in noise:
                noise[step] = torch.randn(n2, 64, generator=torch.Generator().manual_seed(100 + step))
            return noise[step]
        D, E, S, X = T.speech_diffusion_id, T.speech_end_id, T.speech_start_id, T.eos_token_id
        forced = [[D, D, E, S, D, X]]
        otr, htr = ogen.Trace(), ogen.Trace()
        with torch.no_grad():
            oseq, oaud, _ = ogen.oracle_generate(om, tok, ids, mask, wav, smask, sim, cfg_scale=1.3, num_steps=5, noise_fn=noise_fn,
                                                 prefill_noise=pre, forced_tokens=forced, trace=otr)
        out = model.generate(input_ids=ids, attention_mask=mask, speech_tensors=wav, speech_masks=smask, speech_input_mask=sim,
                             cfg_scale=1.3, tokenizer=T, generation_config={"do_sample": False}, _forced_tokens=forced,
                             _noise_fn=noise_fn, _prefill_noise=pre, _trace=htr, show_progress_bar=False)
        assert torch.equal(out.sequences.cpu(), oseq)
        for a, b in zip(htr.latents, otr.latents):
            assert rel_err(a, b) <= 5e-3, rel_err(a, b)
        for a, b in zip(htr.pos_hidden, otr.pos_hidden):
            assert rel_err(a, b) <= 5e-3, rel_err(a, b)
        assert out.speech_outputs[0].shape[-1] == 3 * 3200
        assert rel_err(out.speech_outputs[0][0], oaud[0][0]) <= 1e-2
        # ---- the configuration bench.py times: xsplit=1 (bf16 activations in the MFMAs) + hipGraph replay, at these widths.
        # Teacher-forced per step against the same oracle trace: latent / hidden rel-L2 <= 5e-2, frame RMS +-0.5 dB.
        model.engine.close()
        model_bf16 = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.float32, None, n_slots=1, max_ctx=512,
                                                                                xsplit=1, use_graph=True, enc_frames=2)
        model_bf16.set_speech_factors(0.2, -0.05)
        model_bf16.set_ddpm_inference_steps(5)
        btr = ogen.Trace()
        outb = model_bf16.generate(input_ids=ids, attention_mask=mask, speech_tensors=wav, speech_masks=smask, speech_input_mask=sim,
                                   cfg_scale=1.3, tokenizer=T, generation_config={"do_sample": False}, _forced_tokens=forced,
                                   _noise_fn=noise_fn, _prefill_noise=pre, _trace=btr, show_progress_bar=False,
                                   _teacher_embeds=lambda step, rows: otr.next_embeds[step][rows])
        assert torch.equal(outb.sequences.cpu(), oseq)
        for a, b in zip(btr.latents, otr.latents):
            assert rel_err(a, b) <= 5e-2, rel_err(a, b)
        for a, b in zip(btr.pos_hidden, otr.pos_hidden):
            assert rel_err(a, b) <= 5e-2, rel_err(a, b)
        wa, wb = outb.speech_outputs[0][0].float().cpu(), oaud[0][0]
        for f in range(3):
            fa, fb = wa[f * 3200:(f + 1) * 3200], wb[f * 3200:(f + 1) * 3200]
            assert abs(20 * torch.log10(fa.norm() / fb.norm())) <= 0.5
    finally:
        model.engine.close()
        if model_bf16 is not None:
            model_bf16.engine.close()


@pytest.mark.parametrize("mode", ["full", "heavy"])
def test_codec_chain_batch_at_real_widths(mode, monkeypatch):
    """vv_codec_chain_batch (several utterances' tokenizer chains in one call, the weight-heavy stages slot-batched) at the
    real tokenizer widths, bf16 mode + hipGraph: against the CPU oracle's streaming decoder / encoder per slot, and against
    the engine's own one-utterance path run on a spare slot with the same inputs.  Covers a mid-stream reset of one slot,
    a changing slot set, and a slot that alternates between the batched and the single path.
    mode "full": every stage of both nets slot-batched (the default); "heavy": only the T <= 8 stages, the rest per utterance
    on forked graph branches (VVHIP_BATCH_CODEC=heavy)."""
    from oracle import codec
    if mode == "heavy":
        monkeypatch.setenv("VVHIP_BATCH_CODEC", "heavy")
    else:
        monkeypatch.delenv("VVHIP_BATCH_CODEC", raising=False)
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    cfg = copy.deepcopy(CONFIGS["1.5b"])
    cfg["decoder_config"]["num_hidden_layers"] = 1
    cfg["decoder_config"]["vocab_size"] = 2048
    cfg["decoder_config"]["max_position_embeddings"] = 64
    gen = torch.Generator().manual_seed(3)
    sd = {k: synthetic.random_tensor(k, shp, gen, "cpu", torch.bfloat16) for k, shp in synthetic.param_shapes(cfg).items()}
    model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.float32, None, n_slots=5, max_ctx=64,
                                                                       xsplit=1, use_graph=True, enc_frames=1)
    scaling, bias = 0.2, -0.05
    model.set_speech_factors(scaling, bias)
    eng = model.engine
    try:
        ac_w = {k[len("model.acoustic_tokenizer."):]: v.float() for k, v in sd.items() if k.startswith("model.acoustic_tokenizer.")}
        sem_w = {k[len("model.semantic_tokenizer."):]: v.float() for k, v in sd.items() if k.startswith("model.semantic_tokenizer.")}
        depths = [3, 3, 3, 3, 3, 3, 8]
        ratios = [8, 5, 5, 4, 2, 2]
        st_dec = {s: {} for s in range(5)}
        st_sem = {s: {} for s in range(5)}
        g = torch.Generator().manual_seed(4)
        # (slot set of the batched call, slots run through the one-utterance path) per frame; slot 1 mirrors slot 0's inputs
        plan = [([0, 2, 3], [1]), ([0, 2, 3], [1]), ([3, 0], [1, 2]), ([0, 2, 3, 4], [1]), ([4, 3, 2, 0], [1]), ([0, 2], [1])]
        lat_of = {}
        worst_a = worst_s = worst_pair = 0.0
        for t, (batch, single) in enumerate(plan):
            if t == 2:                       # slot 3 restarts mid-stream (speech_end -> the caches are zeroed)
                with torch.cuda.stream(eng.stream):
                    eng.codec_reset(3)
                codec.zero_state(st_dec[3]); codec.zero_state(st_sem[3])
            for s in set(batch + single):
                lat_of[s] = torch.randn(64, generator=g) * 0.7
            lat_of[1] = lat_of[0].clone()
            n = len(batch)
            lat = torch.stack([lat_of[s] for s in batch]).to(eng.device)
            audio = eng.new(n, 3200)
            sem = eng.new(n, 128)
            with torch.cuda.stream(eng.stream):
                eng.codec_chain_batch(batch, lat, audio, sem)
                outs = {}
                for s in single:
                    a1, s1 = eng.new(3200), eng.new(128)
                    eng.codec_decode(s, lat_of[s][None].to(eng.device), a1)
                    eng.semantic_encode(s, a1, s1)
                    outs[s] = (a1, s1)
            eng.sync()
            for j, s in enumerate(batch):
                outs[s] = (audio[j], sem[j])
            for s, (a, se) in outs.items():
                x = lat_of[s] / scaling - bias
                ra = codec.decoder_forward(ac_w, x[None, :, None], ratios, list(reversed(depths)), st_dec[s], 1e-5)[0, 0]
                rs = codec.encoder_forward(sem_w, ra[None, None], ratios, depths, st_sem[s], 1e-5)[0, :, 0]
                worst_a = max(worst_a, rel_err(a, ra)); worst_s = max(worst_s, rel_err(se, rs))
                assert rel_err(a, ra) <= 5e-2, (t, s, rel_err(a, ra))
                assert rel_err(se, rs) <= 5e-2, (t, s, rel_err(se, rs))
            # the batched path (slot 0) and the one-utterance path (slot 1) saw identical inputs from the start
            pa, ps = rel_err(outs[0][0], outs[1][0].cpu()), rel_err(outs[0][1], outs[1][1].cpu())
            worst_pair = max(worst_pair, pa, ps)
            assert pa <= 1e-2 and ps <= 1e-2, (t, pa, ps)
        print(f"codec_chain_batch[{mode}]: audio vs oracle {worst_a:.2e}, semantic vs oracle {worst_s:.2e}, batched vs single {worst_pair:.2e}")
    finally:
        eng.close()
