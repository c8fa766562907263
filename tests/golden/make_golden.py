"""Generate tests/golden/*.npz by running the REFERENCE's own classes.

Run only in the build container (needs /root/reference):
    python tests/golden/make_golden.py

What is executed is reference code, imported through oracle/refshim.py:
  vibevoice.schedule.dpm_solver.DPMSolverMultistepScheduler
  vibevoice.modular.modular_vibevoice_diffusion_head.VibeVoiceDiffusionHead
  vibevoice.modular.modular_vibevoice_tokenizer.{VibeVoiceAcousticTokenizerModel,
      VibeVoiceSemanticTokenizerModel, VibeVoiceTokenizerStreamingCache}
  vibevoice.modular.modeling_vibevoice.SpeechConnector
and, for the LM (third-party arithmetic, see oracle/lm.py), the installed
transformers Qwen2Model.  Weights come from tests/synth.py (seeded, small
shapes); only inputs/outputs + a weight checksum are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import refshim  # noqa: E402

refshim.install()
import synth  # noqa: E402


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: tuple(np.shape(v)) for k, v in out.items()})


@torch.no_grad()
def gen_dpm_and_head():
    from vibevoice.schedule.dpm_solver import DPMSolverMultistepScheduler
    from vibevoice.modular.modular_vibevoice_diffusion_head import VibeVoiceDiffusionHead
    from vibevoice.modular.configuration_vibevoice import VibeVoiceDiffusionHeadConfig

    arrs = {}
    for n in (5, 10, 20):
        s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_schedule="cosine",
                                        prediction_type="v_prediction")
        s.set_timesteps(n)
        arrs[f"timesteps_{n}"] = s.timesteps
        arrs[f"sigmas_{n}"] = s.sigmas
    save("dpm_schedule.npz", **arrs)

    hc = synth.HeadCfg()
    w = synth.head_weights(hc)
    cfg = VibeVoiceDiffusionHeadConfig(hidden_size=hc.hidden, head_layers=hc.layers,
                                       head_ffn_ratio=hc.ffn_ratio, rms_norm_eps=hc.eps,
                                       latent_size=hc.latent)
    head = VibeVoiceDiffusionHead(cfg).eval()
    head.load_state_dict(w, strict=True)
    g = synth.Gen(100)
    noisy = g.normal((4, hc.latent), 1.0, mat=False)
    cond = g.normal((4, hc.hidden), 1.0, mat=False)
    outs = {}
    ts = [999, 500, 100, 3]
    for t in ts:
        tt = torch.full((4,), float(t))
        outs[f"out_{t}"] = head(noisy, tt, cond)
    save("head_forward.npz", noisy=noisy, cond=cond, ts=np.array(ts), wsum=synth.checksum(w), **outs)

    # full CFG sampler: the 10 lines of sample_speech_tokens
    # (modeling_vibevoice_inference.py:697-710) driven with the reference scheduler/head
    for n_steps in (5, 10):
        s = DPMSolverMultistepScheduler(num_train_timesteps=1000, beta_schedule="cosine",
                                        prediction_type="v_prediction")
        g = synth.Gen(200 + n_steps)
        pos = g.normal((2, hc.hidden), 1.0, mat=False)
        neg = g.normal((2, hc.hidden), 1.0, mat=False)
        noise = g.normal((4, hc.latent), 1.0, mat=False)
        cfg_scale = 1.3
        s.set_timesteps(n_steps)
        condition = torch.cat([pos, neg], dim=0)
        speech = noise.clone()
        per_step = []
        for t in s.timesteps:
            half = speech[: len(speech) // 2]
            combined = torch.cat([half, half], dim=0)
            eps = head(combined, t.repeat(combined.shape[0]).to(combined), condition=condition)
            cond_eps, uncond_eps = torch.split(eps, len(eps) // 2, dim=0)
            half_eps = uncond_eps + cfg_scale * (cond_eps - uncond_eps)
            eps = torch.cat([half_eps, half_eps], dim=0)
            speech = s.step(eps, t, speech).prev_sample
            per_step.append(speech[:2].clone())
        save(f"sampler_{n_steps}.npz", pos=pos, neg=neg, noise=noise, cfg_scale=cfg_scale,
             latent=speech[:2], per_step=torch.stack(per_step), wsum=synth.checksum(w))


@torch.no_grad()
def gen_codec():
    from vibevoice.modular.configuration_vibevoice import (VibeVoiceAcousticTokenizerConfig,
                                                           VibeVoiceSemanticTokenizerConfig)
    from vibevoice.modular.modular_vibevoice_tokenizer import (VibeVoiceAcousticTokenizerModel,
                                                               VibeVoiceSemanticTokenizerModel,
                                                               VibeVoiceTokenizerStreamingCache)
    cc = synth.CodecCfg()
    acfg = VibeVoiceAcousticTokenizerConfig(encoder_n_filters=cc.n_filters, decoder_n_filters=cc.n_filters,
                                            vae_dim=cc.vae_dim, encoder_depths=cc.depth_str,
                                            layernorm_eps=cc.eps)
    ac = VibeVoiceAcousticTokenizerModel(acfg).eval()
    w = {}
    w.update(synth.encoder_weights(cc, seed=2))
    w.update(synth.decoder_weights(cc, seed=3))
    ac.load_state_dict(w, strict=True)

    g = synth.Gen(300)
    nfr = 7
    lat = g.normal((1, nfr, cc.vae_dim), 1.0, mat=False)           # [B, T, vae]
    cache = VibeVoiceTokenizerStreamingCache()
    idx = torch.tensor([0])
    chunks = []
    reset_after = 3
    for t in range(nfr):
        chunks.append(ac.decode(lat[:, t:t + 1], cache=cache, sample_indices=idx, use_cache=True))
        if t == reset_after:
            cache.set_to_zero(idx)
    stream = torch.cat(chunks, dim=-1)
    full_a = ac.decode(lat[:, :reset_after + 1])                  # non-streaming, first segment
    full_b = ac.decode(lat[:, reset_after + 1:])
    save("codec_decode.npz", latents=lat, stream=stream, reset_after=reset_after,
         nonstream_a=full_a, nonstream_b=full_b, wsum=synth.checksum(w))

    # acoustic encoder, non-streaming (voice-prompt path, _process_speech_inputs :154)
    wav = g.uniform((2, 3200 * 3), -0.5, 0.5)
    mean = ac.encode(wav.unsqueeze(1)).mean                        # [2, 3, vae]
    save("acoustic_encode.npz", wav=wav, mean=mean, wsum=synth.checksum(w))

    # semantic encoder, streaming, incl. mid-stream zeroing
    sc = synth.CodecCfg(vae_dim=128)
    scfg = VibeVoiceSemanticTokenizerConfig(encoder_n_filters=sc.n_filters, vae_dim=sc.vae_dim,
                                            encoder_depths=sc.depth_str, layernorm_eps=sc.eps)
    sem = VibeVoiceSemanticTokenizerModel(scfg).eval()
    sw = synth.encoder_weights(sc, seed=7)
    sem.load_state_dict(sw, strict=True)
    nfr = 6
    audio = g.uniform((1, 1, 3200 * nfr), -0.5, 0.5)
    cache = VibeVoiceTokenizerStreamingCache()
    outs = []
    for t in range(nfr):
        outs.append(sem.encode(audio[:, :, t * 3200:(t + 1) * 3200], cache=cache, sample_indices=idx,
                               use_cache=True).mean)
        if t == 2:
            cache.set_to_zero(idx)
    stream = torch.cat(outs, dim=1)                                # [1, nfr, 128]
    full_a = sem.encode(audio[:, :, :3 * 3200]).mean
    save("semantic_encode.npz", audio=audio, stream=stream, reset_after=2, nonstream_a=full_a,
         wsum=synth.checksum(sw))


@torch.no_grad()
def gen_connector():
    from vibevoice.modular.modeling_vibevoice import SpeechConnector
    for name, din in (("ac", 64), ("sem", 128)):
        H = 96
        w = synth.connector_weights(din, H, seed=4 if name == "ac" else 8)
        c = SpeechConnector(din, H).eval()
        c.load_state_dict(w, strict=True)
        x = synth.Gen(400).normal((3, din), 1.0, mat=False)
        save(f"connector_{name}.npz", x=x, y=c(x), wsum=synth.checksum(w))


@torch.no_grad()
def gen_lm():
    from transformers import Qwen2Config, Qwen2Model
    for tag, cfg in (("d64", synth.LMCfg()), ("d128", synth.LMCfg(hidden=256, heads=2, kv_heads=1, inter=384)),
                     ("gqa", synth.LMCfg(hidden=256, heads=4, kv_heads=2, inter=320, layers=3))):
        w = synth.lm_weights(cfg)
        hc = Qwen2Config(hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                         num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                         vocab_size=cfg.vocab, max_position_embeddings=cfg.max_pos,
                         rope_theta=cfg.theta, rms_norm_eps=cfg.eps, tie_word_embeddings=True,
                         attn_implementation="eager")
        m = Qwen2Model(hc).eval()
        m.load_state_dict(w, strict=True)
        g = synth.Gen(500)
        L0, nd = 13, 5
        ids = torch.from_numpy(g.rng.integers(0, cfg.vocab, (1, L0)))
        out = m(input_ids=ids, use_cache=True)
        hs = [out.last_hidden_state[0]]
        past = out.past_key_values
        dec_in = g.normal((nd, cfg.hidden), 1.0, mat=False)
        for i in range(nd):
            out = m(inputs_embeds=dec_in[i][None, None], past_key_values=past, use_cache=True)
            past = out.past_key_values
            hs.append(out.last_hidden_state[0])
        save(f"lm_{tag}.npz", ids=ids, dec_in=dec_in, prefill_hidden=hs[0],
             decode_hidden=torch.cat(hs[1:], dim=0), wsum=synth.checksum(w))


if __name__ == "__main__":
    torch.manual_seed(0)
    gen_dpm_and_head()
    gen_codec()
    gen_connector()
    gen_lm()
