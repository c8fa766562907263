"""ORACLE (test infrastructure) -- restatement of the Streaming-0.5B generate() loop.

Follows vibevoice/modular/modeling_vibevoice_streaming_inference.py:
  forward_lm (text LM: lower layers, final norm = Identity)          :181-241
  forward_tts_lm (splice hidden states, + tts_input_types, EOS head)  :243-318
  generate: windows of 5 text tokens / 6 speech frames                :40-42, :465-725
  sample_speech_tokens                                                :727-751
and modeling_vibevoice_streaming.py:42-53 (BinaryClassifier), :134-146 (split LM).
HF cache plumbing is replaced by compact caches (batch size is 1 in the reference, :511).
The four prefilled states of a voice preset (`all_prefilled_outputs`: lm, tts_lm, neg_lm,
neg_tts_lm) are inputs; neg_lm is never advanced by the reference loop.

PARITY UNPINNED for the orchestration (the reference loop cannot execute under
transformers 5.x); every arithmetic stage is pinned by tests/golden.
"""
from dataclasses import dataclass, field
from typing import Callable, Optional

import torch
import torch.nn.functional as F

from . import codec, connector, dpm, head

TTS_TEXT_WINDOW_SIZE = 5
TTS_SPEECH_WINDOW_SIZE = 6


@dataclass
class StreamingOracleModel:
    lm: object                  # Qwen2Oracle over the text-LM layers (final norm unused)
    tts_lm: object              # Qwen2Oracle over the TTS-LM layers (with its final norm)
    tts_types: torch.Tensor     # [2, H]  tts_input_types.weight
    eos: dict                   # fc1.weight/bias, fc2.weight/bias
    head_w: dict
    head_layers: int
    ac_w: dict
    ac_conn: dict
    ratios: list
    dec_depths: list
    scaling: float
    bias: float
    head_eps: float = 1e-5
    codec_eps: float = 1e-5
    t_cast_dtype: Optional[torch.dtype] = None      # bf16 GPU path: the timestep is cast to the model dtype (999 -> 1000)


@dataclass
class Preset:
    """State of the four prefilled branches (compact caches + last hidden state of the TTS branches)."""
    lm_cache: object
    tts_cache: object
    neg_tts_cache: object
    tts_last: torch.Tensor      # [H]
    neg_tts_last: torch.Tensor  # [H]


def eos_logit(eos, h):
    return F.linear(torch.relu(F.linear(h, eos["fc1.weight"], eos["fc1.bias"])), eos["fc2.weight"], eos["fc2.bias"])


def make_preset(m: StreamingOracleModel, prompt_ids, neg_id):
    """A stand-in for demo/voices/streaming_model/*.pt built with the oracle itself
    (the reference ships only the pickled results, not the code that made them)."""
    lm_c, tts_c, neg_lm_c, neg_tts_c = m.lm.new_cache(), m.tts_lm.new_cache(), m.lm.new_cache(), m.tts_lm.new_cache()
    h = m.lm.forward(m.lm.embed(prompt_ids), lm_c, final_norm=False)
    t = m.tts_lm.forward(h + m.tts_types[1], tts_c)
    hn = m.lm.forward(m.lm.embed(torch.tensor([neg_id])), neg_lm_c, final_norm=False)
    tn = m.tts_lm.forward(hn + m.tts_types[1], neg_tts_c)
    return Preset(lm_c, tts_c, neg_tts_c, t[-1], tn[-1])


def oracle_generate_streaming(m: StreamingOracleModel, preset: Preset, tts_text_ids, cfg_scale, num_steps,
                              noise_fn: Callable, max_length: int, trace: Optional[list] = None):
    """tts_text_ids: LongTensor [N].  Returns (n_tokens, audio [1, samples] or None, reach_max, finished)."""
    lm_c, tts_c, neg_c = preset.lm_cache, preset.tts_cache, preset.neg_tts_cache
    tts_last, neg_last = preset.tts_last, preset.neg_tts_last
    n_tok = tts_c.length
    finished = False
    reach_max = False
    chunks = []
    state = {}
    w = 0
    frame = 0
    while True:
        if finished:
            break
        cur = tts_text_ids[w * TTS_TEXT_WINDOW_SIZE:(w + 1) * TTS_TEXT_WINDOW_SIZE]
        w += 1
        if cur.numel() > 0:
            n_tok += cur.numel()
            if n_tok > max_length:
                reach_max = True
                break
            h = m.lm.forward(m.lm.embed(cur), lm_c, final_norm=False)
            tts_last = m.tts_lm.forward(h + m.tts_types[1], tts_c)[-1]
        for i in range(TTS_SPEECH_WINDOW_SIZE):
            noise = noise_fn(frame, 2)
            lat = dpm.sample_speech_tokens(
                lambda x, t, c: head.head_forward(m.head_w, x, t, c, m.head_layers, m.head_eps),
                tts_last[None], neg_last[None], cfg_scale, num_steps, noise, m.t_cast_dtype)
            scaled = lat / m.scaling - m.bias
            chunk = codec.decoder_forward(m.ac_w, scaled[0][None, :, None], m.ratios, m.dec_depths, state, m.codec_eps)
            if not finished:
                chunks.append(chunk[0])
            emb = connector.connector_forward(m.ac_conn, lat)          # [1, H]
            frame += 1
            n_tok += 1
            if n_tok > max_length:
                break
            x = emb + m.tts_types[0]
            tts_last = m.tts_lm.forward(x, tts_c)[-1]
            neg_last = m.tts_lm.forward(x, neg_c)[-1]
            logit = eos_logit(m.eos, tts_last[None])[0, 0]
            if trace is not None:
                trace.append({"latent": lat.clone(), "tts_last": tts_last.clone(), "eos": float(logit)})
            if torch.sigmoid(logit).item() > 0.5:
                finished = True
        if n_tok > max_length:
            if not finished:
                reach_max = True
            break
    audio = torch.cat(chunks, dim=-1) if chunks else None
    return n_tok, audio, reach_max, finished
