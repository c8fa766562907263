"""ORACLE (test infrastructure) -- restatement of the multi-speaker generate() loop.

Follows vibevoice/modular/modeling_vibevoice_inference.py:
  _process_speech_inputs (voice-prompt prefill)     :149-163
  setup (max_steps, valid tokens, negative prompt)   :372-422
  hot loop                                            :432-675
  sample_speech_tokens                                :697-710
with the HF-GenerationMixin plumbing (DynamicCache, attention masks,
cache_position) replaced by its arithmetic meaning on *compact per-utterance
caches* (SURVEY.md 8c): a left-padded row's pads carry no information, position
ids are cumsum(mask)-1 == index in the compact cache.

Negative (CFG) branch (:379-386, :503-516, :549-565, :576-624): restated LITERALLY (class NegativeRow below: every entry ever
appended, the attention mask over them, the correction counter), not in compact form.  An earlier compact form ("the spurious
entry of a non-diffusing row is dropped; <speech_start> restarts the context empty") is the net effect almost everywhere, and
exactly wrong in one case the fuzz tool found (tools/fuzz_generate_vs_reference.py): the correction guards its mask shift and its
K/V shift differently, so a row holding one valid entry keeps the NEW entry and loses the old one.

PARITY PINNED: the reference's own generate() runs in the build container under transformers 5.15 through the
API shims of oracle/refshim.install_generate_shims(); tests/golden/make_golden.py::gen_generate recorded it on the
tiny seeded model (generate_forced_b1 / _b2, generate_greedy_b1, generate_cap_b1, generate_ragged_voice_b1,
generate_sampled_b1, generate_sde_b1 / _b2 under the gradio demo's stochastic scheduler) and tests/test_oracle_golden.py holds this loop to those files (token sequences identical,
waveform rel-L2 <= 1e-4).  Caveat: the pinned dependency is transformers==4.51.3 (pyproject.toml:22), the recording
ran on 5.15 with its 4.51.3 behaviours restored by the shims.
"""
from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch
import torch.nn.functional as F

from . import codec, connector, dpm, head


@dataclass
class TokenIds:
    speech_start_id: int
    speech_end_id: int
    speech_diffusion_id: int
    eos_token_id: int
    bos_token_id: Optional[int] = None
    pad_token_id: Optional[int] = None


@dataclass
class OracleModel:
    lm: object                      # oracle.lm.Qwen2Oracle
    lm_head: torch.Tensor           # [V, H]
    head_w: dict
    head_layers: int
    ac_w: dict                      # acoustic tokenizer weights (encoder.* / decoder.*)
    sem_w: dict                     # semantic tokenizer weights (encoder.*)
    ac_conn: dict
    sem_conn: dict
    ratios: list                    # [8,5,5,4,2,2]
    enc_depths: list                # [3,3,3,3,3,3,8]
    dec_depths: list                # reversed
    sem_depths: list
    scaling: float
    bias: float
    fix_std: float = 0.5
    max_position_embeddings: int = 65536
    head_eps: float = 1e-5
    codec_eps: float = 1e-5
    t_cast_dtype: Optional[torch.dtype] = None


def cast_model(m: "OracleModel", dtype) -> "OracleModel":
    """the same model with every weight in `dtype` (what from_pretrained(torch_dtype=...) gives the reference: bf16 on a GPU,
    demo/inference_from_file.py:284-292); rotary frequencies stay fp32, as HF computes them"""
    import copy
    import dataclasses
    c = lambda d: {k: v.to(dtype) for k, v in d.items()}
    lm = copy.copy(m.lm)
    lm.w = c(m.lm.w)
    return dataclasses.replace(m, lm=lm, lm_head=m.lm_head.to(dtype), head_w=c(m.head_w), ac_w=c(m.ac_w), sem_w=c(m.sem_w),
                               ac_conn=c(m.ac_conn), sem_conn=c(m.sem_conn))


@dataclass
class Trace:
    pos_hidden: list = field(default_factory=list)
    neg_hidden: list = field(default_factory=list)
    latents: list = field(default_factory=list)
    semantic: list = field(default_factory=list)
    next_embeds: list = field(default_factory=list)
    tokens: list = field(default_factory=list)
    logits: list = field(default_factory=list)      # [B, n_valid] per step: the lm_head scores of the valid ids, in `valid` order
    audio: list = field(default_factory=list)       # [n, 3200] per diffusion step: the decoded frames of that step's rows


def process_speech_inputs(m: OracleModel, speech_tensors, speech_masks, prefill_noise):
    """:149-163.  prefill_noise = (randn[n_spk], randn[n_spk, T, 64]) replaces the
    two device-RNG draws of VibeVoiceTokenizerEncoderOutput.sample('gaussian')
    (modular_vibevoice_tokenizer.py:980-989)."""
    lat = codec.encoder_forward(m.ac_w, speech_tensors.unsqueeze(1), m.ratios, m.enc_depths,
                                state=None, eps=m.codec_eps)
    mean = lat.permute(0, 2, 1)                                  # [n_spk, T, 64]
    std = prefill_noise[0] * (m.fix_std / 0.8)
    x = mean + std[:, None, None] * prefill_noise[1]
    feats = (x + m.bias) * m.scaling
    connected = connector.connector_forward(m.ac_conn, feats)[speech_masks]
    return feats, connected


class NegativeRow:
    """One row of the reference's negative (CFG) branch, kept the way the reference keeps it: EVERY entry the negative pass ever
    appended (K/V per layer), an attention mask over them plus the slot of the token fed next (HF extends the mask by one after
    every forward, _update_model_kwargs_for_generation), and the row's correction counter (correct_cnt, :69).  The three things the
    loop does to it are restated array operation by array operation, because their net effect is NOT always "drop the spurious
    entry": the mask and the K/V shifts of the correction have different guards (:603 `start + 1 < seq_len - 1` with seq_len =
    cache length + 1, :613 `start + 1 < cache length - 1`), so for a row with exactly one valid entry the mask moves and the K/V
    does not -- the entry appended at this step STAYS and the older one is masked out (found by tools/fuzz_generate_vs_reference.py:
    a row that emits <speech_diffusion> at step 0 and something else at step 1 while another row diffuses).
    Position ids are mask-derived (cumsum - 1; what transformers 4.51.3's prepare_inputs_for_generation gives and oracle/refshim
    restores): the new token sits at position = number of valid entries before it; stored keys keep the rotation they were
    computed with."""

    def __init__(self, lm):
        self.lm = lm
        self.full = lm.new_cache()          # all entries ever appended, masked ones included
        self.mask = [1]                     # :379-386: the lone <speech_start> prompt token's slot
        self.cnt = 0                        # correct_cnt[b]

    def forward(self, e):
        """one negative pass of this row on input embedding e [1, H]; returns the final hidden state [H]"""
        c = self.full.length
        assert len(self.mask) == c + 1
        keep = [i for i in range(c) if self.mask[i]]
        tmp = self.lm.new_cache()
        if keep:
            idx = torch.tensor(keep, device=self.full.k[0].device)
            for l in range(len(tmp.k)):
                tmp.k[l], tmp.v[l] = self.full.k[l][:, idx], self.full.v[l][:, idx]
            tmp.length = len(keep)
        h = self.lm.forward(e, tmp)[-1]
        for l in range(len(tmp.k)):
            nk, nv = tmp.k[l][:, -1:], tmp.v[l][:, -1:]
            self.full.k[l] = nk if self.full.k[l] is None else torch.cat([self.full.k[l], nk], dim=1)
            self.full.v[l] = nv if self.full.v[l] is None else torch.cat([self.full.v[l], nv], dim=1)
        self.full.length = c + 1
        self.mask.append(1)
        return h

    def reset_on_speech_start(self):
        """:551-560: mask := 0 except the next token's slot; K/V[last] := K/V[0] (which that mask hides)"""
        self.mask = [0] * len(self.mask)
        self.mask[-1] = 1
        if self.full.length > 0:
            for l in range(len(self.full.k)):
                self.full.k[l][:, -1] = self.full.k[l][:, 0].clone()
                self.full.v[l][:, -1] = self.full.v[l][:, 0].clone()

    def correct_non_diffusing(self):
        """:594-624 for a live row that did not diffuse at a step where the negative pass ran"""
        c, s, n = self.full.length, self.cnt, len(self.mask)
        if s + 1 < n - 1:
            self.mask[s + 1:] = self.mask[s:-1]
        self.mask[s] = 0
        if s + 1 < c - 1:
            for l in range(len(self.full.k)):
                self.full.k[l][:, s + 1:] = self.full.k[l][:, s:-1].clone()
                self.full.v[l][:, s + 1:] = self.full.v[l][:, s:-1].clone()
        self.cnt += 1


def oracle_generate(m: OracleModel, tok: TokenIds, input_ids, attention_mask,
                    speech_tensors=None, speech_masks=None, speech_input_mask=None,
                    cfg_scale=1.3, num_steps=10, max_new_tokens=None, max_length_times=2,
                    noise_fn: Callable = None, prefill_noise=None,
                    forced_tokens: Optional[List[List[int]]] = None,
                    do_sample=False, trace: Optional[Trace] = None,
                    algorithm_type="dpmsolver++", sde_noise_fn: Callable = None,
                    teacher_embeds: Callable = None, refresh_negative: bool = True, batch_cache_quirk: bool = True):
    """Returns (sequences [B, L0+steps], speech_outputs list, reach_max_step_sample).
    algorithm_type "sde-dpmsolver++": the scheduler demo/gradio_demo.py:142-146 installs; sde_noise_fn(step, N, 2n) ->
    [N, 2n, 64], the variance noise scheduler.step() draws per solver step (dpm_solver.py:994-997).
    teacher_embeds(step) -> [B, H] or None: test hook (SURVEY 8d "teacher-forced per step") -- the NEXT positive pass consumes these
    embeddings instead of the loop's own, so two implementations are compared step by step on identical inputs.
    batch_cache_quirk=False: every row keeps its own tokenizer history whatever the other rows do (what a queue of independent
    requests computes: generate_continuous) instead of the lock-step batch's behaviour described at the decode step below.
    refresh_negative=False (:503-516; the reset of :550-565 and the forward of :576-588 are then skipped): the negative pass runs at
    EVERY step for every row, right after the token choice, on the embedding the positive pass consumed at this step (the lone
    <speech_start> prompt token at step 0, where inputs_embeds is still None, :395); it is never reset.  The correction of
    :590-624 still runs whenever some row diffuses: it shifts the valid part of a non-diffusing live row's negative cache right by
    one and masks the slot that frees up -- on the compact cache "the entry appended at this step is dropped again"."""
    B, L0 = input_ids.shape
    if max_new_tokens is None:
        max_new_tokens = m.max_position_embeddings - L0
    max_length = L0 + max_new_tokens
    init_len = attention_mask.sum(-1)
    max_steps = min(max_length - L0, int(max_length_times * L0))
    max_step_per_sample = torch.min(max_length - init_len, (max_length_times * init_len).long())
    valid = [tok.speech_start_id, tok.speech_end_id, tok.speech_diffusion_id, tok.eos_token_id]
    if tok.bos_token_id is not None:
        valid.append(tok.bos_token_id)

    finished = torch.zeros(B, dtype=torch.bool)
    reach_max = torch.zeros(B, dtype=torch.bool)
    pos_cache = [m.lm.new_cache() for _ in range(B)]
    neg = [NegativeRow(m.lm) for _ in range(B)]
    ac_state = [dict() for _ in range(B)]
    sem_state = [dict() for _ in range(B)]
    audio_chunks = [[] for _ in range(B)]
    seq = input_ids.clone()
    inputs_embeds = None                     # [B, H] embeds the NEXT positive pass consumes

    for step in range(max_steps):
        if finished.all():
            break
        if seq.shape[-1] >= max_length:
            reach_max[~finished] = True
            break
        # ---- positive LM pass (:466-485) ----
        hidden = []
        if step == 0:
            emb = m.lm.embed(input_ids)                       # [B, L0, H]
            if speech_tensors is not None:
                _, sp = process_speech_inputs(m, speech_tensors, speech_masks, prefill_noise)
                emb[speech_input_mask] = sp
            for b in range(B):
                e = emb[b][attention_mask[b].bool()]
                hidden.append(m.lm.forward(e, pos_cache[b])[-1])
        else:
            for b in range(B):
                hidden.append(m.lm.forward(inputs_embeds[b][None], pos_cache[b])[-1])
        hidden = torch.stack(hidden)                          # [B, H]
        consumed = inputs_embeds
        # ---- token selection (:488-501) ----
        logits = F.linear(hidden, m.lm_head).float()
        mask = torch.full_like(logits, float("-inf"))
        mask[:, valid] = 0
        scores = logits + mask
        if forced_tokens is not None:
            nxt = torch.tensor([forced_tokens[b][step] if step < len(forced_tokens[b])
                                else tok.eos_token_id for b in range(B)])
        elif do_sample:
            nxt = torch.multinomial(torch.softmax(scores, -1), 1).squeeze(1)
        else:
            nxt = torch.argmax(scores, dim=-1)
        nxt[finished] = tok.eos_token_id
        seq = torch.cat([seq, nxt[:, None]], dim=-1)
        if trace is not None:
            trace.pos_hidden.append(hidden.clone())
            trace.tokens.append(nxt.clone())
            trace.logits.append(logits[:, valid].clone())
        # ---- bookkeeping (:518-539) ----
        finished = finished | (nxt == tok.eos_token_id)
        hit = (step >= max_step_per_sample) & ~finished
        finished = finished | hit
        reach_max = reach_max | hit
        # ---- <speech_end>: zero both conv caches (:542-546) ----
        for b in (nxt == tok.speech_end_id).nonzero().flatten().tolist():
            codec.zero_state(ac_state[b])
            codec.zero_state(sem_state[b])
        def neg_input(b):                  # :505-508 / :578-581: the embedding the positive pass consumed, or the lone prompt token
            return m.lm.embed(torch.tensor([tok.speech_start_id])) if consumed is None else consumed[b][None]
        neg_all = None
        if not refresh_negative:
            # ---- :503-516: negative pass of every row at every step (finished rows too in the reference; theirs is never read) ----
            neg_all = {b: neg[b].forward(neg_input(b)) for b in range(B)}
        # ---- <speech_start>: reset the negative branch (:549-565) ----
        if refresh_negative:
            for b in (~finished & (nxt == tok.speech_start_id)).nonzero().flatten().tolist():
                neg[b].reset_on_speech_start()
        next_embeds = m.lm.embed(nxt)                          # [B, H]
        diff = (~finished & (nxt == tok.speech_diffusion_id)).nonzero().flatten().tolist()
        if diff:
            n = len(diff)
            if refresh_negative:
                # :576-588: the negative pass runs for EVERY row whenever some row diffuses
                neg_all = {b: neg[b].forward(neg_input(b)) for b in range(B) if not finished[b] or b in diff}
            neg_hidden = torch.stack([neg_all[b] for b in diff])
            # :590-624: the rows that are live and did not diffuse are "corrected"
            for b in (~finished & (nxt != tok.speech_diffusion_id)).nonzero().flatten().tolist():
                neg[b].correct_non_diffusing()
            pos_cond = hidden[diff]
            noise = noise_fn(step, 2 * n)
            sn = sde_noise_fn(step, num_steps, 2 * n) if algorithm_type == "sde-dpmsolver++" else None
            lat = dpm.sample_speech_tokens(
                lambda x, t, c: head.head_forward(m.head_w, x, t, c, m.head_layers, m.head_eps),
                pos_cond, neg_hidden, cfg_scale, num_steps, noise, m.t_cast_dtype,
                algorithm_type=algorithm_type, step_noise=sn)
            scaled = lat / m.scaling - m.bias
            # VibeVoiceTokenizerStreamingCache.get (modular_vibevoice_tokenizer.py:198-207) returns None -- "no history", i.e. zero
            # left context for EVERY row of the call -- as soon as ONE of the requested rows has no entry yet.  A row that diffuses
            # for the first time therefore costs the rows decoded in the same call their conv history for that frame (both
            # tokenizers; the new states are stored for all of them afterwards).  Every processor-built prompt ends in
            # <speech_start> and all rows take their first frame together at step 0, where "no history" is the right answer, so the
            # quirk never fires there; under a forced plan that starts a row late it does (pinned by generate_late_start_b2*.npz).
            if batch_cache_quirk and any(not ac_state[b] for b in diff) and any(ac_state[b] for b in diff):
                for b in diff:
                    codec.zero_state(ac_state[b])
                    codec.zero_state(sem_state[b])
            sem_list = []
            for j, b in enumerate(diff):
                chunk = codec.decoder_forward(m.ac_w, scaled[j][None, :, None], m.ratios, m.dec_depths,
                                              state=ac_state[b], eps=m.codec_eps)      # [1,1,3200]
                audio_chunks[b].append(chunk[0])
                sem = codec.encoder_forward(m.sem_w, chunk, m.ratios, m.sem_depths,
                                            state=sem_state[b], eps=m.codec_eps)        # [1,128,1]
                sem_list.append(sem[0, :, 0])
            sem_feat = torch.stack(sem_list)
            emb = connector.connector_forward(m.ac_conn, lat) + connector.connector_forward(m.sem_conn, sem_feat)
            next_embeds[diff] = emb
            if trace is not None:
                trace.neg_hidden.append(neg_hidden.clone())
                trace.latents.append(lat.clone())
                trace.semantic.append(sem_feat.clone())
                trace.audio.append(torch.stack([audio_chunks[b][-1].reshape(-1) for b in diff]))
        if trace is not None:
            trace.next_embeds.append(next_embeds.clone())
        inputs_embeds = next_embeds
        if teacher_embeds is not None:
            te = teacher_embeds(step)
            if te is not None:
                inputs_embeds = te.to(next_embeds)

    outs = [torch.cat(c, dim=-1) if c else None for c in audio_chunks]
    return seq, outs, reach_max
