"""Builds a small engine + the matching oracle weight dicts from tests/synth.py."""
from dataclasses import dataclass

import torch

import synth
from oracle import generate as ogen
from oracle import lm as olm


@dataclass
class Small:
    eng: object
    lmcfg: synth.LMCfg
    hc: synth.HeadCfg
    cc: synth.CodecCfg
    sc: synth.CodecCfg
    lm_w: dict
    lm_head: torch.Tensor
    head_w: dict
    ac_w: dict
    sem_w: dict
    ac_conn: dict
    sem_conn: dict
    scaling: float = 0.2
    bias: float = -0.05

    def oracle_lm(self, kv_round_bf16=True):
        c = self.lmcfg
        return olm.Qwen2Oracle(self.lm_w, c.layers, c.heads, c.kv_heads, c.head_dim, c.theta, c.eps,
                               kv_round_bf16=kv_round_bf16)

    def oracle_model(self, kv_round_bf16=True):
        return ogen.OracleModel(
            lm=self.oracle_lm(kv_round_bf16), lm_head=self.lm_head, head_w=self.head_w, head_layers=self.hc.layers,
            ac_w=self.ac_w, sem_w=self.sem_w, ac_conn=self.ac_conn, sem_conn=self.sem_conn,
            ratios=self.cc.ratios, enc_depths=self.cc.enc_depths, dec_depths=self.cc.dec_depths,
            sem_depths=self.sc.enc_depths, scaling=self.scaling, bias=self.bias,
            max_position_embeddings=self.lmcfg.max_pos, head_eps=self.hc.eps, codec_eps=self.cc.eps)


def build_small(lmcfg=None, xsplit=3, use_graph=False, n_slots=2, max_ctx=512, tied=False, max_rows=16, head_layers=2, head_ffn_ratio=3.0):
    from vibevoice_amd.engine import Engine, EngineConfig
    lmcfg = lmcfg or synth.LMCfg()
    H = lmcfg.hidden
    hc = synth.HeadCfg(hidden=H, layers=head_layers, ffn_ratio=head_ffn_ratio)
    cc = synth.CodecCfg()
    sc = synth.CodecCfg(vae_dim=128)
    lm_w = synth.lm_weights(lmcfg)
    lm_head = lm_w["embed_tokens.weight"] if tied else synth.lm_head_weight(lmcfg)
    head_w = synth.head_weights(hc)
    ac_w = {**synth.encoder_weights(cc, 2), **synth.decoder_weights(cc, 3)}
    sem_w = synth.encoder_weights(sc, 7)
    ac_conn = synth.connector_weights(64, H, 4)
    sem_conn = synth.connector_weights(128, H, 8)
    ecfg = EngineConfig(lm_hidden=H, lm_layers=lmcfg.layers, lm_heads=lmcfg.heads, lm_kv_heads=lmcfg.kv_heads,
                        lm_inter=lmcfg.inter, lm_vocab=lmcfg.vocab, lm_eps=lmcfg.eps, rope_theta=lmcfg.theta,
                        head_layers=hc.layers, head_ffn_ratio=hc.ffn_ratio, head_eps=hc.eps,
                        n_filters=cc.n_filters, ratios=cc.ratios, enc_depths=cc.enc_depths, sem_dim=128,
                        codec_eps=cc.eps, n_slots=n_slots, max_ctx=max_ctx, xsplit=xsplit, use_graph=use_graph, max_rows=max_rows,
                        enc_frames=2)
    eng = Engine(ecfg)
    sd = {}
    sd.update({"lm." + k: v for k, v in lm_w.items()})
    if not tied:
        sd["lm_head.weight"] = lm_head
    sd.update({"head." + k: v for k, v in head_w.items()})
    for k, v in ac_w.items():
        sd[("dec." + k[len("decoder."):]) if k.startswith("decoder.") else ("aenc." + k[len("encoder."):])] = v
    sd.update({"senc." + k[len("encoder."):]: v for k, v in sem_w.items()})
    sd.update({"ac_conn." + k: v for k, v in ac_conn.items()})
    sd.update({"sem_conn." + k: v for k, v in sem_conn.items()})
    eng.load_state_dict(sd, mapped=True, strict=True)
    s = Small(eng, lmcfg, hc, cc, sc, lm_w, lm_head, head_w, ac_w, sem_w, ac_conn, sem_conn)
    eng.set_speech_factors(s.scaling, s.bias)
    return s


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def max_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
