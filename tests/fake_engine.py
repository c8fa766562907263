"""CPU stand-in for vibevoice_amd.engine.Engine, built from the ORACLE's arithmetic (tests only).

Purpose: run the product's host orchestration -- vibevoice_amd/modeling.py::generate, unmodified -- on a machine
without a GPU and compare it with the goldens recorded from the reference's own generate().  Every numeric stage is
delegated to oracle/ (itself pinned to the reference); what is under test is the host logic: row tables, negative-branch
bookkeeping, speculation, codec state resets, EOS / max-length handling, output assembly.
"""
import contextlib
import types

import torch
import torch.nn.functional as F

from oracle import codec, connector, dpm, head


class _Event:
    def record(self, stream=None):
        pass

    def synchronize(self):
        pass


class _Stream:
    def wait_event(self, ev):
        pass


@contextlib.contextmanager
def cpu_cuda_shims(monkeypatch):
    """modeling.py talks to torch.cuda streams / events / pinned memory; give it inert CPU versions."""
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: _Event())
    monkeypatch.setattr(torch.cuda, "Stream", lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(torch.Tensor, "pin_memory", lambda self, *a, **k: self)
    yield


class FakeEngine:
    def __init__(self, om, n_slots=2, max_rows=16, max_ctx=512):
        self.om = om
        self.device = torch.device("cpu")
        self.stream = None
        self.max_ctx = max_ctx
        self.cfg = types.SimpleNamespace(lm_hidden=om.lm_head.shape[1], latent_dim=64, hop=3200, sem_dim=128,
                                         n_slots=n_slots, max_rows=max_rows, lm_vocab=om.lm_head.shape[0])
        self.caches = {}
        self.ac_state = [dict() for _ in range(n_slots)]
        self.sem_state = [dict() for _ in range(n_slots)]
        self.valid = None
        self.n_steps = 10
        self.calls = {"lm_rows": 0, "samples": 0, "spec_wasted": 0}

    # ---- plumbing ----
    def fork(self, **runtime):
        """the CPU image of Engine.fork: a second engine over the same (read-only) weights with its own state"""
        f = FakeEngine(self.om, n_slots=runtime.get("n_slots", self.cfg.n_slots), max_rows=runtime.get("max_rows", self.cfg.max_rows),
                       max_ctx=runtime.get("max_ctx", self.max_ctx))
        f.shared_from = self
        return f

    def close(self):
        self.closed = True

    def new(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def sync(self):
        pass

    def set_valid_tokens(self, valid):
        self.valid = list(valid)

    def set_num_steps(self, n, t_cast_bf16=False, algorithm_type="dpmsolver++"):
        self.n_steps = n
        self.algorithm_type = algorithm_type

    def set_speech_factors(self, scaling, bias):
        pass

    # ---- LM ----
    def embed(self, ids, out):
        out[:len(ids)] = self.om.lm.embed(torch.tensor(ids, dtype=torch.long))

    def lm_forward(self, rows, x_in, hidden):
        for i, (cache, pos) in enumerate(rows):
            c = self.caches.setdefault(cache, self.om.lm.new_cache())
            if pos < c.length:
                c.truncate(pos)                      # a row table may rewind a cache (negative-branch reset)
            assert pos == c.length, (cache, pos, c.length)
            hidden[i] = self.om.lm.forward(x_in[i][None], c)[-1]
            self.calls["lm_rows"] += 1

    def kv_move(self, cache, src_pos, dst_pos):
        c = self.caches[cache]
        for l in range(len(c.k)):
            c.k[l][:, dst_pos] = c.k[l][:, src_pos].clone()
            c.v[l][:, dst_pos] = c.v[l][:, src_pos].clone()

    def lm_logits(self, n, hidden, out):
        # the C ABI's layout: a dense [n][n_valid] block at the start of `out` (include/vvhip.h, vv_lm_logits)
        nv = len(self.valid)
        out.reshape(-1)[:n * nv].copy_(F.linear(hidden[:n], self.om.lm_head)[:, self.valid].reshape(-1))

    def lm_logits_full(self, n, hidden, out):
        # [n][lm_vocab] over the whole table (include/vvhip.h, vv_lm_logits_full)
        V = self.om.lm_head.shape[0]
        out.reshape(-1)[:n * V].copy_(F.linear(hidden[:n].float(), self.om.lm_head).reshape(-1))

    # ---- diffusion ----
    def diffusion_sample(self, n, cond, noise, cfg_scale, latent_out, step_noise=None):
        om = self.om
        nz = torch.cat([noise[:n], noise[:n]])
        algo = getattr(self, "algorithm_type", "dpmsolver++")
        assert (step_noise is not None) == (algo == "sde-dpmsolver++")       # the C ABI refuses the mismatch too
        sn = None if step_noise is None else torch.cat([step_noise[:, :n], step_noise[:, :n]], dim=1)    # [N, 2n, 64]
        lat = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(om.head_w, x, t, c, om.head_layers, om.head_eps),
                                       cond[:n].clone(), cond[n:2 * n].clone(), cfg_scale, self.n_steps, nz, om.t_cast_dtype,
                                       algorithm_type=algo, step_noise=sn)
        latent_out[:n] = lat
        self.calls["samples"] += 1

    # ---- tokenizers / connectors ----
    def codec_decode(self, slot, latent, audio_out, apply_speech_factors=True, stream=None):
        om = self.om
        x = latent[0] / om.scaling - om.bias if apply_speech_factors else latent[0]
        chunk = codec.decoder_forward(om.ac_w, x[None, :, None], om.ratios, om.dec_depths, state=self.ac_state[slot], eps=om.codec_eps)
        audio_out.copy_(chunk[0, 0])

    def semantic_encode(self, slot, audio, sem_out, stream=None):
        om = self.om
        sem = codec.encoder_forward(om.sem_w, audio[None, None, :], om.ratios, om.sem_depths, state=self.sem_state[slot], eps=om.codec_eps)
        sem_out.copy_(sem[0, :, 0])

    def codec_chain_batch(self, slots, latent, audio_out, sem_out=None, apply_speech_factors=True):
        for j, sl in enumerate(slots):
            self.codec_decode(sl, latent[j:j + 1], audio_out[j], apply_speech_factors)
            if sem_out is not None:
                self.semantic_encode(sl, audio_out[j], sem_out[j])

    def acoustic_encode(self, frames, wav, mean_out, valid_samples=None):
        om = self.om
        if valid_samples is not None:
            wav = wav[:valid_samples]          # the oracle encoder pads per conv layer, as the reference does
        lat = codec.encoder_forward(om.ac_w, wav[None, None, :], om.ratios, om.enc_depths, state=None, eps=om.codec_eps)
        mean_out.copy_(lat[0].t())

    def connect(self, n, latent, sem, out):
        om = self.om
        e = connector.connector_forward(om.ac_conn, latent[:n])
        if sem is not None:
            e = e + connector.connector_forward(om.sem_conn, sem[:n])
        out[:n] = e

    def codec_reset(self, slot):
        codec.zero_state(self.ac_state[slot])
        codec.zero_state(self.sem_state[slot])


class FakeStreamingEngine:
    """Same idea for the Streaming-0.5B host loop (vibevoice_amd/modeling_streaming.py): the split LM is two oracle
    stacks (text LM = layers [0, n_lm) without final norm, TTS LM = the rest), caches 0 / 1 / 2 = lm / tts / neg tts."""

    def __init__(self, om, n_lm, n_tts, max_ctx=512):
        self.om, self.n_lm, self.n_tts = om, n_lm, n_tts
        self.device = torch.device("cpu")
        self.stream = None
        self.max_ctx = max_ctx
        self.cfg = types.SimpleNamespace(lm_hidden=om.tts_types.shape[1], latent_dim=64, hop=3200, sem_dim=0, n_slots=1, max_rows=16,
                                         lm_layers=n_lm + n_tts, tts_layers=n_tts)
        self.caches = {0: om.lm.new_cache(), 1: om.tts_lm.new_cache(), 2: om.tts_lm.new_cache()}
        self.state = {}
        self.n_steps = 5

    def new(self, *shape):
        return torch.zeros(*shape, dtype=torch.float32)

    def sync(self):
        pass

    def set_num_steps(self, n, t_cast_bf16=False):
        self.n_steps = n

    def set_speech_factors(self, scaling, bias):
        pass

    def codec_reset(self, slot):
        codec.zero_state(self.state)

    def kv_import(self, cache, layer, k, v):
        c = self.caches[cache]
        li = layer if cache == 0 else layer - self.n_lm
        c.k[li] = k.clone().float()
        c.v[li] = v.clone().float()
        c.length = k.shape[1]

    def embed(self, ids, out):
        out[:len(ids)] = self.om.lm.embed(torch.tensor(ids, dtype=torch.long))

    def lm_forward_range(self, rows, x_in, hidden, l0, l1, final_norm):
        stack = self.om.lm if l0 == 0 else self.om.tts_lm
        for i, (cache, pos) in enumerate(rows):
            c = self.caches[cache]
            assert pos == c.length, (cache, pos, c.length)
            hidden[i] = stack.forward(x_in[i][None], c, final_norm=bool(final_norm))[-1]

    def add_type_embedding(self, n, x, type_id, out):
        out[:n] = x[:n] + self.om.tts_types[type_id]

    def eos_logit(self, n, hidden, out):
        from oracle.generate_streaming import eos_logit
        out[:n] = eos_logit(self.om.eos, hidden[:n]).reshape(-1)

    def diffusion_sample(self, n, cond, noise, cfg_scale, latent_out):
        om = self.om
        nz = torch.cat([noise[:n], noise[:n]])
        lat = dpm.sample_speech_tokens(lambda x, t, c: head.head_forward(om.head_w, x, t, c, om.head_layers, om.head_eps),
                                       cond[:n].clone(), cond[n:2 * n].clone(), cfg_scale, self.n_steps, nz, None)
        latent_out[:n] = lat

    def codec_decode(self, slot, latent, audio_out, apply_speech_factors=True, stream=None):
        om = self.om
        x = latent[0] / om.scaling - om.bias
        chunk = codec.decoder_forward(om.ac_w, x[None, :, None], om.ratios, om.dec_depths, state=self.state, eps=om.codec_eps)
        audio_out.copy_(chunk[0, 0])

    def connect(self, n, latent, sem, out):
        out[:n] = connector.connector_forward(self.om.ac_conn, latent[:n])


class LoadableFakeEngine(FakeEngine):
    """`Engine(ecfg, device)` stand-in for the construction path (from_state_dict / from_pretrained / WeightHandle /
    load_lora_assets): collects vv_upload()s under the engine's parameter names and builds the oracle model from them on
    first use, so the class can be driven from a checkpoint directory on a machine without a GPU."""

    class _Any(dict):
        def __contains__(self, k):
            return True

    def __init__(self, ecfg, device=None):
        self.ecfg = ecfg
        self._w = {}
        self._om = None
        self._factors = (1.0, 0.0)
        self.device = torch.device("cpu")
        self.stream = None
        self.max_ctx = (ecfg.max_ctx + 127) // 128 * 128
        self.cfg = types.SimpleNamespace(lm_hidden=ecfg.lm_hidden, latent_dim=ecfg.latent_dim, hop=ecfg.hop, sem_dim=ecfg.sem_dim,
                                         n_slots=ecfg.n_slots, max_rows=ecfg.max_rows, lm_vocab=ecfg.lm_vocab)
        self.caches = {}
        self.ac_state = [dict() for _ in range(ecfg.n_slots)]
        self.sem_state = [dict() for _ in range(ecfg.n_slots)]
        self.valid = None
        self.n_steps = 10
        self.calls = {"lm_rows": 0, "samples": 0, "spec_wasted": 0}
        self.uploads = []
        self._frozen = None

    def expected_weights(self):
        # while the checkpoint streams in, every mapped key is taken; once the load has been checked (missing_weights), the
        # parameter set is the one the checkpoint defined -- as the real engine's registry is fixed by its config
        return self._frozen if self._frozen is not None else self._Any()

    def missing_weights(self):
        self._frozen = {k: int(v.numel()) for k, v in self._w.items()}
        return []

    def upload(self, name, t):
        self._w[name] = t.detach().to(torch.float32).clone()
        self.uploads.append(name)
        self._om = None                       # the next op sees the new snapshot

    def set_speech_factors(self, scaling, bias):
        self._factors = (float(scaling), float(bias))
        if self._om is not None:
            self._om.scaling, self._om.bias = self._factors

    @property
    def om(self):
        if self._om is None:
            from oracle import generate as ogen
            from oracle import lm as olm
            c = self.ecfg

            def sub(p, keep=""):
                return {keep + k[len(p):]: v for k, v in self._w.items() if k.startswith(p)}
            lm_w = sub("lm.")
            lm_w.pop("rope.inv_freq", None)
            lm = olm.Qwen2Oracle(lm_w, c.lm_layers, c.lm_heads, c.lm_kv_heads, c.lm_head_dim, c.rope_theta, c.lm_eps)
            depths = list(c.enc_depths)
            self._om = ogen.OracleModel(
                lm=lm, lm_head=self._w.get("lm_head.weight", lm_w["embed_tokens.weight"]), head_w=sub("head."), head_layers=c.head_layers,
                ac_w={**sub("dec.", "decoder."), **sub("aenc.", "encoder.")}, sem_w=sub("senc.", "encoder."),
                ac_conn=sub("ac_conn."), sem_conn=sub("sem_conn."), ratios=list(c.ratios), enc_depths=depths,
                dec_depths=list(reversed(depths)), sem_depths=depths, scaling=self._factors[0], bias=self._factors[1],
                max_position_embeddings=c.max_ctx, head_eps=c.head_eps, codec_eps=c.codec_eps)
        return self._om

    @om.setter
    def om(self, v):
        self._om = v


class LoadableFakeStreamingEngine(FakeStreamingEngine):
    """`Engine(ecfg, device)` stand-in for the streaming class's construction path (from_state_dict / from_pretrained /
    WeightHandle): collects vv_upload()s under the engine's parameter names and builds the split oracle model from them
    on first use (text LM = layers [0, n_lm) without final norm, TTS LM = the rest + `lm.norm`)."""

    def __init__(self, ecfg, device=None):
        self.ecfg = ecfg
        self._w = {}
        self._om = None
        self._factors = (1.0, 0.0)
        self._frozen = None
        self.uploads = []
        self.n_tts = ecfg.tts_layers
        self.n_lm = ecfg.lm_layers - ecfg.tts_layers
        self.device = torch.device("cpu")
        self.stream = None
        self.max_ctx = (ecfg.max_ctx + 127) // 128 * 128
        self.cfg = types.SimpleNamespace(lm_hidden=ecfg.lm_hidden, latent_dim=ecfg.latent_dim, hop=ecfg.hop, sem_dim=0, n_slots=ecfg.n_slots,
                                         max_rows=ecfg.max_rows, lm_layers=ecfg.lm_layers, tts_layers=ecfg.tts_layers)
        self._caches = None
        self.state = {}
        self.n_steps = 5

    def expected_weights(self):
        return self._frozen if self._frozen is not None else LoadableFakeEngine._Any()

    def missing_weights(self):
        self._frozen = {k: int(v.numel()) for k, v in self._w.items()}
        return []

    def upload(self, name, t):
        self._w[name] = t.detach().to(torch.float32).clone()
        self.uploads.append(name)
        self._om = None

    def set_speech_factors(self, scaling, bias):
        self._factors = (float(scaling), float(bias))
        if self._om is not None:
            self._om.scaling, self._om.bias = self._factors

    @property
    def om(self):
        if self._om is None:
            from oracle import generate_streaming as ogs
            from oracle import lm as olm
            c = self.ecfg
            sub = lambda p: {k[len(p):]: v for k, v in self._w.items() if k.startswith(p)}
            w = sub("lm.")
            lm_w = {k: v for k, v in w.items() if k.startswith("embed") or any(k.startswith(f"layers.{i}.") for i in range(self.n_lm))}
            tts_w = {"norm.weight": w["norm.weight"], "embed_tokens.weight": w["embed_tokens.weight"]}
            for j in range(self.n_tts):
                pre = f"layers.{self.n_lm + j}."
                for k, v in w.items():
                    if k.startswith(pre):
                        tts_w[f"layers.{j}." + k[len(pre):]] = v
            mk = lambda ww, L: olm.Qwen2Oracle(ww, L, c.lm_heads, c.lm_kv_heads, c.lm_head_dim, c.rope_theta, c.lm_eps, kv_round_bf16=False)
            depths = list(c.enc_depths)
            self._om = ogs.StreamingOracleModel(
                lm=mk(lm_w, self.n_lm), tts_lm=mk(tts_w, self.n_tts), tts_types=self._w["tts_input_types.weight"], eos=sub("eos."),
                head_w=sub("head."), head_layers=c.head_layers, ac_w={"decoder." + k: v for k, v in sub("dec.").items()},
                ac_conn=sub("ac_conn."), ratios=list(c.ratios), dec_depths=list(reversed(depths)),
                scaling=self._factors[0], bias=self._factors[1])
        return self._om

    @om.setter
    def om(self, v):
        self._om = v

    @property
    def caches(self):
        if self._caches is None:
            om = self.om
            self._caches = {0: om.lm.new_cache(), 1: om.tts_lm.new_cache(), 2: om.tts_lm.new_cache()}
        return self._caches

    @caches.setter
    def caches(self, v):
        self._caches = v
