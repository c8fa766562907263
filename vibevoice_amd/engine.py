"""Thin Python wrapper over the libvvhip.so C ABI (include/vvhip.h).

PyTorch is used here only for device memory (tensors whose data_ptr() is handed
to the engine) and for the HIP stream; every arithmetic op of the hot path runs
inside libvvhip.so.  There is no fallback path: constructing an Engine without
the shared library or without a GPU raises.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from . import schedule as _schedule


@dataclass
class EngineConfig:
    # Qwen2 decoder
    lm_hidden: int
    lm_layers: int
    lm_heads: int
    lm_kv_heads: int
    lm_inter: int
    lm_vocab: int
    lm_head_dim: Optional[int] = None
    lm_eps: float = 1e-6
    rope_theta: float = 1e6
    # diffusion head
    head_layers: int = 4
    head_ffn_ratio: float = 3.0
    latent_dim: int = 64
    head_eps: float = 1e-5
    # tokenizers
    n_filters: int = 32
    ratios: Sequence[int] = (8, 5, 5, 4, 2, 2)
    enc_depths: Sequence[int] = (3, 3, 3, 3, 3, 3, 8)
    sem_dim: int = 128
    has_acoustic_encoder: bool = True
    codec_eps: float = 1e-5
    # runtime
    n_slots: int = 1
    max_ctx: int = 4096
    max_rows: int = 16
    xsplit: int = 2
    attn_splits: int = 128      # upper bound on flash-decoding splits; a launch uses one per 512 positions of its longest row
    enc_frames: int = 75         # voice-prompt frames per acoustic-encoder pass: one pass per 10-s speaker prompt (any pass size gives the
                                 # same result, tests/test_gpu_shipped.py; 0.042 s at 5 frames per pass -> 0.027 s at 75 for two speakers)
    use_graph: bool = True
    tts_layers: int = 0          # Streaming-0.5B: the last tts_layers of lm_layers form the TTS LM

    def __post_init__(self):
        if self.lm_head_dim is None:
            self.lm_head_dim = self.lm_hidden // self.lm_heads

    @property
    def head_ffn(self):
        return int(self.lm_hidden * self.head_ffn_ratio)

    @property
    def hop(self):
        return int(np.prod(self.ratios))


# reference state_dict prefix -> engine parameter prefix
PREFIX_MAP = (
    ("model.language_model.", "lm."),
    ("model.prediction_head.", "head."),
    ("model.acoustic_tokenizer.decoder.", "dec."),
    ("model.acoustic_tokenizer.encoder.", "aenc."),
    ("model.semantic_tokenizer.encoder.", "senc."),
    ("model.acoustic_connector.", "ac_conn."),
    ("model.semantic_connector.", "sem_conn."),
    ("lm_head.", "lm_head."),
)


def map_param_name(ref_key: str) -> Optional[str]:
    for a, b in PREFIX_MAP:
        if ref_key.startswith(a):
            return b + ref_key[len(a):]
    return None


class EngineError(RuntimeError):
    pass


class Engine:
    def __init__(self, cfg: EngineConfig, device: Optional[torch.device] = None, share_from: Optional["Engine"] = None):
        """share_from: another Engine of the same model on the same device whose weights are fully uploaded -- this engine then reads
        THOSE weights (vv_create_shared) and owns only its runtime state (KV caches, activations, tokenizer state, graphs) and its
        stream; uploads / LoRA merges go through the owner.  See Engine.fork()."""
        if not torch.cuda.is_available():
            raise EngineError("vibevoice_amd.Engine needs an AMD GPU (torch.cuda.is_available() is False); "
                              "there is no CPU fallback")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.index is None:                 # device_map="cuda" (demo/inference_from_file.py:303): the current device
            self.device = torch.device("cuda", torch.cuda.current_device())
        torch.cuda.set_device(self.device)
        c = _lib.VVConfig()
        for f in ("lm_hidden", "lm_layers", "lm_heads", "lm_kv_heads", "lm_head_dim", "lm_inter", "lm_vocab",
                  "lm_eps", "head_layers", "latent_dim", "head_eps", "n_filters", "sem_dim", "codec_eps",
                  "n_slots", "max_ctx", "max_rows", "xsplit", "attn_splits", "enc_frames"):
            setattr(c, f, getattr(cfg, f))
        c.head_ffn = cfg.head_ffn
        c.n_ratios = len(cfg.ratios)
        for i, r in enumerate(cfg.ratios):
            c.ratios[i] = int(r)
        c.n_stages = len(cfg.enc_depths)
        for i, d in enumerate(cfg.enc_depths):
            c.enc_depths[i] = int(d)
        c.has_acoustic_encoder = int(cfg.has_acoustic_encoder)
        c.use_graph = int(cfg.use_graph)
        c.tts_layers = int(cfg.tts_layers)
        self._ctx = C.c_void_p()
        # a dedicated non-default stream: hipGraph capture is illegal on the null stream
        self.stream = torch.cuda.Stream(device=self.device)
        self.shared_from = share_from
        if share_from is not None:
            if share_from.device != self.device:
                raise EngineError("a shared engine lives on its owner's device")
            rc = self.lib.vv_create_shared(C.byref(c), share_from._ctx, C.byref(self._ctx))
            if rc != 0:
                msg = self._err()
                if self._ctx:
                    self.lib.vv_destroy(self._ctx)
                    self._ctx = C.c_void_p()
                raise EngineError("vv_create_shared failed: " + msg)
            self.max_ctx = (cfg.max_ctx + 127) // 128 * 128
            self._n_steps = None
            self._loaded = set(share_from._loaded)
            return
        rc = self.lib.vv_create(C.byref(c), C.byref(self._ctx))
        if rc != 0:
            raise EngineError("vv_create failed: " + self._err())
        self.max_ctx = (cfg.max_ctx + 127) // 128 * 128
        self._n_steps = None
        self._loaded = set()
        # HF Qwen2RotaryEmbedding inv_freq, computed exactly as transformers does
        d = cfg.lm_head_dim
        inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, d, 2, dtype=torch.int64).float() / d))
        self.upload("lm.rope.inv_freq", inv_freq)

    def fork(self, **runtime) -> "Engine":
        """A second engine over this engine's weights (one copy in HBM) with its own runtime state and stream; runtime: n_slots, max_ctx,
        max_rows, attn_splits, use_graph overrides.  Two engines driven from two host threads interleave two independent utterance
        batches on the GPU: each chain's launch boundaries are filled by the other's kernels."""
        import dataclasses
        owner = self.shared_from or self
        return Engine(dataclasses.replace(self.cfg, **runtime), self.device, share_from=owner)

    # ------------------------------------------------------------------ plumbing
    def _err(self):
        s = self.lib.vv_last_error(self._ctx)
        return s.decode() if s else "?"

    def _chk(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed: {self._err()}")

    @property
    def _s(self):
        return C.c_void_p(self.stream.cuda_stream)

    def close(self):
        if self._ctx:
            self.lib.vv_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        """wait for the engine stream, then surface asynchronous device-side errors (vv_check: the prefill GEMM's K-split
        hand-off reports a lost producer through a host word instead of hanging)"""
        self.stream.synchronize()
        if self._ctx:
            self._chk(self.lib.vv_check(self._ctx, self._s), "vv_check")
            self._warn_capture_fallbacks()
            n = int(self.lib.vv_stat(self._ctx, 5))
            if n and os.environ.get("VVHIP_ALLOW_FOREIGN_NODES") != "1":      # the escape: A/B runs against library builds from before round 6
                # an invariant of the library, checked where every generate() ends: a memset node of a replayed hipGraph was seen to
                # fill with stale words on this runtime, so captured sequences hold kernel launches only
                raise RuntimeError(f"vibevoice_amd: {n} memset / memcpy node(s) inside this engine's captured hipGraphs -- a copy or fill "
                                   "was enqueued with hipMemsetAsync / hipMemcpyAsync inside a captured sequence; use the library's copy "
                                   "and fill kernels (csrc/misc.hip: vv_copy_launch, vv_zero_launch)")

    def _warn_capture_fallbacks(self):
        """One warning per engine the first time a stream capture did not close and its work ran eagerly instead (vv_stat(ctx, 4) > 0):
        the step is still correct, but that launch sequence will be enqueued kernel by kernel from now on -- a serving process that
        lost its graphs should hear about it.  The known cause is a device-wide synchronize (torch.cuda.synchronize(),
        hipDeviceSynchronize) in ANOTHER host thread while this engine was capturing: synchronize streams or events instead."""
        if getattr(self, "_fallback_warned", False) or not self._ctx:
            return
        n = int(self.lib.vv_stat(self._ctx, 4))
        if n > 0:
            self._fallback_warned = True
            import warnings
            warnings.warn(f"vibevoice_amd: {n} hipGraph capture(s) of this engine did not close and ran eagerly instead -- those launch sequences "
                          "stay un-graphed (slower steps, same results).  Usual cause: another host thread called a device-wide synchronize "
                          "(torch.cuda.synchronize()) during the capture; synchronize streams or events in a process that generates.",
                          RuntimeWarning, stacklevel=3)

    def new(self, *shape, dtype=torch.float32):
        """zero tensor whose fill is ordered on the engine stream"""
        with torch.cuda.stream(self.stream):
            return torch.zeros(*shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ parameters
    def expected_weights(self) -> Dict[str, int]:
        out = {}
        buf = C.create_string_buffer(256)
        n = C.c_int64()
        ld = C.c_int()
        for i in range(self.lib.vv_num_weights(self._ctx)):
            self._chk(self.lib.vv_weight_info(self._ctx, i, buf, 256, C.byref(n), C.byref(ld)), "vv_weight_info")
            out[buf.value.decode()] = n.value
        return out

    def missing_weights(self) -> List[str]:
        miss = []
        buf = C.create_string_buffer(256)
        n = C.c_int64()
        ld = C.c_int()
        for i in range(self.lib.vv_num_weights(self._ctx)):
            self.lib.vv_weight_info(self._ctx, i, buf, 256, C.byref(n), C.byref(ld))
            if not ld.value:
                miss.append(buf.value.decode())
        return miss

    def upload(self, name: str, t: torch.Tensor):
        """t: fp32 or bf16 tensor on any device (host tensors are staged)."""
        t = t.detach()
        if t.dtype not in (torch.float32, torch.bfloat16):
            t = t.to(torch.float32)
        t = t.contiguous()
        # the tensor must exist before the library reads it (vv_upload works on the null stream).  The producing stream only: a
        # device-wide synchronize from one host thread invalidates a stream capture another thread's context has open (lanes)
        torch.cuda.current_stream(self.device).synchronize()
        rc = self.lib.vv_upload(self._ctx, name.encode(), C.c_void_p(t.data_ptr()),
                                1 if t.dtype == torch.bfloat16 else 0, t.numel())
        self._chk(rc, f"vv_upload({name})")
        self._loaded.add(name)
        # state the engine DERIVES from parameters must follow them (load_state_dict on a live model, LoRA merges): the
        # timestep-embedding table comes from head.t_embedder, the packed valid-token rows from lm_head / embed_tokens
        if name.startswith("head."):
            self._n_steps = None
        if name in ("lm.embed_tokens.weight", "lm_head.weight") and getattr(self, "_valid_ids", None):
            self.set_valid_tokens(self._valid_ids)

    def load_state_dict(self, sd: Dict[str, torch.Tensor], mapped=False, strict=True):
        """sd keyed by reference names (model.language_model....) unless mapped=True."""
        exp = self.expected_weights()
        for k, v in sd.items():
            name = k if mapped else map_param_name(k)
            if name is None or name not in exp:
                continue
            self.upload(name, v)
        if strict:
            miss = self.missing_weights()
            if miss:
                raise EngineError(f"{len(miss)} parameters were not provided, e.g. {miss[:5]}")

    def set_speech_factors(self, scaling: float, bias: float):
        self._chk(self.lib.vv_set_speech_factors(self._ctx, float(scaling), float(bias)), "vv_set_speech_factors")

    def set_valid_tokens(self, ids: Sequence[int]):
        arr = (C.c_int * len(ids))(*[int(i) for i in ids])
        self._chk(self.lib.vv_set_valid_tokens(self._ctx, arr, len(ids)), "vv_set_valid_tokens")
        self.n_valid = len(ids)
        self._valid_ids = [int(i) for i in ids]

    def set_num_steps(self, n_steps: int, t_cast_bf16: bool = False, algorithm_type: str = "dpmsolver++"):
        """Solver table for N steps.  algorithm_type: "dpmsolver++" (the model classes' scheduler) or "sde-dpmsolver++" (what
        demo/gradio_demo.py:142-146 installs); the stochastic one is sampled with diffusion_sample(..., step_noise=...)."""
        key = (int(n_steps), bool(t_cast_bf16), str(algorithm_type))
        if self._n_steps == key:
            return
        tv, coef = _schedule.make_table(n_steps, t_cast_bf16, algorithm_type)
        fn, name = ((self.lib.vv_set_schedule_sde, "vv_set_schedule_sde") if coef.shape[1] == 6
                    else (self.lib.vv_set_schedule, "vv_set_schedule"))
        self._chk(fn(self._ctx, n_steps, tv.ctypes.data_as(C.POINTER(C.c_float)),
                     np.ascontiguousarray(coef).ctypes.data_as(C.POINTER(C.c_float)), self._s), name)
        self._n_steps = key
        self.n_solver_steps = int(n_steps)
        self.stochastic = coef.shape[1] == 6

    # ------------------------------------------------------------------ ops
    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()

    @staticmethod
    def _rows(rows):
        """(cache, pos) pairs -> a VVRow array without a per-row Python loop (a 10,922-row prompt chunk is one call)"""
        a = np.ascontiguousarray(np.asarray(rows, dtype=np.int32).reshape(-1, 2))
        return (_lib.VVRow * a.shape[0]).from_buffer(a), a

    def lm_forward(self, rows: Sequence[tuple], x_in: torch.Tensor, hidden_out: torch.Tensor):
        arr, keep = self._rows(rows)
        self._chk(self.lib.vv_lm_forward(self._ctx, self._s, len(arr), arr, self._p(x_in), self._p(hidden_out)), "vv_lm_forward")

    def lm_forward_span(self, cache: int, pos0: int, n: int, x_in: torch.Tensor, hidden_out: torch.Tensor):
        """n consecutive positions pos0 .. pos0+n-1 of one cache (a prompt chunk)"""
        a = np.empty((n, 2), dtype=np.int32)
        a[:, 0] = cache
        a[:, 1] = np.arange(pos0, pos0 + n, dtype=np.int32)
        arr = (_lib.VVRow * n).from_buffer(a)
        self._chk(self.lib.vv_lm_forward(self._ctx, self._s, n, arr, self._p(x_in), self._p(hidden_out)), "vv_lm_forward")

    def lm_forward_range(self, rows: Sequence[tuple], x_in: torch.Tensor, hidden_out: torch.Tensor, l0: int, l1: int,
                         final_norm: bool):
        arr, keep = self._rows(rows)
        n = len(arr)
        self._chk(self.lib.vv_lm_forward_range(self._ctx, self._s, n, arr, self._p(x_in), self._p(hidden_out),
                                               int(l0), int(l1), int(final_norm)), "vv_lm_forward_range")

    def kv_import(self, cache: int, layer: int, k: torch.Tensor, v: torch.Tensor):
        """k, v: [kv_heads, n_pos, head_dim] (keys already rotated), fp32 or bf16, on the engine device."""
        assert k.shape == v.shape and k.dim() == 3
        k = k.contiguous()
        v = v.contiguous()
        if k.dtype not in (torch.float32, torch.bfloat16):
            k, v = k.float(), v.float()
        self._chk(self.lib.vv_kv_import(self._ctx, self._s, cache, layer, k.shape[1], self._p(k), self._p(v),
                                        1 if k.dtype == torch.bfloat16 else 0), "vv_kv_import")

    def kv_move(self, cache: int, src_pos: int, dst_pos: int):
        """cached position src_pos copied onto dst_pos, every layer of `cache` (keys keep their rotation)"""
        self._chk(self.lib.vv_kv_move(self._ctx, self._s, int(cache), int(src_pos), int(dst_pos)), "vv_kv_move")

    def kv_import_at(self, cache: int, layer: int, pos0: int, k: torch.Tensor, v: torch.Tensor):
        """k, v: [kv_heads, n_pos, head_dim] -> cache positions [pos0, pos0 + n_pos)."""
        assert k.shape == v.shape and k.dim() == 3
        k = k.contiguous()
        v = v.contiguous()
        if k.dtype not in (torch.float32, torch.bfloat16):
            k, v = k.float(), v.float()
        self._chk(self.lib.vv_kv_import_at(self._ctx, self._s, cache, layer, int(pos0), k.shape[1], self._p(k), self._p(v),
                                           1 if k.dtype == torch.bfloat16 else 0), "vv_kv_import_at")

    def add_type_embedding(self, n: int, x: torch.Tensor, type_id: int, out: torch.Tensor):
        self._chk(self.lib.vv_add_type_embedding(self._ctx, self._s, n, self._p(x), int(type_id), self._p(out)),
                  "vv_add_type_embedding")

    def eos_logit(self, n: int, hidden: torch.Tensor, out: torch.Tensor):
        self._chk(self.lib.vv_eos_logit(self._ctx, self._s, n, self._p(hidden), self._p(out)), "vv_eos_logit")

    @property
    def embed_chunk(self) -> int:
        """token ids one vv_embed call takes: max(64, max_rows)"""
        return max(64, int(self.cfg.max_rows))

    def embed(self, ids: Sequence[int], out: torch.Tensor):
        a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32).reshape(-1))
        arr = (C.c_int * a.shape[0]).from_buffer(a)
        self._chk(self.lib.vv_embed(self._ctx, self._s, a.shape[0], arr, self._p(out)), "vv_embed")

    def lm_logits(self, n: int, hidden: torch.Tensor, logits_out: torch.Tensor):
        self._chk(self.lib.vv_lm_logits(self._ctx, self._s, n, self._p(hidden), self._p(logits_out)), "vv_lm_logits")

    def lm_logits_full(self, n: int, hidden: torch.Tensor, logits_out: torch.Tensor):
        """logits over the whole vocabulary, [n, lm_vocab] fp32 (the full-vocabulary logits processors' input)"""
        assert logits_out.dtype == torch.float32 and logits_out.is_contiguous() and logits_out.numel() >= n * self.cfg.lm_vocab
        self._chk(self.lib.vv_lm_logits_full(self._ctx, self._s, n, self._p(hidden), self._p(logits_out)), "vv_lm_logits_full")

    def diffusion_sample(self, n: int, cond: torch.Tensor, noise: torch.Tensor, cfg_scale: float, latent_out: torch.Tensor,
                         step_noise: Optional[torch.Tensor] = None):
        """step_noise [n_steps, n, latent] fp32 (contiguous, on the device): the per-step variance noise of the stochastic solver."""
        if step_noise is None:
            self._chk(self.lib.vv_diffusion_sample(self._ctx, self._s, n, self._p(cond), self._p(noise),
                                                   float(cfg_scale), self._p(latent_out)), "vv_diffusion_sample")
        else:
            assert step_noise.dtype == torch.float32 and step_noise.is_contiguous() and step_noise.shape[1] == n
            self._chk(self.lib.vv_diffusion_sample_sde(self._ctx, self._s, n, self._p(cond), self._p(noise), self._p(step_noise),
                                                       float(cfg_scale), self._p(latent_out)), "vv_diffusion_sample_sde")

    def head_forward(self, noisy: torch.Tensor, t: float, cond: torch.Tensor, out: torch.Tensor):
        n = noisy.shape[0]
        tarr = (C.c_float * n)(*([float(t)] * n))
        self._chk(self.lib.vv_head_forward(self._ctx, self._s, n, self._p(noisy), tarr, self._p(cond), self._p(out)),
                  "vv_head_forward")

    def _sp(self, stream):
        """hipStream_t of an optional torch stream (default: the engine's own stream)"""
        return self._s if stream is None else C.c_void_p(stream.cuda_stream)

    def codec_decode(self, slot: int, latent: torch.Tensor, audio_out: torch.Tensor, apply_speech_factors=True, stream=None):
        self._chk(self.lib.vv_codec_decode(self._ctx, self._sp(stream), slot, 1, self._p(latent), self._p(audio_out),
                                           int(apply_speech_factors)), "vv_codec_decode")

    def semantic_encode(self, slot: int, audio: torch.Tensor, sem_out: torch.Tensor, stream=None):
        self._chk(self.lib.vv_semantic_encode(self._ctx, self._sp(stream), slot, 1, self._p(audio), self._p(sem_out)),
                  "vv_semantic_encode")

    def codec_chain_batch(self, slots, latent: torch.Tensor, audio_out: torch.Tensor, sem_out=None, apply_speech_factors=True):
        """One frame of len(slots) utterances through the acoustic decoder and (sem_out given) the semantic encoder: row j of
        latent [n, latent] / audio_out [n, hop] / sem_out [n, sem_dim] belongs to streaming slot slots[j].  The reference
        decodes / re-encodes the step's diffusion rows as one batch (modeling_vibevoice_inference.py:636-672); here the
        weight-heavy stages of both nets read their weights once for the whole batch."""
        n = len(slots)
        assert latent.is_contiguous() and audio_out.is_contiguous() and (sem_out is None or sem_out.is_contiguous())
        arr = (C.c_int * n)(*[int(s) for s in slots])
        self._chk(self.lib.vv_codec_chain_batch(self._ctx, self._s, n, arr, self._p(latent), self._p(audio_out),
                                                self._p(sem_out) if sem_out is not None else None, int(apply_speech_factors)),
                  "vv_codec_chain_batch")

    def acoustic_encode(self, frames: int, wav: torch.Tensor, mean_out: torch.Tensor, valid_samples: Optional[int] = None):
        """wav [frames * hop] -> mean_out [frames, latent].  valid_samples: real signal length when it does not fill the last frame (wav is
        zero beyond it): the reference's per-conv-layer right padding is reproduced (vv_acoustic_encode_ragged)."""
        if valid_samples is None or valid_samples >= frames * self.cfg.hop:
            self._chk(self.lib.vv_acoustic_encode(self._ctx, self._s, frames, self._p(wav), self._p(mean_out)), "vv_acoustic_encode")
        else:
            self._chk(self.lib.vv_acoustic_encode_ragged(self._ctx, self._s, frames, int(valid_samples), self._p(wav), self._p(mean_out)),
                      "vv_acoustic_encode_ragged")

    def set_enc_pass_frames(self, frames_per_pass: int):
        """frames per voice-prompt encoder pass, 1..cfg.enc_frames"""
        self._chk(self.lib.vv_set_enc_pass_frames(self._ctx, int(frames_per_pass)), "vv_set_enc_pass_frames")

    def audio_to_pcm16(self, audio: torch.Tensor, pcm_out: torch.Tensor, stream=None):
        """audio [n, samples] fp32 (contiguous) -> pcm_out [n, samples] int16, per-chunk peak normalisation as the reference's
        convert_to_16_bit_wav (demo/gradio_demo.py:1058-1073)."""
        n, samples = audio.shape
        assert audio.is_contiguous() and pcm_out.is_contiguous() and pcm_out.dtype == torch.int16 and audio.dtype == torch.float32
        self._chk(self.lib.vv_audio_to_pcm16(self._ctx, self._sp(stream), n, samples, self._p(audio), self._p(pcm_out)),
                  "vv_audio_to_pcm16")

    def codec_reset(self, slot: int):
        self._chk(self.lib.vv_codec_reset(self._ctx, self._s, slot), "vv_codec_reset")

    def connect(self, n: int, latent: torch.Tensor, sem: Optional[torch.Tensor], out: torch.Tensor):
        self._chk(self.lib.vv_connect(self._ctx, self._s, n, self._p(latent), self._p(sem), self._p(out)), "vv_connect")

    def profile_begin(self):
        self._chk(self.lib.vv_profile_begin(self._ctx), "vv_profile_begin")

    def profile_end(self):
        n, ms, by = (C.c_int64 * 2)(), (C.c_double * 2)(), (C.c_double * 2)()
        self._chk(self.lib.vv_profile_end(self._ctx, n, ms, by), "vv_profile_end")
        return (n[0], ms[0], by[0]), (n[1], ms[1], by[1])

    def profile_replay(self, reps=3, family=0):
        """(launches, total_ms, bytes) of the recorded launches of one kernel family replayed as one dependent hipGraph chain:
        family 0 = vv_gemv_kernel, 1 = vv_gemv16p_kernel (batch decode), 2 = decode attention (fused kernel + merge)."""
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        self._chk(self.lib.vv_profile_replay_family(self._ctx, self._s, int(family), int(reps), C.byref(n), C.byref(ms), C.byref(by)),
                  "vv_profile_replay_family")
        return n.value, ms.value, by.value

    def stat(self, what=0):
        return int(self.lib.vv_stat(self._ctx, what))

    # ------------------------------------------------------------------ low level (tests / microbench)
    def pack_matrix(self, w: torch.Tensor) -> torch.Tensor:
        N, K = w.shape
        out = torch.empty(int(self.lib.vv_packed_bytes(N, K)), dtype=torch.uint8, device=self.device)
        w = w.to(self.device, torch.float32).contiguous()
        with torch.cuda.stream(self.stream):
            rc = self.lib.vv_pack_matrix(self._s, self._p(w), self._p(out), N, K)
        if rc != 0:
            raise EngineError("vv_pack_matrix failed")
        self.sync()
        return out

    def gemm3_raw(self, wp, x, y, N, K, epi=0, w2p=None, nw=None, eps=1e-6, bias=None, ksplit=True):
        """the prefill GEMM (prefill.hip) on fp32 rows x [T, K] -> y [T, N]; ksplit=False computes every tile whole"""
        T = x.shape[0]
        xp = torch.zeros(int(self.lib.vv_packed_bytes(T, K)), dtype=torch.uint8, device=self.device)
        yp = torch.zeros(int(self.lib.vv_packed_bytes(T, N)), dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        rc = self.lib.vv_gemm3_raw(self._ctx if ksplit else None, self._s, self._p(wp), self._p(w2p), self._p(x), T, N, K, epi, self._p(nw), float(eps),
                                   self._p(bias), self._p(y), self._p(xp), self._p(yp))
        if rc != 0:
            raise EngineError(f"vv_gemm3_raw failed ({rc})")
        self.sync()

    def gemm_raw(self, wp, x, y, N, K, T=None, ldx=None, ldy=None, pro=0, epi=0, w2p=None, nw=None, eps=1e-6,
                 bias=None, nscale=None, xsplit=None, ksplit=0, nontemporal=0):
        T = x.shape[0] if T is None else T
        rc = self.lib.vv_gemm_raw(self._s, self._p(wp), self._p(w2p), self._p(x), self._p(y), T, N, K,
                                  ldx or K, ldy or N, pro, epi, self._p(nw), float(eps), self._p(bias),
                                  self._p(nscale), xsplit or self.cfg.xsplit, ksplit, nontemporal)
        if rc != 0:
            raise EngineError(f"vv_gemm_raw failed ({rc})")
