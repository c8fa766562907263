"""Fine-tuned assets on the HIP path: what the reference's `load_lora_assets`
(vibevoice/modular/lora_loading.py:140-176) does by wrapping live modules in peft, done here at weight-snapshot time:
a LoRA pair is folded into its base matrix, W' = W + (alpha / r) * B @ A, and the merged tensor is re-uploaded (the
engine re-packs it into MFMA tiles); full diffusion-head / connector state dicts are uploaded as they are.

Adapter layout read (the one the reference's trainer writes, finetune/train_vibevoice.py:161-176):
    <dir>/lora/adapter_config.json + adapter_model.{safetensors,bin}                  language model (peft)
    <dir>/lora/diffusion_head/adapter_config.json + adapter_model.*                    diffusion head (peft)
    <dir>/lora/diffusion_head/diffusion_head_full.bin | <dir>/lora/diffusion_head_full.bin   full head state dict
    <dir>/lora/{acoustic,semantic}_connector/pytorch_model.bin                        connector state dicts
peft itself is not needed (and not installed here): only its file format is read.
"""
import json
import os
import re
from dataclasses import dataclass
from typing import Callable, Dict, Iterator, Optional, Tuple

import torch

LM_PREFIX = "model.language_model."
HEAD_PREFIX = "model.prediction_head."
AC_CONN_PREFIX = "model.acoustic_connector."
SEM_CONN_PREFIX = "model.semantic_connector."


@dataclass
class LoadReport:
    language_model: bool = False
    diffusion_head_lora: bool = False
    diffusion_head_full: bool = False
    acoustic_connector: bool = False
    semantic_connector: bool = False
    adapter_root: Optional[str] = None
    merged_tensors: int = 0


def lora_scale(cfg: dict) -> float:
    """peft's scaling: alpha / r, or alpha / sqrt(r) with rank-stabilised LoRA."""
    r = int(cfg["r"])
    alpha = float(cfg.get("lora_alpha", r))
    return alpha / (r ** 0.5) if cfg.get("use_rslora") else alpha / r


MERGE_DTYPES = ("float32", "bfloat16")


def merge_lora(weight: torch.Tensor, a: torch.Tensor, b: torch.Tensor, scale: float, fan_in_fan_out=False,
               merge_dtype: str = "float32") -> torch.Tensor:
    """W' = W + scale * B @ A (A: [r, in], B: [out, r]); returned in W's dtype.

    merge_dtype selects WHERE the delta is rounded, i.e. which of peft's two code paths the merged tensor equals bit for bit
    (peft/tuners/lora/layer.py: Linear.get_delta_weight + `base_layer.weight.data += delta_weight`, what the reference's
    scripts/merge_vibevoice_models.py:67-88 reaches through `merge_and_unload()`):

    "float32" (default) -- the adapter matrices are fp32 when they are merged: peft's loader upcasts half-precision adapter weights
        (`autocast_adapter_dtype=True`, its default) and the reference's trainer saves them in fp32.  delta = (B @ A) * scale is fp32,
        and the in-place `+=` onto a bf16 parameter computes fp32(W) + delta and rounds ONCE: W' = bf16(fp32(W) + scale * B @ A).
    "bfloat16" -- the adapter matrices are bf16 like the model (adapters cast with the model, or `autocast_adapter_dtype=False`), merged
        on the CPU as the reference's merge script does: get_delta_weight upcasts A and B (`cast_to_fp32`), forms (B @ A) * scale in
        fp32, ROUNDS THE DELTA TO bf16, and the `+=` rounds again: W' = bf16(fp32(W) + fp32(bf16(scale * bf16(B) @ bf16(A)))).
        (On a GPU peft multiplies in bf16 instead -- one more rounding of the product before the scale; not reproduced here.)  The
        result differs from the default by at most one bf16 ulp of the larger of |W| and |delta| per element.

    Default "float32": it is what the reference's own merge script produces with the assets its trainer writes, and it is the
    more accurate of the two (one rounding)."""
    if merge_dtype not in MERGE_DTYPES:
        raise ValueError(f"merge_dtype must be one of {MERGE_DTYPES}, got {merge_dtype!r}")
    if merge_dtype == "bfloat16":
        delta = b.to(torch.bfloat16).to(torch.float32) @ a.to(torch.bfloat16).to(torch.float32)
        if fan_in_fan_out:
            delta = delta.t()
        delta = (delta * scale).to(torch.bfloat16)
    else:
        delta = b.to(torch.float32) @ a.to(torch.float32)
        if fan_in_fan_out:
            delta = delta.t()
        delta = scale * delta
    if tuple(delta.shape) != tuple(weight.shape):
        raise ValueError(f"LoRA delta {tuple(delta.shape)} does not match the base weight {tuple(weight.shape)}")
    return (weight.to(torch.float32) + delta.to(torch.float32)).to(weight.dtype)


_LORA_KEY = re.compile(r"^(?:base_model\.model\.)?(?P<mod>.+?)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")


def lora_pairs(adapter_sd: Dict[str, torch.Tensor], ref_prefix: str, strip: str = "") -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """{reference checkpoint key: (A, B)} from a peft adapter state dict.  `strip`: wrapper attribute peft saw in front
    of the real module path (the reference wraps the head in a shim whose attribute is `base`)."""
    halves: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in adapter_sd.items():
        m = _LORA_KEY.match(k)
        if not m:
            continue
        mod = m.group("mod")
        if strip and mod.startswith(strip):
            mod = mod[len(strip):]
        halves.setdefault(ref_prefix + mod + ".weight", {})[m.group("ab")] = v
    out = {}
    for k, h in halves.items():
        if "A" not in h or "B" not in h:
            raise ValueError(f"incomplete LoRA pair for {k}")
        out[k] = (h["A"], h["B"])
    return out


def _read_state(path_no_ext: str) -> Optional[Dict[str, torch.Tensor]]:
    if os.path.exists(path_no_ext + ".safetensors"):
        from safetensors.torch import load_file
        return load_file(path_no_ext + ".safetensors")
    if os.path.exists(path_no_ext + ".bin"):
        return torch.load(path_no_ext + ".bin", map_location="cpu", weights_only=True)
    return None


def _adapter(dirname: str):
    cfg_p = os.path.join(dirname, "adapter_config.json")
    if not os.path.exists(cfg_p):
        return None
    sd = _read_state(os.path.join(dirname, "adapter_model"))
    if sd is None:
        return None
    with open(cfg_p) as f:
        return json.load(f), sd


def resolve_adapter_root(checkpoint_dir: str) -> str:
    """lora_loading.py:47-55"""
    p = checkpoint_dir
    if os.path.isfile(p):
        p = os.path.dirname(p)
    return os.path.join(p, "lora") if os.path.isdir(os.path.join(p, "lora")) else p


def planned_updates(adapter_root: str, base: Callable[[str], torch.Tensor], merge_dtype: str = "float32") -> Iterator[Tuple[str, torch.Tensor, str]]:
    """Yields (reference key, tensor to upload, kind) for everything found under adapter_root.
    base(key) returns the base checkpoint tensor for a LoRA target; merge_dtype: see merge_lora."""
    lm = _adapter(adapter_root)
    if lm is not None:
        cfg, sd = lm
        for k, (a, b) in sorted(lora_pairs(sd, LM_PREFIX).items()):
            yield k, merge_lora(base(k), a, b, lora_scale(cfg), bool(cfg.get("fan_in_fan_out")), merge_dtype), "language_model"
    hd = _adapter(os.path.join(adapter_root, "diffusion_head"))
    if hd is not None:
        cfg, sd = hd
        for k, (a, b) in sorted(lora_pairs(sd, HEAD_PREFIX, strip="base.").items()):
            yield k, merge_lora(base(k), a, b, lora_scale(cfg), bool(cfg.get("fan_in_fan_out")), merge_dtype), "diffusion_head_lora"
    else:
        for p in (os.path.join(adapter_root, "diffusion_head", "diffusion_head_full.bin"), os.path.join(adapter_root, "diffusion_head_full.bin")):
            if os.path.exists(p):
                for k, v in torch.load(p, map_location="cpu", weights_only=True).items():
                    yield HEAD_PREFIX + k, v, "diffusion_head_full"
                break
    for sub, prefix, kind in (("acoustic_connector", AC_CONN_PREFIX, "acoustic_connector"), ("semantic_connector", SEM_CONN_PREFIX, "semantic_connector")):
        p = os.path.join(adapter_root, sub, "pytorch_model.bin")
        if os.path.exists(p):
            for k, v in torch.load(p, map_location="cpu", weights_only=True).items():
                yield prefix + k, v, kind


def load_lora_assets(model, checkpoint_dir: str, base_state: Optional[Callable[[str], torch.Tensor]] = None,
                     merge_dtype: str = "float32") -> LoadReport:
    """Drop-in for the reference's `load_lora_assets(model, checkpoint_dir)`.  `model` is the HIP-path
    VibeVoiceForConditionalGenerationInference; LoRA targets need their base tensors: `base_state(key)` or, by default,
    the safetensors checkpoint the model was loaded from (`model.source_path`).  merge_dtype ("float32" default | "bfloat16"): where
    the LoRA delta is rounded -- which of peft's merge paths the uploaded tensors equal bit for bit (merge_lora).  Forked engine
    contexts (model.fork(), generate_interleaved lanes) must be closed first: they hold snapshots derived from the parameters."""
    from .engine import map_param_name
    root = resolve_adapter_root(checkpoint_dir)
    if not os.path.isdir(root):
        raise FileNotFoundError(f"Adapter directory not found: {root}")
    if base_state is None:
        base_state = getattr(model, "base_tensor", None)
        if base_state is None:
            raise ValueError("load_lora_assets needs the base weights: load the model with from_pretrained() or pass base_state")
    rep = LoadReport(adapter_root=root)
    exp = model.engine.expected_weights()
    for key, tensor, kind in planned_updates(root, base_state, merge_dtype):
        name = map_param_name(key)
        if name is None or name not in exp:
            continue
        model.engine.upload(name, tensor)
        setattr(rep, kind, True)
        rep.merged_tensors += 1
    return rep
