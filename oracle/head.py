"""ORACLE (test infrastructure) -- diffusion head forward, functional form.

Restates VibeVoiceDiffusionHead.forward
(vibevoice/modular/modular_vibevoice_diffusion_head.py:254-280) over a plain
dict of weights keyed like the reference module's state_dict():

  noisy_images_proj.weight [H,64]   cond_proj.weight [H,H]
  t_embedder.mlp.0.weight [H,256]   t_embedder.mlp.2.weight [H,H]
  layers.{i}.norm.weight [H]        layers.{i}.adaLN_modulation.1.weight [3H,H]
  layers.{i}.ffn.{gate,up}_proj.weight [F,H]   layers.{i}.ffn.down_proj.weight [H,F]
  final_layer.adaLN_modulation.1.weight [2H,H] final_layer.linear.weight [64,H]
"""
import math

import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps):
    """RMSNorm.forward, modular_vibevoice_diffusion_head.py:31-38."""
    out = (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + eps)).type_as(x)
    if weight is not None:
        out = out * weight
    return out


def timestep_embedding(t, dim=256, max_period=10000):
    """TimestepEmbedder.timestep_embedding, :66-88 (returns in t.dtype)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    return emb.to(t.dtype)


def head_forward(w, noisy, timesteps, condition, n_layers, eps=1e-5, mfma_in_bf16=False):
    """mfma_in_bf16: round every matrix-unit INPUT to bf16 (sums, norms and the residual stay fp32) -- the rounding points
    of the HIP bf16 mode, see oracle/lm.py; the adaLN-modulated norm is fed as rs * W.(x*w*(1+scale)) + W.shift with the
    two operands rounded separately, which is the same linear map."""
    r = (lambda t: t.bfloat16().to(t.dtype)) if mfma_in_bf16 else (lambda t: t)
    x = F.linear(r(noisy), w["noisy_images_proj.weight"])
    t_freq = timestep_embedding(timesteps)
    t = F.linear(r(F.silu(F.linear(r(t_freq), w["t_embedder.mlp.0.weight"]))),
                 w["t_embedder.mlp.2.weight"])
    c = F.linear(r(condition), w["cond_proj.weight"]) + t

    def modulated(x, nw, scale, shift, W):
        if not mfma_in_bf16:
            return F.linear(rmsnorm(x, nw, eps) * (1 + scale) + shift, W)
        xf = x.float()
        rs = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
        xm = xf * (1 + scale) if nw is None else xf * nw * (1 + scale)
        return rs * F.linear(r(xm), W) + F.linear(r(shift), W)
    for i in range(n_layers):
        p = f"layers.{i}."
        mod = F.linear(r(F.silu(c)), w[p + "adaLN_modulation.1.weight"])
        shift, scale, gate = mod.chunk(3, dim=-1)
        nw = w[p + "norm.weight"]
        u = F.silu(modulated(x, nw, scale, shift, w[p + "ffn.gate_proj.weight"])) * modulated(x, nw, scale, shift, w[p + "ffn.up_proj.weight"])
        x = x + gate * F.linear(r(u), w[p + "ffn.down_proj.weight"])
    mod = F.linear(r(F.silu(c)), w["final_layer.adaLN_modulation.1.weight"])
    shift, scale = mod.chunk(2, dim=-1)
    return modulated(x, None, scale, shift, w["final_layer.linear.weight"])
