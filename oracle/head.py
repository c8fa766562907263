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


def head_forward(w, noisy, timesteps, condition, n_layers, eps=1e-5):
    x = F.linear(noisy, w["noisy_images_proj.weight"])
    t_freq = timestep_embedding(timesteps)
    t = F.linear(F.silu(F.linear(t_freq, w["t_embedder.mlp.0.weight"])),
                 w["t_embedder.mlp.2.weight"])
    c = F.linear(condition, w["cond_proj.weight"]) + t
    for i in range(n_layers):
        p = f"layers.{i}."
        mod = F.linear(F.silu(c), w[p + "adaLN_modulation.1.weight"])
        shift, scale, gate = mod.chunk(3, dim=-1)
        m = rmsnorm(x, w[p + "norm.weight"], eps) * (1 + scale) + shift
        u = F.silu(F.linear(m, w[p + "ffn.gate_proj.weight"])) * F.linear(m, w[p + "ffn.up_proj.weight"])
        x = x + gate * F.linear(u, w[p + "ffn.down_proj.weight"])
    mod = F.linear(F.silu(c), w["final_layer.adaLN_modulation.1.weight"])
    shift, scale = mod.chunk(2, dim=-1)
    x = rmsnorm(x, None, eps) * (1 + scale) + shift
    return F.linear(x, w["final_layer.linear.weight"])
