"""ORACLE (test infrastructure) -- Qwen2 decoder forward with a compact KV cache.

The reference delegates the LM to a third-party dependency that is NOT under
/root/reference: `transformers==4.51.3` models/qwen2/modeling_qwen2.py
(pin: /root/reference/pyproject.toml:22; call sites
vibevoice/modular/modeling_vibevoice.py:121 `AutoModel.from_config(lm_config)`
and :187-199).  This file restates that published algorithm for the dense
Qwen2 decoder (identical in the installed transformers 5.15
modeling_qwen2.py: RMSNorm :238-252, rotary :51-135, eager attention
:150-173, attention module with q/k/v bias :176-235, MLP :35-48):

  h   = embeds
  per layer:  n = RMSNorm(h); q,k,v = Linear_bias(n); RoPE(q,k) (theta, rotate-half)
              K,V appended to the cache; GQA softmax(q k^T / sqrt(d)) v (fp32 softmax)
              h += o_proj(attn); h += down(silu(gate(n2)) * up(n2)), n2 = RMSNorm(h)
  out = RMSNorm(h)   (last_hidden_state)

One cache per utterance branch ("compact": no left padding, position == index).
HF semantics being restated: position_ids = cumsum(mask)-1 over
unmasked entries, cache grows by concatenation, causal mask.  With compact
per-row caches these reduce to position = current cache length.

weights dict keys follow Qwen2Model.state_dict():
  embed_tokens.weight, layers.{i}.input_layernorm.weight,
  layers.{i}.self_attn.{q,k,v}_proj.{weight,bias}, layers.{i}.self_attn.o_proj.weight,
  layers.{i}.post_attention_layernorm.weight, layers.{i}.mlp.{gate,up,down}_proj.weight,
  norm.weight
"""
import torch
import torch.nn.functional as F


def rmsnorm(x, weight, eps):
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)
    return weight * xf.to(x.dtype)


class KVCache:
    def __init__(self, n_layers):
        self.k = [None] * n_layers     # [kvh, L, d]
        self.v = [None] * n_layers
        self.length = 0

    def truncate(self, n):
        for i in range(len(self.k)):
            if self.k[i] is not None:
                self.k[i] = self.k[i][:, :n].contiguous()
                self.v[i] = self.v[i][:, :n].contiguous()
        self.length = n


class Qwen2Oracle:
    def __init__(self, weights, n_layers, n_heads, n_kv_heads, head_dim,
                 rope_theta=1e6, eps=1e-6, kv_round_bf16=False, rope_bf16=False, mfma_in_bf16=False, attn_rows=None):
        self.w = weights
        self.L, self.nh, self.nkv, self.d = n_layers, n_heads, n_kv_heads, head_dim
        self.theta, self.eps = rope_theta, eps
        self.kv_round_bf16 = kv_round_bf16   # emulate a bf16 KV cache (GPU bf16 path)
        self.rope_bf16 = rope_bf16           # emulate cos/sin cast to bf16 (HF casts to x.dtype)
        # emulate the HIP bf16 mode's rounding points: every matrix-unit INPUT is rounded to bf16 (the normed rows fed to
        # q/k/v and gate/up, the rotated query, the softmax weights, the attention output fed to o_proj, the SwiGLU product
        # fed to down_proj) while sums, norms, softmax statistics and the residual stream stay fp32 -- the same algorithm, the
        # roundings a bf16-activation GPU path (the reference on GPU rounds every op's output, :290-292 bf16 load) cannot
        # avoid.  Against this form the HIP kernels are held to ~1e-3 instead of the ~3e-2 the fp32 form allows.
        self.mfma_in_bf16 = mfma_in_bf16
        self.attn_rows = attn_rows           # query rows per attention block (None: all at once); bounds the score matrix
        self.inv_freq = 1.0 / (rope_theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))

    def new_cache(self):
        return KVCache(self.L)

    def embed(self, ids):
        return F.embedding(ids, self.w["embed_tokens.weight"])

    def _rope(self, x, pos):
        # x [T, heads, d]; pos [T]
        freqs = pos.float()[:, None] * self.inv_freq[None, :]
        emb = torch.cat([freqs, freqs], dim=-1)
        cos, sin = emb.cos(), emb.sin()
        if self.rope_bf16:
            cos, sin = cos.bfloat16().float(), sin.bfloat16().float()
        cos, sin = cos[:, None, :], sin[:, None, :]
        half = x.shape[-1] // 2
        rot = torch.cat([-x[..., half:], x[..., :half]], dim=-1)
        return (x * cos + rot * sin).to(x.dtype)     # fp32: a no-op; bf16 weights (bench.py's eager-GPU leg): back to the model dtype

    def forward(self, embeds, cache, final_norm=True):
        """embeds [T, H] appended at positions cache.length .. +T-1 (causal)."""
        w = self.w
        T = embeds.shape[0]
        p0 = cache.length
        pos = torch.arange(p0, p0 + T)
        h = embeds
        for i in range(self.L):
            p = f"layers.{i}."
            r16 = (lambda t: t.bfloat16().to(t.dtype)) if self.mfma_in_bf16 else (lambda t: t)
            n = r16(rmsnorm(h, w[p + "input_layernorm.weight"], self.eps))
            q = F.linear(n, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"]).view(T, self.nh, self.d)
            k = F.linear(n, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"]).view(T, self.nkv, self.d)
            v = F.linear(n, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"]).view(T, self.nkv, self.d)
            q = r16(self._rope(q, pos))
            k = self._rope(k, pos)
            if self.kv_round_bf16:
                k, v = k.bfloat16().float(), v.bfloat16().float()
            k = k.transpose(0, 1)
            v = v.transpose(0, 1)
            if cache.k[i] is None:
                cache.k[i], cache.v[i] = k, v
            else:
                cache.k[i] = torch.cat([cache.k[i][:, :p0], k], dim=1)
                cache.v[i] = torch.cat([cache.v[i][:, :p0], v], dim=1)
            K, V = cache.k[i], cache.v[i]                     # [kvh, p0+T, d]
            g = self.nh // self.nkv
            Kr = K.repeat_interleave(g, dim=0)                # [nh, Ltot, d]
            Vr = V.repeat_interleave(g, dim=0)
            Ltot = p0 + T
            blk = self.attn_rows or T
            a = torch.empty(T, self.nh * self.d, dtype=q.dtype, device=q.device)
            for t0 in range(0, T, blk):                       # the same attention, `blk` query rows at a time
                t1 = min(T, t0 + blk)
                Lb = int(pos[t1 - 1]) + 1                     # causal: later positions carry zero weight for these rows
                sc = torch.einsum("thd,hld->htl", q[t0:t1], Kr[:, :Lb]) * (self.d ** -0.5)
                mask = torch.arange(Lb)[None, :] > pos[t0:t1, None]
                sc = sc.masked_fill(mask[None], float("-inf"))
                if self.mfma_in_bf16:
                    # un-normalised weights exp(s - max) rounded to bf16, row sum kept in fp32 from the UNROUNDED weights,
                    # normalisation after the product -- flash attention's arithmetic (the weights it multiplies are exp(s - m))
                    ex = torch.exp(sc.float() - sc.float().amax(dim=-1, keepdim=True))
                    den = ex.sum(dim=-1, keepdim=True)
                    ab = torch.einsum("htl,hld->htd", r16(ex).to(q.dtype), Vr[:, :Lb]) / den.to(q.dtype)
                    a[t0:t1] = ab.transpose(0, 1).reshape(t1 - t0, self.nh * self.d)
                else:
                    pr = torch.softmax(sc.float(), dim=-1).to(q.dtype)
                    a[t0:t1] = torch.einsum("htl,hld->thd", pr, Vr[:, :Lb]).reshape(t1 - t0, self.nh * self.d)
            h = h + F.linear(r16(a), w[p + "self_attn.o_proj.weight"])
            n2 = r16(rmsnorm(h, w[p + "post_attention_layernorm.weight"], self.eps))
            mlp = F.linear(r16(F.silu(F.linear(n2, w[p + "mlp.gate_proj.weight"])) *
                               F.linear(n2, w[p + "mlp.up_proj.weight"])), w[p + "mlp.down_proj.weight"])
            h = h + mlp
        cache.length = p0 + T
        return rmsnorm(h, w["norm.weight"], self.eps) if final_norm else h
