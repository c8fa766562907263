"""ORACLE (test infrastructure) -- streaming acoustic decoder / tokenizer encoder.

Functional restatement over a state_dict-keyed weight dict of

  SConv1d._forward_streaming            modular_vibevoice_tokenizer.py:327-382
  SConv1d._forward_non_streaming        :384-418   (causal, pad_mode='constant')
  SConvTranspose1d._forward_streaming   :478-549
  SConvTranspose1d._forward_non_streaming :551-576
  ConvRMSNorm.forward                   :77-91
  FFN.forward                           :579-596   (exact-erf GELU, bias on)
  Block1D (cache-aware inline copy)     :924-942 / :786-804
  TokenizerDecoder.forward              :914-951
  TokenizerEncoder.forward              :776-813
  VibeVoiceTokenizerStreamingCache      :193-256   (state = dict name -> tensor)

Layout is the reference's [B, C, T].  `state` is a plain dict (one per
utterance); `state=None` selects the non-streaming path.
"""
import math

import torch
import torch.nn.functional as F


def conv_rmsnorm(x, weight, eps):
    xt = x.transpose(1, 2)
    out = (xt.float() * torch.rsqrt(xt.float().pow(2).mean(-1, keepdim=True) + eps)).type_as(xt)
    if weight is not None:
        out = out * weight
    return out.transpose(1, 2)


def sconv1d(x, w, b, state, key, stride=1, groups=1):
    k = w.shape[-1]
    ctx = (k - 1) - (stride - 1)
    if state is None:
        # non-streaming causal: left pad ctx zeros (+ right extra padding for stride)
        length = x.shape[-1]
        n_frames = (length - k + ctx) / stride + 1
        ideal = (math.ceil(n_frames) - 1) * stride + (k - ctx)
        x = F.pad(x, (ctx, ideal - length))
        return F.conv1d(x, w, b, stride=stride, groups=groups)
    cached = state.get(key)
    if cached is None:
        cached = torch.zeros(x.shape[0], x.shape[1], ctx, dtype=x.dtype)
    inp = torch.cat([cached, x], dim=2) if cached.shape[2] > 0 else x
    out = F.conv1d(inp, w, b, stride=stride, groups=groups)
    if ctx > 0:
        state[key] = inp[:, :, -ctx:] if inp.shape[2] >= ctx else inp
    return out


def sconvtr1d(x, w, b, state, key, stride):
    k = w.shape[-1]
    pad_total = k - stride            # causal, trim_right_ratio = 1 -> all on the right
    if state is None:
        y = F.conv_transpose1d(x, w, b, stride=stride)
        return y[..., : y.shape[-1] - pad_total] if pad_total > 0 else y
    T = x.shape[2]
    cached = state.get(key)
    if cached is None:
        cached = torch.zeros(x.shape[0], x.shape[1], 0, dtype=x.dtype)
    full_in = torch.cat([cached, x], dim=2)
    full_out = F.conv_transpose1d(full_in, w, b, stride=stride)
    if pad_total > 0:
        full_out = full_out[..., : full_out.shape[-1] - pad_total]
    if cached.shape[2] == 0:
        out = full_out
    else:
        want = T * stride
        out = full_out[:, :, -want:] if full_out.shape[2] >= want else full_out
    ctx = k - 1
    state[key] = full_in[:, :, -ctx:] if full_in.shape[2] > ctx else full_in
    return out


def block1d(x, w, p, state, eps):
    C = x.shape[1]
    res = x
    h = conv_rmsnorm(x, w[p + "norm.weight"], eps)
    h = sconv1d(h, w[p + "mixer.conv.conv.conv.weight"], w[p + "mixer.conv.conv.conv.bias"],
                state, p + "mixer", groups=C)
    h = h * w[p + "gamma"].unsqueeze(-1)
    x = res + h
    res = x
    h = conv_rmsnorm(x, w[p + "ffn_norm.weight"], eps).permute(0, 2, 1)
    h = F.linear(h, w[p + "ffn.linear1.weight"], w[p + "ffn.linear1.bias"])
    h = F.gelu(h)
    h = F.linear(h, w[p + "ffn.linear2.weight"], w[p + "ffn.linear2.bias"]).permute(0, 2, 1)
    h = h * w[p + "ffn_gamma"].unsqueeze(-1)
    return res + h


def decoder_forward(w, latents, ratios, depths, state=None, eps=1e-5, prefix="decoder."):
    """latents [B, vae_dim, T] -> audio [B, 1, T*prod(ratios)]."""
    x = sconv1d(latents, w[prefix + "upsample_layers.0.0.conv.conv.weight"],
                w[prefix + "upsample_layers.0.0.conv.conv.bias"], state, prefix + "stem")
    for i in range(len(depths)):
        if i > 0:
            pu = prefix + f"upsample_layers.{i}.0.convtr.convtr."
            x = sconvtr1d(x, w[pu + "weight"], w[pu + "bias"], state, pu, ratios[i - 1])
        for j in range(depths[i]):
            x = block1d(x, w, prefix + f"stages.{i}.{j}.", state, eps)
    return sconv1d(x, w[prefix + "head.conv.conv.weight"], w[prefix + "head.conv.conv.bias"],
                   state, prefix + "head")


def encoder_forward(w, audio, ratios, depths, state=None, eps=1e-5, prefix="encoder."):
    """audio [B, 1, T] -> latents [B, vae_dim, T/prod(ratios)].
    `ratios` is config.encoder_ratios; the encoder applies them reversed (:701)."""
    rr = list(reversed(ratios))
    x = audio
    for i in range(len(depths)):
        pd = prefix + f"downsample_layers.{i}.0.conv.conv."
        stride = 1 if i == 0 else rr[i - 1]
        x = sconv1d(x, w[pd + "weight"], w[pd + "bias"], state, pd, stride=stride)
        for j in range(depths[i]):
            x = block1d(x, w, prefix + f"stages.{i}.{j}.", state, eps)
    return sconv1d(x, w[prefix + "head.conv.conv.weight"], w[prefix + "head.conv.conv.bias"],
                   state, prefix + "head")


def zero_state(state):
    """VibeVoiceTokenizerStreamingCache.set_to_zero (:234-241) for one utterance."""
    for k in list(state.keys()):
        state[k] = torch.zeros_like(state[k])
