"""ORACLE (test infrastructure) -- SpeechConnector.forward
(vibevoice/modular/modeling_vibevoice.py:58-69): fc2(RMSNorm_1e-6(fc1(x))),
both Linear with bias; LlamaRMSNorm computes in fp32 and multiplies the
weight after casting back.

weights: fc1.weight/bias, norm.weight, fc2.weight/bias
"""
import torch
import torch.nn.functional as F


def connector_forward(w, x, eps=1e-6):
    h = F.linear(x, w["fc1.weight"], w["fc1.bias"])
    hf = h.float()
    hf = hf * torch.rsqrt(hf.pow(2).mean(-1, keepdim=True) + eps)
    h = w["norm.weight"] * hf.to(h.dtype)
    return F.linear(h, w["fc2.weight"], w["fc2.bias"])
