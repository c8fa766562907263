#!/usr/bin/env python
"""Decode-attention probe: a ONE-layer engine at a model's attention geometry (thin MLP), a KV cache filled to L positions,
`steps` decode steps.  Run it under `rocprofv3 --kernel-trace --stats` and read vv_attn_fused_kernel's average duration
(tools/rocprof_summary.py); prints the wall time per step too.

    python tools/probe_attn.py --geom 7b --len 32000 --rows 1 --steps 200
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GEOM = {"7b": (3584, 28, 4), "1.5b": (1536, 12, 2), "0.5b": (896, 14, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="7b")
    ap.add_argument("--len", type=int, default=32000)
    ap.add_argument("--rows", type=int, default=1, help="utterances (each: one long row + one short negative row)")
    ap.add_argument("--steps", type=int, default=200)
    a = ap.parse_args()
    from vibevoice_amd.engine import Engine, EngineConfig
    H, hq, hkv = GEOM[a.geom]
    cfg = EngineConfig(lm_hidden=H, lm_layers=1, lm_heads=hq, lm_kv_heads=hkv, lm_inter=256, lm_vocab=64, n_filters=4,
                       enc_depths=(1, 1, 1, 1, 1, 1, 2), head_layers=1, head_ffn_ratio=0.25, n_slots=a.rows, max_ctx=a.len + a.steps + 64,
                       max_rows=2 * a.rows, xsplit=1, use_graph=True)
    eng = Engine(cfg)
    g = torch.Generator(device=eng.device).manual_seed(0)
    for name, n in eng.expected_weights().items():
        eng.upload(name, torch.randn(n, generator=g, device=eng.device) * 0.02)
    d = H // hq
    for r in range(a.rows):
        k = (torch.randn(hkv, a.len, d, generator=g, device=eng.device)).to(torch.bfloat16)
        v = (torch.randn(hkv, a.len, d, generator=g, device=eng.device)).to(torch.bfloat16)
        eng.kv_import(2 * r, 0, k, v)
    x = torch.randn(2 * a.rows, H, generator=g, device=eng.device) * 0.1
    out = eng.new(2 * a.rows, H)
    with torch.cuda.stream(eng.stream):
        for i in range(5):
            eng.lm_forward([(2 * r, a.len + i) for r in range(a.rows)] + [(2 * r + 1, i) for r in range(a.rows)], x, out)
        eng.sync()
        t0 = time.perf_counter()
        for i in range(5, 5 + a.steps):
            eng.lm_forward([(2 * r, a.len + i) for r in range(a.rows)] + [(2 * r + 1, i) for r in range(a.rows)], x, out)
        eng.sync()
        dt = time.perf_counter() - t0
    kv_bytes = a.rows * 2 * hkv * d * 2 * a.len
    print(f"probe_attn geom={a.geom} len={a.len} rows={a.rows}: {dt / a.steps * 1e6:.2f} us per one-layer step "
          f"(KV {kv_bytes / 1e6:.1f} MB -> {kv_bytes / 1e3 / (dt / a.steps * 1e6):.0f} GB/s if the step were attention only)", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
