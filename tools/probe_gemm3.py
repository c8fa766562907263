#!/usr/bin/env python
"""Prefill GEMM probe: the four projection shapes of one 7B layer at a 10,922-token prompt through gemm3_raw (pack + GEMM
(+ unpack)); run under `rocprofv3 --kernel-trace --stats` and read the vv_gemm3 / vv_gemm4 kernel durations.

    python tools/probe_gemm3.py [--T 10922] [--reps 3]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=10922)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    from vibevoice_amd.engine import Engine, EngineConfig
    cfg = EngineConfig(lm_hidden=128, lm_layers=1, lm_heads=2, lm_kv_heads=1, lm_inter=256, lm_vocab=64, n_filters=4,
                       enc_depths=(1, 1, 1, 1, 1, 1, 2), head_layers=1, head_ffn_ratio=1.0, n_slots=1, max_ctx=64, max_rows=4, xsplit=1)
    eng = Engine(cfg)
    g = torch.Generator(device=eng.device).manual_seed(0)
    shapes = [("qkv", 4608, 3584, 1), ("o", 3584, 3584, 4), ("gate_up", 18944, 3584, 3), ("down", 3584, 18944, 4)]
    for name, N, K, epi in shapes:
        w = eng.pack_matrix(torch.randn(N, K, generator=g, device=eng.device) * 0.02)
        w2 = eng.pack_matrix(torch.randn(N, K, generator=g, device=eng.device) * 0.02) if epi == 3 else None
        x = torch.randn(a.T, K, generator=g, device=eng.device)
        y = torch.zeros(a.T, N, device=eng.device)
        bias = torch.zeros(N, device=eng.device) if epi == 1 else None
        t = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            with torch.cuda.stream(eng.stream):
                eng.gemm3_raw(w, x, y, N, K, epi=epi, w2p=w2, bias=bias)
            t.append(time.perf_counter() - t0)
        fl = 2.0 * a.T * N * K * (2 if epi == 3 else 1)
        print(f"{name:8s} N={N:6d} K={K:6d}: best wall {min(t) * 1e3:8.3f} ms incl. pack/unpack/alloc  ({fl / 1e12:.2f} TFLOP)", flush=True)
        del w, w2, x, y
    eng.close()


if __name__ == "__main__":
    main()
