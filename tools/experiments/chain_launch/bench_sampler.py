#!/usr/bin/env python
"""A/B of the diffusion sampler (vv_diffusion_sample, one utterance) as ~10 N dependent launches vs ONE chained launch
(csrc/chain.hip), at a model's head widths, inside hipGraph replay as in the timed loop.

    python tools/bench_sampler.py [7b|1.5b|0.5b] [N]

Prints one JSON line: ms per sampler call for both arms (alternating, several rounds), max |latent difference|."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WIDTHS = {"7b": (3584, 28, 4, 18944), "1.5b": (1536, 12, 2, 8960), "0.5b": (896, 14, 2, 4864)}


def build(tag, chain):
    from vibevoice_amd.engine import Engine, EngineConfig
    H, heads, kvh, inter = WIDTHS[tag]
    if chain:
        os.environ["VVHIP_CHAIN"] = "1"
    else:
        os.environ.pop("VVHIP_CHAIN", None)
    cfg = EngineConfig(lm_hidden=H, lm_layers=1, lm_heads=heads, lm_kv_heads=kvh, lm_inter=inter, lm_vocab=64, head_layers=4,
                       n_filters=4, ratios=(8, 5, 5, 4, 2, 2), enc_depths=(1, 1, 1, 1, 1, 1, 2), sem_dim=0, has_acoustic_encoder=False, n_slots=1, max_ctx=128,
                       max_rows=16, xsplit=1, use_graph=True)
    eng = Engine(cfg)
    g = torch.Generator(device=eng.device).manual_seed(5)
    for name, n in eng.expected_weights().items():
        if name == "lm.rope.inv_freq":
            continue
        std = 0.02 if ("weight" in name and not name.endswith("norm.weight")) else 0.0
        t = torch.randn(n, generator=g, device=eng.device) * (std if std else 0.1) + (0.0 if std else 1.0)
        if "adaLN" in name or "final_layer.linear" in name:
            t = torch.randn(n, generator=g, device=eng.device) * 0.03
        eng.upload(name, t.to(torch.bfloat16))
    return eng


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "7b"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    H = WIDTHS[tag][0]
    res = {"model": tag, "solver_steps": N}
    outs = {}
    engines = {arm: build(tag, arm == "chain") for arm in ("launches", "chain")}
    g = torch.Generator(device="cuda").manual_seed(1)
    cond = torch.randn(2, H, generator=g, device="cuda")
    noise = torch.randn(1, 64, generator=g, device="cuda")
    times = {arm: [] for arm in engines}
    for arm, eng in engines.items():
        eng.set_num_steps(N)
        lat = eng.new(1, 64)
        outs[arm] = lat
        with torch.cuda.stream(eng.stream):
            for _ in range(4):                       # eager, capture, replay
                eng.diffusion_sample(1, cond, noise, 1.3, lat)
        eng.sync()
    for rnd in range(5):
        for arm, eng in engines.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eng.stream):
                e0.record(eng.stream)
                for _ in range(20):
                    eng.diffusion_sample(1, cond, noise, 1.3, outs[arm])
                e1.record(eng.stream)
            eng.sync()
            times[arm].append(e0.elapsed_time(e1) / 20.0)
    for arm, eng in engines.items():
        res[arm + "_ms"] = round(sorted(times[arm])[len(times[arm]) // 2], 4)
        res[arm + "_all_ms"] = [round(t, 4) for t in times[arm]]
        res[arm + "_chain_err"] = eng.stat(5)
        res[arm + "_chains"] = eng.stat(6)
    a, b = outs["launches"].float().cpu(), outs["chain"].float().cpu()
    res["max_abs_diff"] = float((a - b).abs().max())
    res["rel_l2_diff"] = float((a - b).norm() / a.norm())
    res["finite"] = bool(torch.isfinite(b).all())
    res["speedup"] = round(res["launches_ms"] / res["chain_ms"], 4)
    print(json.dumps(res))
    for eng in engines.values():
        eng.close()


if __name__ == "__main__":
    main()
