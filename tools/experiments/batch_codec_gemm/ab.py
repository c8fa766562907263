"""The batched tokenizer stages' FFN GEMMs (eight utterances in flight: T = 8 x the stage's frame rows): the 16-row form of
vv_gemv_kernel that the batch-decode path launches today (vv_gemm_raw: RMSNorm prologue, bias + GELU epilogue) against the
LDS-staged MFMA tile GEMM of the prompt path on the same product (vv_gemm3_raw: norm in the operand pack, bias epilogue -- the
GELU is not in that kernel yet, so this is its lower bound).  Prints us per launch for each stage shape.

    python tools/experiments/batch_codec_gemm/ab.py          (on a GPU box)
"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "..", "tests"))
import numpy as np
import torch
import synth
from test_gpu_geometry import GEOM, build_fast


def main():
    s = build_fast(GEOM["0.5b"], xsplit=1, max_ctx=128, max_rows=1024, head_layers=1)
    eng = s.eng
    g = torch.Generator(device="cuda").manual_seed(1)
    print("T (8 utterances) x K -> N      16-row gemv form (us)   pack + tile GEMM (us)")
    for T, K, N in [(8 * 200, 256, 1024), (8 * 200, 1024, 256), (8 * 800, 128, 512), (8 * 800, 512, 128), (8 * 40, 512, 2048),
                    (8 * 40, 2048, 512), (8 * 3200, 32, 128), (8 * 3200, 128, 32)]:
        w = (torch.randn(N, K, generator=g, device="cuda") / np.sqrt(K)).to(torch.bfloat16).float()
        x = torch.randn(T, K, generator=g, device="cuda")
        nw = torch.ones(K, device="cuda")
        bias = torch.zeros(N, device="cuda")
        y = torch.zeros(T, N, device="cuda")
        wp = eng.pack_matrix(w.cpu())
        res = []
        for which in ("gemv16", "tile"):
            def run():
                if which == "gemv16":
                    eng.gemm_raw(wp, x, y, N, K, pro=1, epi=2, nw=nw, eps=1e-5, bias=bias)
                else:
                    eng.gemm3_raw(wp, x, y, N, K, epi=1, nw=nw, eps=1e-5, bias=bias)
            with torch.cuda.stream(eng.stream):
                for _ in range(3):
                    run()
            eng.sync()
            t0 = time.perf_counter()
            with torch.cuda.stream(eng.stream):
                for _ in range(20):
                    run()
            eng.sync()
            res.append((time.perf_counter() - t0) / 20 * 1e6)
        print(f"{T:6d} x {K:5d} -> {N:5d}        {res[0]:10.1f}            {res[1]:10.1f}   (includes host enqueue: gemm3_raw = pack + GEMM + a sync per call)")
    eng.close()


if __name__ == "__main__":
    main()
