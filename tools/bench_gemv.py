"""Microbenchmark of the weight-streaming GEMM at LM decode shapes (T = 2 rows)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import synth
from gpu_util import build_small

s = build_small(synth.LMCfg(), xsplit=2)
eng = s.eng
shapes = {
    "7b_qkv": (4608, 3584, False), "7b_o": (3584, 3584, False), "7b_gateup": (18944, 3584, True), "7b_down": (3584, 18944, False),
    "1p5b_qkv": (2048, 1536, False), "1p5b_gateup": (8960, 1536, True), "1p5b_down": (1536, 8960, False),
    "dec_ffn1": (8192, 2048, False), "dec_ffn2": (2048, 8192, False),
    "1p5b_o": (1536, 1536, False), "1p5b_hdown": (1536, 4608, False), "1p5b_hgu": (4608, 1536, True), "1p5b_ada": (21504, 1536, False),
    "1p5b_final": (64, 1536, False),
}
T = 2
res = {}
for name, (N, K, dual) in shapes.items():
    nrep = 24   # rotate over distinct weight copies so L2/MALL (256 MiB) cannot serve the stream
    nbytes = int(eng.lib.vv_packed_bytes(N, K))
    ncopies = max(2, min(nrep, int(3e9 // (nbytes * (2 if dual else 1)))))
    ws = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=eng.device) for _ in range(ncopies)]
    # make bf16 patterns finite: clear exponent msb of every 2nd byte
    for w in ws:
        w[1::2] &= 0x3F
    w2s = [w.clone() for w in ws] if dual else [None] * ncopies
    x = torch.randn(T, K, device=eng.device)
    y = torch.zeros(T, N, device=eng.device)
    torch.cuda.synchronize()
    for xs in (1,):
        for ks in (0, 8):
            with torch.cuda.stream(eng.stream):
                for i in range(ncopies):
                    eng.gemm_raw(ws[i], x, y, N, K, epi=3 if dual else 0, w2p=w2s[i], xsplit=xs, ksplit=ks, nontemporal=1)
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(eng.stream)
                for rep in range(3):
                    for i in range(ncopies):
                        eng.gemm_raw(ws[i], x, y, N, K, epi=3 if dual else 0, w2p=w2s[i], xsplit=xs, ksplit=ks, nontemporal=1)
                e1.record(eng.stream)
            eng.sync()
            ms = e0.elapsed_time(e1) / (3 * ncopies)
            gb = nbytes * (2 if dual else 1) / 1e9
            res[f"{name}_xs{xs}_ks{ks}"] = {"us": ms * 1e3, "GBps": gb / (ms * 1e-3), "MB": gb * 1e3}
            print(f"{name:12s} xs={xs} ks={ks} N={N} K={K} {gb*1e3:8.1f} MB  {ms*1e3:8.1f} us  {gb/(ms*1e-3):8.0f} GB/s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_gemv.json", "w"), indent=1)
