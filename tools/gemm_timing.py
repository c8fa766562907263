"""Phase timestamps (s_memtime) of one workgroup of vv_gemm_kernel.  Needs a VV_GEMM_TIMING build:
   VVHIP_CFLAGS=-DVV_GEMM_TIMING python -m vibevoice_amd.build --force"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth
from gpu_util import build_small
s = build_small(synth.LMCfg(), xsplit=1)
eng = s.eng
names = ["entry->loads issued", "loads issued->x staged", "x staged->loop done", "loop done->partials written", "barrier", "reduce+epilogue"]
for (N, K, pro, epi) in [(64, 1536, 0, 0), (1536, 1536, 0, 4), (2048, 1536, 1, 1), (1536, 8960, 0, 4), (8960, 1536, 1, 3), (4608, 3584, 1, 1), (18944, 3584, 1, 3), (3584, 18944, 0, 4)]:
    w = torch.randint(0, 255, (int(eng.lib.vv_packed_bytes(N, K)),), dtype=torch.uint8, device=eng.device); w[1::2] &= 0x3F
    x = torch.randn(2, K, device=eng.device); y = torch.zeros(2, N, device=eng.device)
    nw = torch.ones(K, device=eng.device); bias = torch.zeros(N, device=eng.device)
    dbg = torch.zeros(16, dtype=torch.int64, device=eng.device)
    torch.cuda.synchronize()
    acc = None
    for it in range(6):
        with torch.cuda.stream(eng.stream):
            eng.gemm_raw(w, x, y, N, K, pro=pro, epi=epi, nw=nw, bias=bias, nscale=dbg, xsplit=1, nontemporal=2, w2p=(w if epi == 3 else None))
        eng.sync()
        d = dbg.cpu().tolist()
        if it >= 2:
            dl = [d[i + 1] - d[i] for i in range(6)]
            acc = dl if acc is None else [a + b for a, b in zip(acc, dl)]
    print(f"N={N} K={K} pro={pro} epi={epi}: total {sum(acc)/4:.0f} ticks :: " + ", ".join(f"{n}={v/4:.0f}" for n, v in zip(names, acc)))
