"""Occupancy timeline of one vv_gemv_kernel launch: per-workgroup entry/exit wall-clock stamps (100 MHz).
   Needs a VV_GEMM_TIMING build (tools/variant.sh t8 "-DVV_GEMM_TIMING"; VVHIP_LIB=build/variants/libvvhip_t8.so)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, synth, numpy as np
from gpu_util import build_small
s = build_small(synth.LMCfg(), xsplit=1)
eng = s.eng
for (N, K, pro, epi) in [(1536, 1536, 0, 4), (2048, 1536, 1, 1), (1536, 8960, 0, 4), (8960, 1536, 1, 3), (4608, 1536, 1, 3), (18944, 3584, 1, 3), (3584, 18944, 0, 4)]:
    nt = (N + 15) // 16
    ws = []
    for i in range(4):
        w = torch.randint(0, 255, (int(eng.lib.vv_packed_bytes(N, K)),), dtype=torch.uint8, device=eng.device); w[1::2] &= 0x3F
        ws.append(w)
    x = torch.randn(2, K, device=eng.device); y = torch.zeros(2, N, device=eng.device)
    nw = torch.ones(K, device=eng.device); bias = torch.zeros(N, device=eng.device)
    dbg = torch.zeros(16 + 2 * nt, dtype=torch.int64, device=eng.device)
    torch.cuda.synchronize()
    for it in range(4):
        w = ws[it]
        with torch.cuda.stream(eng.stream):
            eng.gemm_raw(w, x, y, N, K, pro=pro, epi=epi, nw=nw, bias=bias, nscale=dbg, xsplit=1, nontemporal=2, w2p=(ws[(it + 1) % 4] if epi == 3 else None))
        eng.sync()
    d = dbg.cpu().numpy()[16:].reshape(nt, 2).astype(np.int64)
    t0 = d[:, 0].min()
    st = (d[:, 0] - t0) * 0.01; en = (d[:, 1] - t0) * 0.01      # us
    life = en - st
    mb = int(eng.lib.vv_packed_bytes(N, K)) * (2 if epi == 3 else 1) / 1e6
    print(f"N={N} K={K} pro={pro} epi={epi} {mb:.1f} MB blocks={nt}: span {en.max():.2f} us ({mb/en.max()/1e3:.2f} TB/s) | start p50 {np.median(st):.2f} p90 {np.percentile(st,90):.2f} max {st.max():.2f} | "
          f"life min {life.min():.2f} p50 {np.median(life):.2f} max {life.max():.2f} | end p10 {np.percentile(en,10):.2f} p50 {np.median(en):.2f}")
    hist, _ = np.histogram(st, bins=np.arange(0, en.max() + 1, 1.0))
    print("   starts per us:", hist.tolist())
