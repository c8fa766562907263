#!/usr/bin/env python
"""A/B of the diffusion head's MLP chain (S solver steps x 4 layers x {RMSNorm + gate/up + SwiGLU, down + residual}, two
activation rows) as (a) dependent launches of the product's decode GEMV (libvvhip vv_gemm_raw, captured into one hipGraph,
as the product replays its sampler) and (b) ONE persistent launch on the loader / consumer engine (lc.hip).

    python tools/experiments/loader_consumer/bench_lc.py [1.5b|7b|test] [S]

Prints one JSON line: us per layer (A + B) for both arms, the max relative difference of the final residual stream after ONE
step (parity) and after S steps, the persistent kernel's abort word (0 = every spin ended normally)."""
import ctypes as C
import json
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = {"1.5b": (1536, 4608), "7b": (3584, 10752), "test": (1024, 3072)}


def build_lib():
    so = os.path.join(HERE, os.environ.get("LC_SO", "liblc.so"))
    src = os.path.join(HERE, "lc.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + os.environ.get("LC_CFLAGS", "").split() + [src, "-o", so], check=True)
    lib = C.CDLL(so)
    lib.lc_run.restype = C.c_int
    lib.lc_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_int, C.c_int, C.c_float]
    return lib


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "1.5b"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    H, F = SHAPES[tag]
    L, eps = 4, 1e-5
    import synth
    from gpu_util import build_small
    eng = build_small(synth.LMCfg(), xsplit=1).eng          # any engine: only its stream, pack_matrix and gemm_raw are used
    dev = eng.device
    g = torch.Generator(device=dev).manual_seed(11)
    Wg = [(torch.randn(F, H, generator=g, device=dev) * H ** -0.5).to(torch.bfloat16) for _ in range(L)]
    Wu = [(torch.randn(F, H, generator=g, device=dev) * H ** -0.5).to(torch.bfloat16) for _ in range(L)]
    Wd = [(torch.randn(H, F, generator=g, device=dev) * 0.3 * F ** -0.5).to(torch.bfloat16) for _ in range(L)]
    x0 = torch.randn(2, H, generator=g, device=dev)
    res = {"experiment": "loader_consumer", "shape": tag, "H": H, "F": F, "layers": L, "steps": S,
           "bytes_per_layer": 3 * H * F * 2}

    # ---------------- arm (a): the launch chain ----------------
    pg = [eng.pack_matrix(w.float()) for w in Wg]
    pu = [eng.pack_matrix(w.float()) for w in Wu]
    pd = [eng.pack_matrix(w.float()) for w in Wd]
    ones = torch.ones(H, device=dev)
    xa = torch.empty(2, H, device=dev)
    ua = torch.empty(2, F, device=dev)

    def chain(steps):
        for _ in range(steps):
            for l in range(L):
                eng.gemm_raw(pg[l], xa, ua, F, H, T=2, pro=1, epi=3, w2p=pu[l], nw=ones, eps=eps, xsplit=1, nontemporal=1)
                eng.gemm_raw(pd[l], ua, xa, H, F, T=2, pro=0, epi=4, xsplit=1, nontemporal=1)

    def run_chain_eager(steps):
        with torch.cuda.stream(eng.stream):
            xa.copy_(x0)
            chain(steps)
        eng.sync()
        return xa.clone()
    ref1 = run_chain_eager(1)
    refS = run_chain_eager(S)

    def torch_ref(steps, layers=L):
        """the chain in plain torch ops (fp32 accumulate, bf16-rounded matrix inputs as both arms round them)"""
        x = x0.clone()
        for _ in range(steps):
            for l in range(layers):
                xb = x.to(torch.bfloat16).float()
                rs = torch.rsqrt((x * x).mean(-1, keepdim=True) + eps)
                gg = (xb @ Wg[l].float().T) * rs
                uu = (xb @ Wu[l].float().T) * rs
                u = (torch.nn.functional.silu(gg) * uu).to(torch.bfloat16).float()
                x = x + u @ Wd[l].float().T
        return x
    t1 = torch_ref(1)
    res["launch_chain_vs_torch_after_1_step"] = ((ref1 - t1).norm() / t1.norm()).item()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(eng.stream):
        xa.copy_(x0)
    eng.sync()
    with torch.cuda.graph(graph, stream=eng.stream, capture_error_mode="relaxed"):
        chain(S)
    eng.sync()

    def time_it(fn, reps=20, rounds=5):
        out = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(eng.stream):
                e0.record(eng.stream)
                for _ in range(reps):
                    fn()
                e1.record(eng.stream)
            eng.sync()
            out.append(e0.elapsed_time(e1) / reps)
        return sorted(out)[len(out) // 2], out
    with torch.cuda.stream(eng.stream):
        graph.replay()
    eng.sync()
    ms_a, all_a = time_it(lambda: graph.replay())
    res["launch_chain"] = {"ms_per_call": round(ms_a, 4), "us_per_layer": round(ms_a * 1e3 / (S * L), 3),
                           "GBps": round(res["bytes_per_layer"] * S * L / 1e9 / (ms_a / 1e3), 1), "all_ms": [round(t, 4) for t in all_a],
                           "note": "vv_gemm_raw: RMS + SwiGLU GEMV, then residual GEMV (no 3-column K split of the down projection: the engine adds "
                                   "that on top for few-tile x long-K shapes)"}

    # ---------------- arm (b): the persistent loader / consumer launch ----------------
    lib = build_lib()
    nA, nB = F // 256, H // 256
    per_layer = []
    for l in range(L):
        a = torch.stack([Wg[l].view(256, nA, H), Wu[l].view(256, nA, H)], dim=2).reshape(256, nA * 2 * H)
        b = Wd[l].view(256, nB, F).reshape(256, nB * F)
        per_layer.append(torch.cat([a, b], dim=1))
    wstream = torch.cat(per_layer, dim=1).contiguous()           # [256][L * per-layer elements] bf16
    assert (wstream.shape[1] * 2) % 1024 == 0
    pad = (-wstream.shape[1] * 2) % 16384                        # one solver step = whole 16 KiB ring slots
    if pad:
        wstream = torch.cat([wstream, torch.zeros(256, pad // 2, dtype=wstream.dtype, device=dev)], dim=1).contiguous()
    xg = torch.zeros(H, dtype=torch.int64, device=dev)
    ug = torch.zeros(F, dtype=torch.int64, device=dev)
    xo = torch.zeros(2, H, device=dev)
    ab = torch.zeros(1, dtype=torch.int32, device=dev)
    st = C.c_void_p(eng.stream.cuda_stream)

    def run_lc(steps):
        rc = lib.lc_run(st, H, F, C.c_void_p(wstream.data_ptr()), C.c_void_p(xg.data_ptr()), C.c_void_p(ug.data_ptr()),
                        C.c_void_p(x0.data_ptr()), C.c_void_p(xo.data_ptr()), C.c_void_p(ab.data_ptr()), None, L, steps, eps)
        if rc != 0:
            raise RuntimeError(f"lc_run failed ({rc})")
    torch.cuda.synchronize()
    with torch.cuda.stream(eng.stream):
        run_lc(1)
    eng.sync()
    res["abort_word_first_call"] = int(ab.item())
    d1 = ((xo - ref1).norm() / ref1.norm()).item()
    res["persistent_vs_torch_after_1_step"] = ((xo - t1).norm() / t1.norm()).item()
    # timeline of one call (S steps): consumer wave 0 of CU 0 / CU 131 stamps every op (gather start, own gather done, all
    # waves' gather done, compute done; 100 MHz wall clock); the loader reports its total / ring-full / vmcnt-wait time
    dbg = torch.zeros(64 + 128 * 4, dtype=torch.int64, device=dev)
    with torch.cuda.stream(eng.stream):
        rc = lib.lc_run(st, H, F, C.c_void_p(wstream.data_ptr()), C.c_void_p(xg.data_ptr()), C.c_void_p(ug.data_ptr()),
                        C.c_void_p(x0.data_ptr()), C.c_void_p(xo.data_ptr()), C.c_void_p(ab.data_ptr()), C.c_void_p(dbg.data_ptr()), L, min(S, 8), eps)
    eng.sync()
    d = dbg.cpu().numpy()
    tl = {}
    for name, off, lo in (("cu0", 64, 0), ("cu131", 64 + 64 * 4, 8)):
        ops = d[off:off + 64 * 4].reshape(64, 4)[8:min(64, 2 * L * min(S, 8))]      # skip the first layers (cold)
        A, B = ops[0::2], ops[1::2]
        f = lambda x: round(float(x.mean()) * 0.01, 2)
        tl[name] = {"A_gather_own_us": f(A[:, 1] - A[:, 0]), "A_gather_wait_others_us": f(A[:, 2] - A[:, 1]), "A_compute_us": f(A[:, 3] - A[:, 2]),
                    "B_gather_own_us": f(B[:, 1] - B[:, 0]), "B_gather_wait_others_us": f(B[:, 2] - B[:, 1]), "B_compute_us": f(B[:, 3] - B[:, 2]),
                    "loader_total_us": round(float(d[lo]) * 0.01, 1), "loader_ring_full_us": round(float(d[lo + 1]) * 0.01, 1),
                    "loader_vmcnt_wait_us": round(float(d[lo + 2]) * 0.01, 1), "loader_slots": int(d[lo + 3])}
    res["timeline"] = tl
    with torch.cuda.stream(eng.stream):
        run_lc(S)
    eng.sync()
    res["abort_word"] = int(ab.item())
    dS = ((xo - refS).norm() / refS.norm()).item()
    res["rel_diff_after_1_step"] = d1
    res["rel_diff_after_S_steps"] = dS
    if res["abort_word"] == 0 and res["abort_word_first_call"] == 0:
        ms_b, all_b = time_it(lambda: run_lc(S), reps=10)
        res["persistent"] = {"ms_per_call": round(ms_b, 4), "us_per_layer": round(ms_b * 1e3 / (S * L), 3),
                             "GBps": round(res["bytes_per_layer"] * S * L / 1e9 / (ms_b / 1e3), 1), "all_ms": [round(t, 4) for t in all_b],
                             "abort_word_after_timing": int(ab.item())}
        res["persistent_over_launch_chain"] = round(ms_b / ms_a, 4)
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
