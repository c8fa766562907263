#!/usr/bin/env python
"""Do two UNEQUAL sub-chains of one MLP, on two graph branches, hide each other's launch boundaries?

Observation behind it (profiles/r04_shared_gpu_n2_controlflow.json): two independent decode chains time-sharing one MI355X reach
1.28x (7B) / 1.63x (1.5B) the aggregate step rate of one -- the boundary (tail + dispatch + ramp, ~5 us of every dependent launch) of
one chain is filled by the other's weight stream.  One utterance has a single dependency line, but an MLP splits along its
intermediate dimension F into two chains that are independent until the down projection's partial sums are added:
    A: gate/up over features [0, F_A)  ->  down over K-range [0, F_A)          B: the same over [F_A, F)
With F_A != F_B the four kernel boundaries do not coincide: one branch streams while the other drains and refills.

    python tools/experiments/staggered_branches/bench_stagger.py [1.5b|7b] [steps]

Arms, each captured into ONE hipGraph of `steps` x 4 layers (two activation rows, the product's decode GEMV through vv_gemm_raw):
  chain      gate/up (RMSNorm prologue, SwiGLU epilogue), then down (+ residual): the product's launch chain
  halves     two branches, F split 1/2 : 1/2 (boundaries coincide: the control)
  staggered  two branches, F split 3/8 : 5/8
TIMING ONLY: branch B's down projection stores its partial sums into a side buffer (the product would add them in the next
prologue, as it does for the K-split parts of the LM's down projection); the residual stream of the two-branch arms is therefore not
the chain's, and no parity is claimed."""
import ctypes as C
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHAPES = {"1.5b": (1536, 4608), "7b": (3584, 10752)}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "7b"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    H, F = SHAPES[tag]
    L, eps = 4, 1e-5
    import synth
    from gpu_util import build_small
    eng = build_small(synth.LMCfg(), xsplit=1).eng
    dev = eng.device
    lib = eng.lib
    g = torch.Generator(device=dev).manual_seed(5)
    Wg = [(torch.randn(F, H, generator=g, device=dev) * H ** -0.5) for _ in range(L)]
    Wu = [(torch.randn(F, H, generator=g, device=dev) * H ** -0.5) for _ in range(L)]
    Wd = [(torch.randn(H, F, generator=g, device=dev) * 0.3 * F ** -0.5) for _ in range(L)]
    x0 = torch.randn(2, H, generator=g, device=dev)
    ones = torch.ones(H, device=dev)
    res = {"experiment": "staggered_branches", "shape": tag, "H": H, "F": F, "layers": L, "steps": S, "bytes_per_layer": 3 * H * F * 2}

    def P(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def gemv(stream, wp, x, y, N, K, pro=0, epi=0, w2p=None, nw=None):
        rc = lib.vv_gemm_raw(C.c_void_p(stream.cuda_stream), P(wp), P(w2p), P(x), P(y), 2, N, K, K, N, pro, epi, P(nw), float(eps),
                             None, None, 1, 0, 1)
        if rc != 0:
            raise RuntimeError(f"vv_gemm_raw failed ({rc})")

    s1 = eng.stream
    s2 = torch.cuda.Stream(device=dev)

    def time_graph(graph, reps=20, rounds=5):
        out = []
        for _ in range(rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s1):
                e0.record(s1)
                for _ in range(reps):
                    graph.replay()
                e1.record(s1)
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / reps)
        return sorted(out)[len(out) // 2], out

    def report(name, graph, launches):
        with torch.cuda.stream(s1):
            graph.replay()
        torch.cuda.synchronize()
        ms, allms = time_graph(graph)
        res[name] = {"ms_per_call": round(ms, 4), "us_per_layer": round(ms * 1e3 / (S * L), 3), "launches_per_layer": launches,
                     "GBps": round(res["bytes_per_layer"] * S * L / 1e9 / (ms / 1e3), 1), "all_ms": [round(t, 4) for t in allms]}

    # ---------------- the launch chain ----------------
    pg = [eng.pack_matrix(w) for w in Wg]
    pu = [eng.pack_matrix(w) for w in Wu]
    pd = [eng.pack_matrix(w) for w in Wd]
    xa = torch.empty(2, H, device=dev)
    ua = torch.empty(2, F, device=dev)
    with torch.cuda.stream(s1):
        xa.copy_(x0)
    torch.cuda.synchronize()
    g_chain = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_chain, stream=s1, capture_error_mode="relaxed"):
        for _ in range(S):
            for l in range(L):
                gemv(s1, pg[l], xa, ua, F, H, pro=1, epi=3, w2p=pu[l], nw=ones)
                gemv(s1, pd[l], ua, xa, H, F, pro=0, epi=4)
    report("chain", g_chain, 2)

    # ---------------- two branches ----------------
    def two_branches(name, FA):
        FB = F - FA
        assert FA % 32 == 0 and FB % 32 == 0
        pgA = [eng.pack_matrix(w[:FA].contiguous()) for w in Wg]
        puA = [eng.pack_matrix(w[:FA].contiguous()) for w in Wu]
        pgB = [eng.pack_matrix(w[FA:].contiguous()) for w in Wg]
        puB = [eng.pack_matrix(w[FA:].contiguous()) for w in Wu]
        pdA = [eng.pack_matrix(w[:, :FA].contiguous()) for w in Wd]
        pdB = [eng.pack_matrix(w[:, FA:].contiguous()) for w in Wd]
        uA = torch.empty(2, FA, device=dev)
        uB = torch.empty(2, FB, device=dev)
        yB = torch.empty(2, H, device=dev)
        xa2 = x0.clone()
        with torch.cuda.stream(s1):
            xa.copy_(x0)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s1, capture_error_mode="relaxed"):
            for _ in range(S):
                for l in range(L):
                    ev_x = torch.cuda.Event()
                    ev_x.record(s1)
                    s2.wait_event(ev_x)
                    gemv(s1, pgA[l], xa, uA, FA, H, pro=1, epi=3, w2p=puA[l], nw=ones)
                    gemv(s2, pgB[l], xa, uB, FB, H, pro=1, epi=3, w2p=puB[l], nw=ones)
                    gemv(s2, pdB[l], uB, yB, H, FB, pro=0, epi=0)          # partial sums to the side buffer (see the header)
                    ev_b = torch.cuda.Event()
                    ev_b.record(s2)
                    gemv(s1, pdA[l], uA, xa2, H, FA, pro=0, epi=4)         # + residual, into a second buffer: no write to the x that B1 reads
                    s1.wait_event(ev_b)
        report(name, gr, 4)
        res[name]["F_split"] = [FA, FB]
        del pgA, puA, pgB, puB, pdA, pdB

    two_branches("halves", F // 2 // 32 * 32)
    two_branches("staggered_3_8", F * 3 // 8 // 32 * 32)
    two_branches("staggered_1_4", F // 4 // 32 * 32)
    for k in ("halves", "staggered_3_8", "staggered_1_4"):
        res[k + "_over_chain"] = round(res[k]["ms_per_call"] / res["chain"]["ms_per_call"], 4)
    print(json.dumps(res))
    eng.close()


if __name__ == "__main__":
    main()
