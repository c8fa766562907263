"""Wall-clock timeline of the GEMV launches of one decode step, from per-workgroup entry/exit stamps.

Needs a timing build:  tools/variant.sh t8 "-DVV_GEMM_TIMING"  (engine.hip must be compiled with the flag too:
VVHIP_CFLAGS=-DVV_GEMM_TIMING python -m vibevoice_amd.build --force), then
    VVHIP_TIMELINE=gpurun_out/tl.npz python bench.py --no-cpu-baseline --no-roofline --steps 20 --warmup 4
    python tools/step_timeline.py gpurun_out/tl.npz
"""
import ctypes
import sys

import numpy as np

TL_MAX, TL_STRIDE = 4096, 16 + 2 * 3200


def dump(eng, path):
    lib = eng.lib
    fn = lib.vv_timeline_dump
    fn.restype = ctypes.c_int
    buf = np.zeros((TL_MAX, TL_STRIDE), dtype=np.uint64)
    meta = np.zeros((TL_MAX, 5), dtype=np.int32)
    n = fn(eng._ctx, buf.ctypes.data_as(ctypes.c_void_p), meta.ctypes.data_as(ctypes.c_void_p), TL_MAX)
    np.savez_compressed(path, stamps=buf[:n], meta=meta[:n])
    print(f"[timeline] {n} launches -> {path}", file=sys.stderr)


def main(path):
    z = np.load(path)
    st, meta = z["stamps"].astype(np.int64), z["meta"]
    rows = []
    for i in range(len(meta)):
        T, N, K, pro, epi = meta[i]
        d = st[i, 16:].reshape(-1, 2)
        d = d[d[:, 0] > 0]                                # workgroups that stamped (the general kernel has 2-D grids)
        if len(d) == 0:
            continue
        d[:, 1] = np.maximum(d[:, 1], d[:, 0])       # the CFG+DPM epilogue returns before its exit stamp
        rows.append((d[:, 0].min(), d[:, 1].max(), d[:, 0].max(), np.median(d[:, 1] - d[:, 0]), len(d), T, N, K, pro, epi))
    rows.sort()
    tmax = rows[-1][1]
    rows = [r for r in rows if r[0] > tmax - 450000]      # last ~4.5 ms (100 MHz clock)
    t0 = rows[0][0]
    prev_end = None
    tot_span = tot_gap = 0.0
    print(" t_start  span   gap_prev  last_blk_start  blk_life_p50  blks    T     N     K pro epi(+100: general kernel)   MB    TB/s(span)")
    for (a, b, c, life, i, T, N, K, pro, epi) in rows:
        span = (b - a) * 0.01
        gap = (a - prev_end) * 0.01 if prev_end is not None else 0.0
        mb = N * K * 2 * (2 if epi == 3 else 1) / 1e6
        print(f"{(a-t0)*0.01:8.2f} {span:6.2f} {gap:8.2f} {(c-a)*0.01:10.2f} {life*0.01:12.2f} {i:5d}  {T:4d} {N:5d} {K:5d} {pro:3d} {epi:3d} {mb:6.1f} {mb/span:8.2f}")
        tot_span += span
        tot_gap += max(gap, 0.0)
        prev_end = b
    print(f"launches {len(rows)}  sum(span) {tot_span:.1f} us  sum(gaps incl. non-GEMV kernels) {tot_gap:.1f} us  window {(rows[-1][1]-t0)*0.01:.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
