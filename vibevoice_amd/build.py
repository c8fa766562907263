"""Build libvvhip.so (hipcc, gfx950 only) in-tree.

    python -m vibevoice_amd.build          # rebuild if sources are newer than the .so
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvvhip.so")
SOURCES = ["gemm.hip", "gemv.hip", "gemv16p.hip", "tile.hip", "prefill.hip", "attn.hip", "misc.hip", "block1d.hip", "engine.hip"]
HEADERS = [os.path.join(CSRC, "vv_common.h"), os.path.join(os.path.dirname(HERE), "include", "vvhip.h")]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or ""):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libvvhip.so cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"] + os.environ.get("VVHIP_CFLAGS", "").split() + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print("[vibevoice_amd] building libvvhip.so:", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
