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
SOURCES = ["gemm.hip", "gemv.hip", "chain.hip", "gemv16p.hip", "tile.hip", "prefill.hip", "attn.hip", "misc.hip", "block1d.hip", "engine.hip"]
HEADERS = [os.path.join(CSRC, "vv_common.h"), os.path.join(CSRC, "gemv_body.h"), os.path.join(os.path.dirname(HERE), "include", "vvhip.h")]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or ""):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libvvhip.so cannot be built")


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-unused-result", "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def _extra_flags():
    return os.environ.get("VVHIP_CFLAGS", "").split()


def source_id():
    """sha256[:16] over the kernel sources, the headers and the compile flags: what the binary must have been built from."""
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS + _extra_flags()).encode())
    return h.hexdigest()[:16]


def binary_id(path=LIB):
    """The id compiled into libvvhip.so (vv_build_id), read from the file without loading it; None if absent."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    k = blob.find(b"VVHIP_BUILD_ID=")
    if k < 0:
        return None
    return blob[k + 15:k + 31].decode("ascii", "replace")


def have_sources():
    return all(os.path.exists(d) for d in [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def stale():
    """True when the in-tree binary is missing or was not compiled from the sources beside it (content hash, not mtime: a
    checkout or a snapshot copy resets mtimes either way)."""
    if not os.path.exists(LIB):
        return True
    return have_sources() and binary_id() != source_id()


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    tmp = f"{LIB}.tmp{os.getpid()}"            # unique per process: concurrent ranks never share a half-written file
    cmd = [_hipcc()] + FLAGS + _extra_flags() + [f'-DVV_BUILD_ID="{source_id()}"'] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ["-o", tmp]
    if verbose:
        print("[vibevoice_amd] building libvvhip.so:", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    os.replace(tmp, LIB)
    return LIB


def build_locked():
    """Rebuild a stale binary under an exclusive file lock: under torchrun every rank finds the same stale .so at the same time;
    the first one compiles (about two minutes), the others block on the lock and then find the binary current."""
    import fcntl
    with open(LIB + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if stale():
                build(force=True, verbose=False)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
