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
SOURCES = ["gemm.hip", "gemv.hip", "gemv16p.hip", "headtail.hip", "tile.hip", "prefill.hip", "attn.hip", "misc.hip", "block1d.hip", "engine.hip"]
HEADERS = [os.path.join(CSRC, "vv_common.h"), os.path.join(os.path.dirname(HERE), "include", "vvhip.h")]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", shutil.which("hipcc") or ""):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libvvhip.so cannot be built")


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fvisibility-inlines-hidden",
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


def _object_key(src, flags, build_id):
    """content hash of one translation unit: its source, every header, the flags (+ the build id where the source embeds it)"""
    import hashlib
    h = hashlib.sha256()
    for d in [src] + HEADERS:
        with open(d, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    if os.path.basename(src) == "engine.hip":
        h.update(build_id.encode())
    return h.hexdigest()[:16]


def build(force=False, verbose=True, _locked=False):
    """One object per source (compiled in parallel, cached under csrc/.obj by content hash), then one link: a change to one kernel
    file recompiles that file only (the whole library in one hipcc command took 2.5 minutes)."""
    if not force and not stale():
        return LIB
    if not _locked:                       # a plain `python -m vibevoice_amd.build` beside a build_locked() rank: same lock, one builder
        import fcntl
        with open(LIB + ".lock", "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if not force and not stale():
                    return LIB
                return build(force=force, verbose=verbose, _locked=True)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    from concurrent.futures import ThreadPoolExecutor
    hipcc = _hipcc()
    bid = source_id()
    cflags = [f for f in FLAGS if f != "-shared"] + _extra_flags()
    objdir = os.path.join(CSRC, ".obj")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for sname in SOURCES:
        src = os.path.join(CSRC, sname)
        obj = os.path.join(objdir, f"{os.path.splitext(sname)[0]}.{_object_key(src, cflags, bid)}.o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        tmp = f"{obj}.tmp{os.getpid()}"
        cmd = [hipcc] + cflags + [f'-DVV_BUILD_ID="{bid}"', "-c", src, "-o", tmp]
        if verbose:
            print("[vibevoice_amd] compiling", os.path.basename(src), file=sys.stderr)
        subprocess.run(cmd, check=True)
        os.replace(tmp, obj)
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(compile_one, jobs))
    keep = set(objs)
    for fn in os.listdir(objdir):                       # objects of older revisions of the same sources
        full = os.path.join(objdir, fn)
        if fn.endswith(".o") and full not in keep:
            try:
                os.remove(full)
            except OSError:
                pass
    tmp = f"{LIB}.tmp{os.getpid()}"            # unique per process: concurrent ranks never share a half-written file
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + _extra_flags() + objs + ["-o", tmp]
    if verbose:
        print("[vibevoice_amd] linking libvvhip.so", file=sys.stderr)
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
                build(force=True, verbose=False, _locked=True)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
