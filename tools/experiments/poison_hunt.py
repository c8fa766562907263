"""Uninitialised-read hunt on ANY GPU: every byte of free HBM is filled with NaN words and handed back to the driver, the torch cache is left
holding NaN blocks, the library hands out its un-zeroed allocations as NaN words (VVHIP_POISON=1) -- then the bench's 1.5B leg with its parity
block.  A read of memory nobody wrote turns into a non-finite latent here instead of on one GPU of the pool in ten."""
import os, sys, importlib.util
os.environ.setdefault("VVHIP_POISON", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
device = torch.device("cuda", 0); torch.cuda.set_device(device)
nan = float("nan")
if os.environ.get("POISON_HBM", "1") == "1":
    free, total = torch.cuda.mem_get_info()
    chunks = [torch.full((1 << 28,), nan, device=device) for _ in range(int(free * 0.92) // (1 << 30))]
    torch.cuda.synchronize(); print(f"[poison] {len(chunks)} GiB of NaN words written", flush=True)
    del chunks; torch.cuda.empty_cache()
    blocks = [torch.full((s,), nan, device=device) for s in (128, 1024, 3072, 8192, 1 << 15, 1 << 17, 1 << 19, 1 << 21, 1 << 23) for _ in range(6)]
    torch.cuda.synchronize(); del blocks
ctx = dict(rank=0, world=1, device=device, use_dist=False)
wl = sys.argv[1] if len(sys.argv) > 1 else "1p5b"
a2 = bench.parse_args(["--workload", wl] + sys.argv[2:])
if wl == "1p5b" and "--steps" not in sys.argv: a2.steps, a2.warmup = 60, 10
r2 = bench.bench_decode(a2, dict(bench.WORKLOADS[wl]), ctx, with_cpu=False, with_roofline=True, with_parity=True)
p = r2.get("parity") or {}
print("[main]", wl, r2["ms_per_step"], "parity", p.get("within_bounds"), (p.get("vs_fp32") or {}).get("latent"), (p.get("vs_fp32") or {}).get("nonfinite_steps"),
      "| long", ((r2.get("parity_long") or {}).get("within_bounds")), flush=True)
