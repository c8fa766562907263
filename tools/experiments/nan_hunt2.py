"""The bench's own sequence (7B leg, then the 1.5B extras leg with its parity block) with a probe behind the parity comparison: when the
engine's latents come back non-finite, replay the sampler on the traced inputs and say which call, which mode and which buffer."""
import os, sys, json, importlib.util
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from oracle import parity as oparity, generate as ogen
from vibevoice_amd import synthetic

orig = oparity.compare_engine
def probe(model, leg, tokens, frames=None, also=None):
    res = orig(model, leg, tokens, frames=frames, also=also)
    print("[probe] compare_engine:", {k: res.get(k) for k in ("frames", "latent", "pos_hidden", "nonfinite_steps")}, flush=True)
    if res.get("nonfinite_steps"):
        e = model.engine
        T = tokens
        # run it again with a trace to get the engine's own conditions
        htr = ogen.Trace()
        n = leg.frames
        out = model.generate(tokenizer=tokens, cfg_scale=leg.cfg_scale, generation_config={"do_sample": False}, max_new_tokens=n, show_progress_bar=False,
                             _forced_tokens=[[T.speech_diffusion_id] * n + [T.eos_token_id]], _noise_fn=lambda step, n2: leg.noise[step][:n2], _trace=htr,
                             _teacher_embeds=lambda step, rows: leg.trace.next_embeds[step][rows].float(), input_ids=leg.ids, attention_mask=torch.ones_like(leg.ids))
        fin = [bool(torch.isfinite(x).all()) for x in htr.latents]
        print("[probe] second traced run: latents finite per step", fin, flush=True)
        for step in range(n):
            cond = torch.cat([htr.pos_hidden[step][:1], htr.neg_hidden[step][:1]]).to(e.device, torch.float32).contiguous()
            nz = leg.noise[step][:1].to(e.device, torch.float32).contiguous()
            outs = []
            for rep in range(3):
                o = e.new(1, 64)
                with torch.cuda.stream(e.stream):
                    e.diffusion_sample(1, cond, nz, leg.cfg_scale, o)
                e.sync()
                outs.append(o.clone())
            print(f"[probe] step {step}: cond finite {bool(torch.isfinite(cond).all())} absmax {float(cond.abs().max()):.3e}; noise absmax {float(nz.abs().max()):.3f}; "
                  f"direct sampler x3 finite {[bool(torch.isfinite(x).all()) for x in outs]} absmax {[float(x.abs().max()) for x in outs]}", flush=True)
        # the model's own buffers as the speculative path uses them
        print("[probe] model._hidden finite", bool(torch.isfinite(model._hidden).all()), "absmax", float(model._hidden.abs().max()),
              "| _noise finite", bool(torch.isfinite(model._noise).all()), "| _cond finite", bool(torch.isfinite(model._cond).all()), flush=True)
        with torch.cuda.stream(e.stream):
            o = e.new(1, 64)
            e.diffusion_sample(1, model._hidden, model._noise, leg.cfg_scale, o)
        e.sync()
        print("[probe] sampler on model._hidden / model._noise:", bool(torch.isfinite(o).all()), float(o.abs().max()), flush=True)
        rows_h = model._hidden
        for r in range(min(4, rows_h.shape[0])):
            print(f"[probe] _hidden row {r}: finite {bool(torch.isfinite(rows_h[r]).all())} absmax {float(rows_h[r].abs().max()):.3e}", flush=True)
    return res
oparity.compare_engine = probe

sys.argv = ["bench.py", "--no-config3", "--no-cpu-baseline", "--no-eager-baseline"] + sys.argv[1:]
# the extras need the parity legs although the CPU baseline is off: patch the flag the extras derive from it
_pa = bench.parse_args
def pa(argv=None):
    a = _pa(argv)
    return a
bench.parse_args = pa
args = bench.parse_args()
device = torch.device("cuda", 0); torch.cuda.set_device(device)
ctx = dict(rank=0, world=1, device=device, use_dist=False)
r = bench.bench_decode(args, dict(bench.WORKLOADS["north-star"]), ctx, with_cpu=False, with_roofline=True, with_parity=False)
print("[main] 7B leg", r["ms_per_step"], flush=True)
a2 = bench.parse_args([]); a2.steps, a2.warmup = 60, 10
r2 = bench.bench_decode(a2, dict(bench.WORKLOADS["1p5b"]), ctx, with_cpu=False, with_roofline=True, with_parity=True)
print("[main] 1.5B leg", r2["ms_per_step"], (r2["parity"] or {}).get("within_bounds"), ((r2["parity"] or {}).get("vs_fp32") or {}).get("nonfinite_steps"), flush=True)
