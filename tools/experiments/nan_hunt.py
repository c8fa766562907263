"""Diagnostic for the box-dependent NaN of the 1.5B parity block: the extras-leg path of bench.py (GPU fp32 oracle as the teacher) in isolation,
with per-step dumps of what the sampler saw."""
import os, sys, json, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import parity as oparity, generate as ogen
from vibevoice_amd import synthetic
from vibevoice_amd.configs import CONFIGS
from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
key = sys.argv[1] if len(sys.argv) > 1 else "1.5b"
NS = 10 if key == "1.5b" else 20
cfg = copy.deepcopy(CONFIGS[key])
sd = dict(synthetic.random_state_dict(cfg, dev, seed=0))
model = VibeVoiceForConditionalGenerationInference.from_state_dict(cfg, sd, torch.bfloat16, None, n_slots=1, max_ctx=1024, xsplit=1, use_graph=True,
                                                                   enc_frames=75, max_rows=512)
model.set_speech_factors(0.2, -0.05)
T = synthetic.TOKENS
for attempt in range(3):
    fp32 = oparity.oracle_leg(cfg, sd, T, NS, 1.3, 4, dev, torch.float32, 20.0)
    print(f"[attempt {attempt}] oracle fp32 leg: frames {fp32.frames}, noise entries {len(fp32.noise)}, next_embeds {len(fp32.trace.next_embeds)}, "
          f"finite: latents {all(bool(torch.isfinite(x).all()) for x in fp32.trace.latents)}, embeds {all(bool(torch.isfinite(x).all()) for x in fp32.trace.next_embeds)}, "
          f"noise {all(bool(torch.isfinite(x).all()) for x in fp32.noise)}; |next_embeds| max {max(float(x.abs().max()) for x in fp32.trace.next_embeds):.3e}")
    for i, nz in enumerate(fp32.noise):
        print(f"   noise[{i}] shape {tuple(nz.shape)} dtype {nz.dtype} absmax {float(nz.abs().max()):.3f}")
    res = oparity.compare_engine(model, fp32, T)
    print("   engine:", {k: res[k] for k in ("frames", "latent", "pos_hidden", "neg_hidden", "nonfinite_steps")})
    if res["nonfinite_steps"]:
        break
print("capture fallbacks", model.engine.stat(4))
model.engine.close()
