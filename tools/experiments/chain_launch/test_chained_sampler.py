"""the parity test of the chained sampler (ran green on the GPU with the experiment's build; needs tests/ on sys.path)"""
import dataclasses
import pytest
import torch
import synth
from gpu_util import build_small, rel_err
from oracle import dpm, head
from test_gpu_geometry import GEOM, build_fast, dev

# ---------------------------------------------------------------------------------------------- chained launches (chain.hip)
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["tiny", "7b", "1.5b", "0.5b"])
def test_chained_sampler_equals_the_launch_per_op_sampler(tag, monkeypatch):
    """vv_diffusion_sample for one utterance as ONE chained launch (VVHIP_CHAIN=1: every GEMV of every solver step a phase of one
    grid, hand-offs through agent-scope counters) against the same sampler issued as one launch per op, and against the
    oracle.  At the tiny widths only the in-workgroup K split differs (4 waves instead of 8): <= 1e-6 apart; at the real head
    widths the chain also splits K of the few-tile projections over workgroup columns, and the re-ordered fp32 sums move a bf16
    rounding here and there: <= 3e-3 apart, both within the bf16-mode bound of the bf16-input oracle.  Repeated calls replay the captured graph; the abort
    word of the chain kernels must stay 0."""
    lmcfg = synth.LMCfg() if tag == "tiny" else dataclasses.replace(GEOM[tag], inter=256)
    res = {}
    for arm in ("launches", "chain"):
        if arm == "chain":
            monkeypatch.setenv("VVHIP_CHAIN", "1")
        else:
            monkeypatch.delenv("VVHIP_CHAIN", raising=False)
        s = (build_small if tag == "tiny" else build_fast)(lmcfg, xsplit=1, use_graph=True, n_slots=1, max_ctx=128, max_rows=16,
                                                            head_layers=2 if tag == "tiny" else 4)
        eng = s.eng
        try:
            H = lmcfg.hidden
            g = synth.Gen(4242)
            pos = g.normal((1, H), 1.0, mat=False)
            neg = g.normal((1, H), 1.0, mat=False)
            noise = g.normal((2, 64), 1.0, mat=False)
            outs = []
            for N in (10, 5):
                eng.set_num_steps(N)
                lat = eng.new(1, 64)
                for rep in range(4):                  # eager, captured, replayed twice
                    lat.zero_()
                    with torch.cuda.stream(eng.stream):
                        eng.diffusion_sample(1, dev(torch.cat([pos, neg]), eng), dev(noise[:1], eng), 1.3, lat)
                    eng.sync()
                    outs.append(lat.float().cpu().clone())
                assert all(torch.equal(outs[-1], o) for o in outs[-4:]), "replays differ"
            assert eng.stat(5) == 0, f"chain abort word {eng.stat(5)}"
            assert (eng.stat(6) > 0) == (arm == "chain")
            with torch.no_grad():
                ref = dpm.sample_speech_tokens(lambda a, t, c: head.head_forward(s.head_w, a, t, c, s.hc.layers, s.hc.eps, mfma_in_bf16=True),
                                               pos, neg, 1.3, 5, noise)
            res[arm] = (outs[3], outs[7], rel_err(outs[7], ref))
        finally:
            eng.close()
    d10, d5 = rel_err(res["chain"][0], res["launches"][0]), rel_err(res["chain"][1], res["launches"][1])
    print(f"[chained sampler, {tag}] chain vs launches: N=10 {d10:.3e}, N=5 {d5:.3e}; vs bf16-input oracle: launches {res['launches'][2]:.3e}, chain {res['chain'][2]:.3e}")
    if tag == "tiny":
        assert d10 <= 1e-6 and d5 <= 1e-6, (d10, d5)             # same arithmetic, K split over 4 waves instead of 8: fp32 summation order only
    else:
        assert d10 <= 3e-3 and d5 <= 3e-3, (d10, d5)
    assert res["chain"][2] <= 1e-2 and res["launches"][2] <= 1e-2
