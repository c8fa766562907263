#!/usr/bin/env python
"""Random forced token plans through the REFERENCE's own generate() (tiny seeded model, tests/golden/make_golden.py's machinery) and
through the oracle loop + the product loop (on the oracle-backed CPU engine of tests/fake_engine.py): token sequences, length flags,
RNG draw counts and waveforms must agree.  Build container only (needs /root/reference); nothing is written into the repository.

    python tools/fuzz_generate_vs_reference.py [n_random_plans] [seed] [--norefresh | --sampled | --warped | --mixed | --streaming]

--sampled: free-running token sampling instead of forced plans (product loop only; pins the RNG consumption order in batches).
--warped: random full-vocabulary logits processors (repetition penalty, temperature, top-k / top-p / min-p), product loop only.
--mixed: batches of 1-4, forced or greedy, length caps, ragged voice samples, the stochastic scheduler (product loop only).

This is how round 4 found the reference's cross-row tokenizer-cache coupling: a plan that starts one row's first
frame later than another's."""
import json
import os
import random
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def hits_single_entry_pattern(plans, refresh_negative=True):
    """Walks the forced plans with the bookkeeping of oracle.generate.NegativeRow (mask, entry count, correction counter -- no
    tensors) and says whether a correction ever meets `entries - correct_cnt == 2` with a valid entry at correct_cnt: the case in
    which the reference's mask shift happens and its K/V shift does not (modeling_vibevoice_inference.py:603 vs :613)."""
    B = len(plans)
    mask, c, cnt, fin = [[1] for _ in range(B)], [0] * B, [0] * B, [False] * B
    for step in range(max(len(p) for p in plans)):
        tok = [plans[b][step] if step < len(plans[b]) else "X" for b in range(B)]
        for b in range(B):
            if fin[b]:
                tok[b] = "X"

        def fwd():
            for b in range(B):
                c[b] += 1
                mask[b].append(1)
        if not refresh_negative:
            fwd()
        for b in range(B):
            fin[b] = fin[b] or tok[b] == "X"
        if refresh_negative:
            for b in range(B):
                if not fin[b] and tok[b] == "S":
                    mask[b] = [0] * len(mask[b])
                    mask[b][-1] = 1
        if any(not fin[b] and tok[b] == "D" for b in range(B)):
            if refresh_negative:
                fwd()
            for b in range(B):
                if not fin[b] and tok[b] != "D":
                    s_, n = cnt[b], len(mask[b])
                    if c[b] - s_ == 2 and mask[b][s_] == 1:
                        return True
                    if s_ + 1 < n - 1:
                        mask[b][s_ + 1:] = mask[b][s_:-1]
                    mask[b][s_] = 0
                    cnt[b] += 1
    return False


def streaming_main(n, seed):
    """--streaming: random (text length, length cap, EOS-classifier bias) triples through the reference's Streaming-0.5B generate()
    (tiny split model, prefilled branches from its own forward passes) and the oracle's streaming loop: token count, stop reason,
    RNG draw count, waveform."""
    import make_golden
    from oracle import generate_streaming as ogs
    from test_oracle_golden import _oracle_streaming_small
    rnd = random.Random(seed)
    runs = [(f"fuzz_z{k}.npz", rnd.randint(1, 23), rnd.randint(4, 70), 500 + seed * 100 + k, rnd.choice([-1.5, -1.0, -0.5, -0.2, 0.0, 0.1, 0.35]))
            for k in range(n)]
    only = [a for a in sys.argv if a.startswith("--only=")]
    if only:                                             # --only=K: the K-th configuration of this (n, seed) alone (to reproduce a failure)
        runs = [runs[int(only[0].split("=")[1])]]
    out = tempfile.mkdtemp(prefix="vv_fuzz_z_")
    make_golden.OUT_DIR = out
    make_golden.gen_generate_streaming(custom=runs)
    bad = 0
    for name, n_text, max_new, sd, eb in runs:
        z = np.load(os.path.join(out, name))
        om = _oracle_streaming_small(eos_bias=float(z["eos_bias"]))

        def cache(tag, lm):
            c = lm.new_cache()
            for li in range(int(z[f"{tag}_layers"])):
                c.k[li] = torch.from_numpy(z[f"{tag}_k{li}"]).clone()
                c.v[li] = torch.from_numpy(z[f"{tag}_v{li}"]).clone()
            c.length = c.k[0].shape[1]
            return c
        preset = ogs.Preset(cache("lm", om.lm), cache("tts", om.tts_lm), cache("neg_tts", om.tts_lm), torch.from_numpy(z["tts_last"]),
                            torch.from_numpy(z["neg_tts_last"]))
        draws = iter([torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))])
        n_tok, audio, reach, fin = ogs.oracle_generate_streaming(om, preset, torch.from_numpy(z["text"]), 1.5, 5, lambda frame, n2: next(draws).reshape(n2, 64),
                                                                 preset.tts_cache.length + int(z["max_new"]))
        ref = torch.from_numpy(z["audio"])
        got = audio.reshape(-1) if audio is not None else torch.zeros(0)
        ok = next(draws, None) is None and n_tok == int(z["n_tokens"]) and bool(reach) == bool(z["reach_max"][0]) and got.shape == ref.shape
        err = float((got - ref).norm() / ref.norm()) if ok and ref.numel() else (0.0 if ok else float("inf"))
        # ---- the product's streaming class on the oracle-backed CPU engine, as demo/streaming_inference_from_file.py drives it ----
        import copy
        import fake_engine
        from test_dropin_cpu import TOK, tiny_streaming_checkpoint
        from transformers.modeling_outputs import BaseModelOutputWithPast
        from vibevoice_amd import modeling_streaming
        import pathlib
        ckdir = pathlib.Path(tempfile.mkdtemp(prefix="vv_fuzz_zck_"))
        path = tiny_streaming_checkpoint(ckdir, float(z["eos_bias"]))
        pdraws = [torch.from_numpy(z[f"draw_{i}"]).reshape(2, 64) for i in range(int(z["n_draws"]))]

        def branch(tag):
            kv = [(torch.from_numpy(z[f"{tag}_k{li}"])[None], torch.from_numpy(z[f"{tag}_v{li}"])[None]) for li in range(int(z[f"{tag}_layers"]))]
            hid = torch.zeros(1, kv[0][0].shape[2], 128)
            hid[0, -1] = torch.from_numpy(z[f"{tag}_last"])
            return BaseModelOutputWithPast(last_hidden_state=hid, past_key_values=kv)
        pre_out = {"lm": branch("lm"), "tts_lm": branch("tts"), "neg_lm": None, "neg_tts_lm": branch("neg_tts")}

        class Patch:
            def setattr(self, obj, name_, val, raising=True):
                setattr(obj, name_, val)

            def setenv(self, k, v):
                os.environ[k] = v

            def delenv(self, k, raising=False):
                os.environ.pop(k, None)
        with fake_engine.cpu_cuda_shims(Patch()):
            modeling_streaming.Engine = fake_engine.LoadableFakeStreamingEngine
            model = modeling_streaming.VibeVoiceStreamingForConditionalGenerationInference.from_pretrained(path, torch_dtype=torch.float32, device_map="cuda")
            model.eval()
            model.set_ddpm_inference_steps(num_steps=5)
            prompt, text = torch.from_numpy(z["prompt"])[None], torch.from_numpy(z["text"])[None]
            lm_len = pre_out["lm"]["last_hidden_state"].size(1)
            o = model.generate(input_ids=torch.zeros(1, lm_len, dtype=torch.long), attention_mask=torch.ones(1, lm_len, dtype=torch.long),
                               tts_lm_input_ids=prompt, tts_lm_attention_mask=torch.ones_like(prompt), tts_text_ids=text,
                               speech_input_mask=torch.zeros(1, prompt.shape[1], dtype=torch.bool), speech_tensors=None, speech_masks=None,
                               max_new_tokens=int(z["max_new"]), cfg_scale=1.5, tokenizer=TOK, generation_config={"do_sample": False}, verbose=False,
                               all_prefilled_outputs=copy.deepcopy(pre_out), _noise_fn=lambda frame, n2: pdraws[frame])
        pgot = o.speech_outputs[0].reshape(-1) if (o.speech_outputs and o.speech_outputs[0] is not None) else torch.zeros(0)
        if os.environ.get("VV_FUZZ_VERBOSE"):
            print(f"   product: {int(o.sequences.shape[1])} tokens (reference {int(z['n_tokens'])}), reach_max {bool(o.reach_max_step_sample[0])} "
                  f"({bool(z['reach_max'][0])}), {pgot.numel()} samples ({ref.numel()})")
        pok = int(o.sequences.shape[1]) == int(z["n_tokens"]) and bool(o.reach_max_step_sample[0]) == bool(z["reach_max"][0]) and pgot.shape == ref.shape
        perr = float((pgot - ref).norm() / ref.norm()) if pok and ref.numel() else (0.0 if pok else float("inf"))
        good = ok and err <= 1e-4 and pok and perr <= 1e-4
        bad += 0 if good else 1
        print(f"{'ok  ' if good else 'FAIL'} {name:14s} text {n_text:2d}, cap {max_new:2d}, eos bias {eb:+.2f}: {int(z['n_tokens'])} tokens, {ref.numel() // 3200} frames, "
              f"reach_max {bool(z['reach_max'][0])}; oracle rel-L2 {err:.1e}, product streaming loop {perr:.1e}")
    print(f"{len(runs)} streaming runs: {bad} mismatches")
    return 1 if bad else 0


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--streaming" in sys.argv:
        return streaming_main(int(args[0]) if args else 8, int(args[1]) if len(args) > 1 else 7)
    n = int(args[0]) if args else 8
    seed = int(args[1]) if len(args) > 1 else 7
    norefresh = "--norefresh" in sys.argv
    kw = {"refresh_negative": False} if norefresh else {}
    rnd = random.Random(seed)
    runs = []
    for k in range(n):
        B = rnd.choice([1, 2, 2, 3, 4])
        plans = [[rnd.choice("DDDES") for _ in range(rnd.randint(3, 9))] + ["X"] for _ in range(B)]
        runs.append((f"fuzz_{k}.npz", B, plans, 1000 + seed * 100 + k, kw))
    runs += [("fuzz_eos0.npz", 2, [list("X"), list("DDEX")], 1900 + seed, kw),
             ("fuzz_ss.npz", 2, [list("DEESSDX"), list("SSDDX")], 1901 + seed, kw),
             ("fuzz_b3late.npz", 3, [list("DDDDX"), list("ESDDX"), list("SESDX")], 1902 + seed, kw)]
    sampled = "--sampled" in sys.argv
    if sampled:
        # free-running token SAMPLING (do_sample, plain multinomial over the valid ids: top_k=0), batches of 1-3: what is compared is the
        # product loop only, seeded like the reference's run -- every draw (prefill Gaussians, per-frame noise, one multinomial per
        # step over the whole batch) must come off the global generator in the reference's order for the tokens to agree at all
        runs = [(f"fuzz_s{k}.npz", rnd.choice([1, 2, 2, 3]), None, 3000 + seed * 100 + k,
                 {"max_new_tokens": 12, "do_sample": True}) for k in range(n)]
    mixed = "--mixed" in sys.argv
    if mixed:
        # everything at once, product loop only (seeded like the reference's run): batches of 1-4, forced plans or greedy free-running
        # decoding, optional max_new_tokens caps, voice samples that are not whole frames, the gradio demo's stochastic scheduler
        runs = []
        for k in range(n):
            B = rnd.choice([1, 2, 3, 4])
            plans = None if rnd.random() < 0.3 else [[rnd.choice("DDDDES") for _ in range(rnd.randint(2, 8))] + ["X"] for _ in range(B)]
            rkw = {"wav_len": rnd.choice([9600, 9600, 8000, 6500])}
            if plans is None or rnd.random() < 0.3:
                rkw["max_new_tokens"] = rnd.randint(3, 11)
            if rnd.random() < 0.3:
                rkw["sde"] = True
            if rnd.random() < 0.4:            # several voice samples per prompt (multi-speaker scripts); whole frames only here
                rkw["voices"] = [[rnd.randint(1, 3) for _ in range(rnd.choice([1, 2]) if b < 3 else 1)] for b in range(B)]
                rkw["wav_len"] = 9600
            runs.append((f"fuzz_m{k}.npz", B, plans, 5000 + seed * 100 + k, rkw))
        sampled = True                        # same comparison code path: product only
    warped = "--warped" in sys.argv
    if warped:
        # the full-vocabulary logits processors in front of the valid-token constraint (HF's list: repetition penalty, temperature,
        # top-k, top-p, min-p), random settings, batches of 1-2; product loop only, seeded.  A setting that filters EVERY valid id of
        # some step makes the reference fail in torch.multinomial (NaN probabilities): such runs are skipped.
        runs = []
        for k in range(n):
            gc = {"do_sample": rnd.random() < 0.8}
            if rnd.random() < 0.7:
                gc["repetition_penalty"] = round(rnd.uniform(1.0, 1.4), 3)
            if gc["do_sample"]:
                gc["top_k"] = rnd.choice([0, 0, 280, 300, 310, 315])
                if rnd.random() < 0.5:
                    gc["top_p"] = round(rnd.uniform(0.99, 0.9999), 4)
                if rnd.random() < 0.5:
                    gc["temperature"] = round(rnd.uniform(0.6, 1.6), 3)
                if rnd.random() < 0.4:
                    gc["min_p"] = round(rnd.uniform(1e-5, 2e-4), 6)
            runs.append((f"fuzz_w{k}.npz", rnd.choice([1, 2]), None, 7000 + seed * 100 + k, {"max_new_tokens": 10, "gen_cfg": gc}))
        sampled = True
    import make_golden
    out = tempfile.mkdtemp(prefix="vv_fuzz_")
    make_golden.OUT_DIR = out
    if warped:
        kept = []
        for r in runs:
            try:
                make_golden.gen_generate(custom=[r])
                kept.append(r)
            except RuntimeError as ex:
                if "probability tensor" not in str(ex):
                    raise
                print(f"skip {r[0]:16s} {r[4]['gen_cfg']}: every valid id filtered at some step (the reference fails in torch.multinomial)")
        runs = kept
    else:
        make_golden.gen_generate(custom=runs)

    # ---- the oracle loop ----
    from oracle import generate as ogen
    from test_oracle_golden import _oracle_small
    tok = ogen.TokenIds(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304, bos_token_id=None, pad_token_id=305)
    om = _oracle_small()
    # ---- the product loop on the oracle-backed CPU engine ----
    import fake_engine
    from safetensors.torch import save_file
    from test_dropin_cpu import TOK, tiny_reference_config, tiny_reference_state_dict
    from vibevoice_amd import modeling
    ck = tempfile.mkdtemp(prefix="vv_fuzz_ck_")
    open(os.path.join(ck, "config.json"), "w").write(json.dumps(tiny_reference_config()))
    save_file({k: v.contiguous() for k, v in tiny_reference_state_dict().items()}, os.path.join(ck, "model.safetensors"))

    class Patch:                                 # the two monkeypatch methods fake_engine.cpu_cuda_shims uses
        def setattr(self, obj, name, val, raising=True):
            setattr(obj, name, val)

        def setenv(self, k, v):
            os.environ[k] = v

        def delenv(self, k, raising=False):
            os.environ.pop(k, None)
    worst, bad_or, bad_pr, n_known = 0.0, 0, 0, 0
    with fake_engine.cpu_cuda_shims(Patch()):
        modeling.Engine = fake_engine.LoadableFakeEngine
        model = modeling.VibeVoiceForConditionalGenerationInference.from_pretrained(ck, torch_dtype=torch.float32, device_map="cuda")
        model.eval()
        model.set_ddpm_inference_steps(num_steps=5)
        for name, B, plans, sd, rkw in runs:
            z = np.load(os.path.join(out, name))
            if sampled:
                ids = torch.from_numpy(z["input_ids"])
                inputs = dict(speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                              speech_input_mask=torch.from_numpy(z["speech_input_mask"]))
                model.speculate_sampling = True
                if mixed:
                    sched = model.model.noise_scheduler
                    model.model.noise_scheduler = sched.from_config(sched.config, algorithm_type="sde-dpmsolver++" if rkw.get("sde") else "dpmsolver++",
                                                                    beta_schedule="squaredcos_cap_v2")
                    model.set_ddpm_inference_steps(num_steps=5)
                    gkw = dict(max_new_tokens=rkw.get("max_new_tokens"), generation_config={"do_sample": False})
                    if plans is not None:
                        sym = {"D": 303, "E": 302, "S": 301, "X": 304}
                        gkw["_forced_tokens"] = [[sym[t] for t in p] for p in plans]
                elif warped:
                    gkw = dict(max_new_tokens=10, generation_config=dict(rkw["gen_cfg"]))
                else:
                    gkw = dict(max_new_tokens=12, generation_config={"do_sample": True, "top_k": 0})
                torch.manual_seed(int(z["seed"]))
                o = model.generate(input_ids=ids, attention_mask=torch.from_numpy(z["attention_mask"]), cfg_scale=1.3, tokenizer=TOK,
                                   verbose=False, is_prefill=True, show_progress_bar=False, **inputs, **gkw)
                okp = torch.equal(o.sequences.cpu(), torch.from_numpy(z["sequences"])) and torch.equal(o.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
                e = []
                for b in range(B):
                    ref = torch.from_numpy(z[f"audio_{b}"])
                    got = o.speech_outputs[b].reshape(-1) if o.speech_outputs[b] is not None else torch.zeros(0)
                    e.append(float((got - ref).norm() / ref.norm()) if got.shape == ref.shape and ref.numel() else (0.0 if got.shape == ref.shape else float("inf")))
                good = okp and max(e) <= 1e-4
                bad_pr += 0 if good else 1
                toks = z["sequences"][:, ids.shape[1]:].tolist()
                # rounds 1-4 excused two patterns here (the single-entry negative correction, the partial last voice frame); since round 5
                # the product follows the reference on both (vv_kv_move, vv_acoustic_encode_ragged): every mismatch is a failure
                known = False
                n_known += 1 if (plans is not None and hits_single_entry_pattern(plans, True)) or (rkw.get("wav_len", 9600) if mixed else 9600) % 3200 else 0
                if not good:
                    pass
                worst = max(worst, max(e))
                what = (f"{'forced ' + str([''.join(p) for p in plans]) if plans is not None else 'free-running'} {rkw}" if mixed else
                        (f"{rkw['gen_cfg']} tokens {toks}" if warped else f"sampled tokens {toks}"))
                print(f"{'ok  ' if good else ('KNWN' if known else 'FAIL')} {name:16s} B={B} {what}  product loop rel-L2 {max(e):.1e}")
                continue
            ids = torch.from_numpy(z["input_ids"])
            draws = [torch.from_numpy(z[f"draw_{i}"]) for i in range(int(z["n_draws"]))]
            pre = (draws[0].reshape(B), draws[1].reshape(B, 3, 64))
            it = iter(draws[2:])
            forced = [z["forced"][b][:int(z["forced_len"][b])].tolist() for b in range(B)]
            inputs = dict(speech_tensors=torch.from_numpy(z["speech_tensors"]), speech_masks=torch.from_numpy(z["speech_masks"]),
                          speech_input_mask=torch.from_numpy(z["speech_input_mask"]))
            seq, audio, reach = ogen.oracle_generate(om, tok, ids, torch.from_numpy(z["attention_mask"]), inputs["speech_tensors"], inputs["speech_masks"],
                                                     inputs["speech_input_mask"], cfg_scale=1.3, num_steps=5, noise_fn=lambda step, n2: next(it).reshape(n2, 64),
                                                     prefill_noise=pre, forced_tokens=forced, refresh_negative=not norefresh)
            ok = torch.equal(seq, torch.from_numpy(z["sequences"])) and torch.equal(reach, torch.from_numpy(z["reach_max"])) and next(it, None) is None
            def werr(outs):
                e = []
                for b in range(B):
                    ref = torch.from_numpy(z[f"audio_{b}"])
                    got = outs[b].reshape(-1) if outs[b] is not None else torch.zeros(0)
                    e.append(float((got - ref).norm() / ref.norm()) if got.shape == ref.shape and ref.numel() else (0.0 if got.shape == ref.shape else float("inf")))
                return max(e)
            w_or = werr(audio)
            ok_or = ok and w_or <= 1e-4
            ok_pr, w_pr = True, 0.0
            for spec in (False, True):
                model.speculate_sampling = spec
                torch.manual_seed(int(z["seed"]))
                o = model.generate(input_ids=ids, attention_mask=torch.from_numpy(z["attention_mask"]), max_new_tokens=None, cfg_scale=1.3, tokenizer=TOK,
                                   generation_config={"do_sample": False}, verbose=False, is_prefill=True, _forced_tokens=forced, **inputs, **kw)
                ok_pr = ok_pr and torch.equal(o.sequences.cpu(), torch.from_numpy(z["sequences"])) and torch.equal(o.reach_max_step_sample.cpu(), torch.from_numpy(z["reach_max"]))
                w_pr = max(w_pr, werr(o.speech_outputs))
            ok_pr = ok_pr and w_pr <= 1e-4
            # the single-entry negative correction (a row holding exactly one valid negative entry emits a non-diffusion token while another
            # row diffuses: the reference keeps THIS step's entry, :603 vs :613) was an excused deviation until round 4; the product follows
            # it now (vv_kv_move), so such plans are only COUNTED here -- a mismatch on them fails like any other
            known = False
            n_known += 1 if hits_single_entry_pattern(plans, not norefresh) else 0
            worst = max(worst, w_or, 0.0 if known else w_pr)
            bad_or += 0 if ok_or else 1
            bad_pr += 0 if (ok_pr or known) else 1
            tag = "ok  " if (ok_or and ok_pr) else ("KNWN" if (ok_or and known) else "FAIL")
            print(f"{tag} {name:16s} {[''.join(p) for p in plans]}  oracle rel-L2 {w_or:.1e}, product loop (speculation off / on) {w_pr:.1e}")
    print(f"{len(runs)} plans{' (refresh_negative=False)' if norefresh else ''}: oracle mismatches {bad_or}, product-loop mismatches {bad_pr} "
          f"({n_known} of the plans exercise a formerly excused pattern -- single-entry correction / partial last voice frame -- and are held to the same bound), worst rel-L2 {worst:.1e}")
    return 1 if (bad_or or bad_pr) else 0


if __name__ == "__main__":
    raise SystemExit(main())
