#!/usr/bin/env python
"""bench.py -- audio-seconds per wall-second of the VibeVoice hot loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model 1.5b|7b] [--solver-steps N]

One "step" = one iteration of the reference's hot loop emitting <speech_diffusion>
(modeling_vibevoice_inference.py:432-675): positive + CFG-negative LM decode,
restricted lm_head, N-step CFG DPM-Solver++ diffusion head, acoustic codec decode
of the 3200-sample frame, semantic re-encode, connectors, token read-back.
Workload at N=1 (BASELINE.json configs[1]): VibeVoice-1.5B shapes, 1 speaker,
prompt sized like demo/text_examples/1p_abs.txt (~220 text tokens + a 75-frame
voice prompt), bf16 weights, 10 solver steps (the file demo's default,
demo/inference_from_file.py:365), synthetic seeded weights and forced token
schedule.  With --gpus N>1 every rank decodes its own utterance
(weak scaling, no collective inside the step loop); weights are generated on rank 0
and broadcast over RCCL/xGMI at start-up.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_SEC = 3200 / 24000.0
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_frame(cfg, n_solver, kv_len_pos, kv_len_neg):
    """bf16 weights read once per frame, head weights once per solver step
    (cond_proj once per frame), cond+uncond LM rows share one weight read, lm_head = 5 rows."""
    from vibevoice_amd.synthetic import param_shapes
    sh = param_shapes(cfg)
    H = cfg["decoder_config"]["hidden_size"]

    def count(prefix, pred=lambda k: True):
        n = 0
        for k, s in sh.items():
            if k.startswith(prefix) and pred(k):
                m = 1
                for d in s:
                    m *= d
                n += m
        return n
    p_lm = count("model.language_model.layers.") + count("model.language_model.norm.")
    p_head = count("model.prediction_head.")
    p_dec = count("model.acoustic_tokenizer.decoder.")
    p_sem = count("model.semantic_tokenizer.encoder.")
    p_conn = count("model.acoustic_connector.") + count("model.semantic_connector.")
    d = cfg["decoder_config"]
    kv_tok = 2 * d["num_hidden_layers"] * d["num_key_value_heads"] * (H // d["num_attention_heads"]) * 2
    b = 2 * p_lm + 2 * (p_head - H * H) * n_solver + 2 * H * H + 2 * p_dec + 2 * p_sem + 2 * p_conn
    b += 2 * 5 * H + kv_tok * (kv_len_pos + kv_len_neg)
    return float(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=150)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="1.5b")
    ap.add_argument("--solver-steps", type=int, default=10)
    ap.add_argument("--cfg-scale", type=float, default=1.3)
    ap.add_argument("--xsplit", type=int, default=int(os.environ.get("VVHIP_XSPLIT", "1")))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the per-launch hipEvent pass (for rocprof runs)")
    ap.add_argument("--cpu-frames", type=int, default=4)
    ap.add_argument("--text-tokens", type=int, default=220)
    ap.add_argument("--voice-frames", type=int, default=75)
    ap.add_argument("--speakers", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1, help="utterances decoded together on each GPU (1 = the BASELINE config; "
                    "up to 8 share every LM / diffusion-head weight pass)")
    ap.add_argument("--max-ctx", type=int, default=0)
    ap.add_argument("--enc-frames", type=int, default=5, help="voice-prompt frames per acoustic-encoder pass")
    ap.add_argument("--prefill-rows", type=int, default=512, help="prompt rows per LM weight pass (engine max_rows)")
    ap.add_argument("--kv-start", type=int, default=0,
                    help="pretend the positive KV cache already holds this many tokens after the prefill "
                         "(long-context decode measurement; the extra entries are zeros)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # torchrun, even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from vibevoice_amd import build as vbuild
    if rank == 0 and vbuild.stale():
        vbuild.build()
    if use_dist:
        dist.barrier()
    from vibevoice_amd import parallel, synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.engine import Engine, map_param_name
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference, engine_config_from_reference

    cfg = CONFIGS[args.model]
    if "streaming" in args.model:
        return bench_streaming(args, cfg, rank, world, device)
    K, W, NS = args.steps, max(1, args.warmup), args.solver_steps
    B = max(1, min(8, args.batch))
    inputs = synthetic.synthetic_inputs(cfg, n_speakers=args.speakers, text_tokens=args.text_tokens,
                                        voice_frames=args.voice_frames, seed=100 + rank, batch=B)
    L0 = inputs["input_ids"].shape[1]
    total_steps = W + K + 2
    max_ctx = args.max_ctx or ((max(L0, args.kv_start) + total_steps + 256 + 127) // 128 * 128)
    ecfg = engine_config_from_reference(cfg, n_slots=B, max_ctx=max_ctx, xsplit=args.xsplit,
                                        use_graph=not args.no_graph, enc_frames=args.enc_frames, max_rows=max(2 * B, args.prefill_rows))
    t_load0 = time.time()
    eng = Engine(ecfg, device)
    exp = eng.expected_weights()
    keep_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    cpu_sd = {}
    # rank 0 draws the weights, RCCL broadcasts them over xGMI (one collective per tensor at start-up, none later)
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    make = lambda k, shape: synthetic.random_tensor(k, shape, gen, device, torch.bfloat16)
    for k, t in parallel.broadcast_params(synthetic.param_shapes(cfg).items(), make, device, torch.bfloat16):
        name = map_param_name(k)
        if name in exp:
            eng.upload(name, t)
        if keep_cpu:
            cpu_sd[k] = t.to("cpu")
        del t
    miss = eng.missing_weights()
    if miss:
        raise SystemExit(f"engine parameters not provided: {miss[:5]}")
    model = VibeVoiceForConditionalGenerationInference(cfg, eng, model_dtype=torch.bfloat16)
    model.set_speech_factors(0.2, -0.05)
    model.set_ddpm_inference_steps(NS)
    load_s = time.time() - t_load0

    forced = [synthetic.forced_schedule(total_steps, turn=150) for _ in range(B)]
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    noise_bank = torch.randn(total_steps + 1, 2 * B, cfg["acoustic_vae_dim"], generator=g, device=device)
    torch.cuda.synchronize()

    marks = {}

    def step_cb(step):
        if step == W or step == W + K:
            eng.sync()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
                torch.cuda.synchronize()
            marks[step] = time.perf_counter()
        if step == 1:
            eng.sync()
            marks["prefill_done"] = time.perf_counter()

    t_gen0 = time.perf_counter()
    out = model.generate(tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale, generation_config={"do_sample": False},
                         max_new_tokens=total_steps, show_progress_bar=False, _forced_tokens=forced,
                         _noise_fn=lambda step, n2: noise_bank[step], _step_callback=step_cb,
                         _kv_start=args.kv_start, **inputs)
    eng.sync()
    t_gen1 = time.perf_counter()
    wall = marks[W + K] - marks[W]
    # steps W..W+K-1: count the <speech_diffusion> frames among them (the schedule inserts 2 control tokens per 150)
    frames = B * sum(1 for t in forced[0][W:W + K] if t == synthetic.TOKENS.speech_diffusion_id)
    frames_all, wall_max = parallel.aggregate_throughput(frames, wall, device)   # sum over ranks / max over ranks
    value = frames_all * FRAME_SEC / wall_max
    audio_total = out.speech_outputs[0].shape[-1] / 24000.0

    if os.environ.get("VVHIP_TIMELINE") and rank == 0:      # timing builds only: tools/step_timeline.py
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import step_timeline
        step_timeline.dump(eng, os.environ["VVHIP_TIMELINE"])

    # ---- roofline of the dominant kernel (vv_gemm_kernel): per-launch hipEvents over K_prof live steps ----
    roof = None
    if rank == 0 and not args.no_roofline:
        kprof = 8
        forced_p = [synthetic.forced_schedule(kprof + 3, turn=150) for _ in range(B)]
        prof = {}

        def prof_cb(step):
            if step == 2:
                eng.sync()
                eng.profile_begin()
            if step == 2 + kprof:
                prof["res"] = eng.profile_end()
        inp2 = synthetic.synthetic_inputs(cfg, n_speakers=args.speakers, text_tokens=args.text_tokens,
                                          voice_frames=args.voice_frames, seed=100, batch=B)
        model.generate(tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale, generation_config={"do_sample": False},
                       max_new_tokens=kprof + 3, show_progress_bar=False, _forced_tokens=forced_p,
                       _noise_fn=lambda step, n2: noise_bank[step], _step_callback=prof_cb, **inp2)
        (n_l, ms_cal, by), (n_o, ms_o, by_o) = prof["res"]       # [decode GEMV kernel], [general GEMM kernel]
        # launch duration = hipEvent pair around each launch minus the in-stream cost of an empty pair (vv_profile_end);
        # cross-check against rocprofv3's per-kernel average: profiles/r01_1p5b_gemv_summary.txt
        ms = ms_cal
        ms_raw = eng.stat(2) / 1e6
        ach = by / 1e9 / (ms / 1e3) if ms > 0 else 0.0
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(args.model, {}).get("hbm_bytes_per_launch")
        except Exception:
            pass
        rp = None
        try:
            with open(os.path.join(ROOT, "profiles", "rocprof_gemv.json")) as f:
                rp = json.load(f).get(args.model, {}).get("avg_launch_us")
        except Exception:
            pass
        formula = algorithmic_bytes_per_frame(cfg, NS, max(L0, args.kv_start) + W + K // 2, 1 + min(150, K) // 2)
        roof = {"bound": "hbm", "kernel": "vv_gemv_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "launches_per_step": round(n_l / kprof, 1), "avg_launch_us": round(ms * 1e3 / max(1, n_l), 3),
                "avg_launch_us_raw_event_pair": round(ms_raw * 1e3 / max(1, n_l), 3), "empty_event_pair_us": round(eng.stat(3) / 1e3, 3),
                # cross-check from the committed rocprofv3 summary of this workload (graph replay: per-kernel intervals overlap,
                # an upper bound on the launch duration -> a lower bound on the fraction); profiles/rocprof_gemv.json
                "rocprof_avg_launch_us": rp, "frac_at_rocprof_duration": round(by / max(1, n_l) / 1e3 / rp / HBM_PEAK_GBS, 4) if rp else None,
                "bytes_per_launch": round(by / max(1, n_l), 1),
                "gemv_bytes_per_step": round(by / kprof, 1),
                "other_gemm": {"kernel": "vv_gemm_kernel", "launches_per_step": round(n_o / kprof, 1),
                               "bytes_per_step": round(by_o / kprof, 1), "GBps": round(by_o / 1e9 / (ms_o / 1e3), 1) if ms_o > 0 else None},
                "formula_bytes_per_step": round(formula, 1),
                "whole_step_GBps": round(formula / 1e9 / (wall_max / K), 1),
                "whole_step_frac": round(formula / 1e9 / (wall_max / K) / HBM_PEAK_GBS, 4)}

    # ---- CPU baseline: the oracle loop on the host cores, bounded sample ----
    cpu = None
    if keep_cpu:
        try:
            cpu = cpu_baseline(cfg, cpu_sd, NS, args.cfg_scale, args.cpu_frames)
        except Exception as ex:   # the baseline is a reported number, never the product path
            cpu = {"value": None, "error": repr(ex)[:200]}
    if rank == 0:
        res = {
            "metric": "audio-sec/wall-sec", "value": round(value, 3), "unit": "audio-s/wall-s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(wall_max / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"VibeVoice-{args.model.upper()} shapes, {args.speakers} speaker, "
                                   f"{L0}-token prompt ({args.text_tokens} text + {args.voice_frames}-frame voice), "
                                   f"{NS} solver steps, cfg {args.cfg_scale}, {B} utterance{'s' if B > 1 else ''} per GPU, forced token schedule",
                       "model": f"VibeVoice-{args.model}", "solver_steps": NS, "prompt_tokens": L0,
                       "xsplit": args.xsplit, "hipgraph": not args.no_graph, "kv_start": args.kv_start, "parallelism": f"utterance-dp{world}"},
            "roofline": roof, "cpu_baseline": cpu,
            "extra": {"frames_timed": frames, "weights_load_s": round(load_s, 2),
                      "prefill_plus_first_frame_s": round(marks.get("prefill_done", t_gen0) - t_gen0, 4),
                      "prefill_phases": getattr(model, "last_prefill", None),
                      "utterance_audio_s": round(audio_total, 2), "utterance_wall_s": round(t_gen1 - t_gen0, 3),
                      "utterance_audio_per_wall": round(audio_total / (t_gen1 - t_gen0), 2)},
        }
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


def bench_streaming(args, cfg, rank, world, device):
    """BASELINE.json configs[4]: Streaming-0.5B, hipGraph-captured decode+diffusion step, p50 first-audio latency.
    Synthetic weights and an Emma-shaped synthetic preset (lm 74 / tts_lm 251 cached positions)."""
    import statistics
    import types
    from vibevoice_amd import synthetic
    from vibevoice_amd.modeling_streaming import VibeVoiceStreamingForConditionalGenerationInference
    NS = args.solver_steps if args.solver_steps != 10 else 5          # the streaming demo default is 5
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    sd = ((k, synthetic.random_tensor(k, shp, gen, device, torch.bfloat16))
          for k, shp in synthetic.streaming_param_shapes(cfg).items())
    model = VibeVoiceStreamingForConditionalGenerationInference.from_state_dict(
        cfg, sd, torch.bfloat16, device, xsplit=args.xsplit, use_graph=not args.no_graph, max_ctx=2048, n_slots=2)
    model.set_speech_factors(0.2, -0.05)
    model.set_ddpm_inference_steps(NS)
    model.engine.upload("eos.fc2.bias", torch.tensor([-30.0]))       # random weights: keep the EOS head from firing
    d = cfg["decoder_config"]
    H, kvh, hd = d["hidden_size"], d["num_key_value_heads"], d["hidden_size"] // d["num_attention_heads"]
    n_tts = cfg["tts_backbone_num_hidden_layers"]
    n_lm = d["num_hidden_layers"] - n_tts

    def branch(n_layers, L):
        kv = [(torch.randn(1, kvh, L, hd, device=device, dtype=torch.bfloat16) * 0.5,
               torch.randn(1, kvh, L, hd, device=device, dtype=torch.bfloat16) * 0.5) for _ in range(n_layers)]
        return types.SimpleNamespace(past_key_values=kv, last_hidden_state=torch.randn(1, L, H, device=device))
    preset = {"lm": branch(n_lm, 74), "tts_lm": branch(n_tts, 251), "neg_lm": None, "neg_tts_lm": branch(n_tts, 1)}
    T = synthetic.TOKENS
    tok = types.SimpleNamespace(convert_tokens_to_ids=lambda s: T.pad_token_id)
    g = torch.Generator().manual_seed(3)
    lat = []
    for trial in range(3 + 30):
        text = torch.randint(0, 151000, (1, 5), generator=g)
        marks = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(tts_text_ids=text, all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok,
                       max_new_tokens=5 + 6, _marks=marks)
        if trial >= 3:
            lat.append((marks["first_audio"] - t0) * 1e3)
    # steady state: K frames
    K = args.steps
    n_text = ((K + 5) // 6) * 5
    text = torch.randint(0, 151000, (1, n_text), generator=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(tts_text_ids=text, all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok,
                         max_new_tokens=n_text + (n_text // 5) * 6)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    audio_s = out.speech_outputs[0].shape[-1] / 24000.0
    res = {"metric": "audio-sec/wall-sec", "value": round(audio_s / wall, 3), "unit": "audio-s/wall-s", "n_gpus": 1,
           "steps": int(audio_s / FRAME_SEC + 0.5), "warmup": 33, "ms_per_step": round(wall / (audio_s / FRAME_SEC) * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"VibeVoice-Streaming-0.5B shapes, Emma-shaped synthetic preset (lm 74 / tts 251), {NS} solver steps, "
                                  "whole generate() incl. preset import, text windows of 5 / speech windows of 6",
                      "model": "VibeVoice-Streaming-0.5B", "solver_steps": NS, "hipgraph": not args.no_graph},
           "roofline": None, "cpu_baseline": None,
           "extra": {"p50_first_audio_ms": round(statistics.median(lat), 3), "p90_first_audio_ms": round(sorted(lat)[int(0.9 * len(lat))], 3),
                     "trials": len(lat), "finished_by_eos_or_cap": True}}
    print(json.dumps(res), flush=True)
    model.engine.close()


def cpu_baseline(cfg, cpu_sd, n_solver, cfg_scale, n_frames):
    """Times oracle/ (the CPU restatement of the reference loop) on this host: `n_frames` decode
    frames after a short prompt, fp32, all host threads.  kind = "port"."""
    from oracle import generate as ogen
    from oracle import lm as olm
    from vibevoice_amd import synthetic
    # the GPU box advertises hundreds of logical CPUs but the job may be cgroup-limited; a modest
    # thread count keeps torch's intra-op pool from thrashing (256 threads measured 200 s/frame)
    ncpu = min(int(os.environ.get("VVHIP_CPU_THREADS", "16")), os.cpu_count() or 1)
    torch.set_num_threads(ncpu)
    t_budget = float(os.environ.get("VVHIP_CPU_BUDGET_S", "45"))
    t_start = time.perf_counter()
    d = cfg["decoder_config"]
    H = d["hidden_size"]

    def sub(prefix):
        return {k[len(prefix):]: v.float() for k, v in cpu_sd.items() if k.startswith(prefix)}
    lm_w = sub("model.language_model.")
    lm = olm.Qwen2Oracle(lm_w, d["num_hidden_layers"], d["num_attention_heads"], d["num_key_value_heads"],
                         H // d["num_attention_heads"], d.get("rope_theta", 1e6), d.get("rms_norm_eps", 1e-6))
    depths = [int(x) for x in cfg["acoustic_tokenizer_config"]["encoder_depths"].split("-")]
    m = ogen.OracleModel(
        lm=lm, lm_head=cpu_sd["lm_head.weight"].float() if "lm_head.weight" in cpu_sd else lm_w["embed_tokens.weight"],
        head_w=sub("model.prediction_head."), head_layers=cfg["diffusion_head_config"]["head_layers"],
        ac_w=sub("model.acoustic_tokenizer."), sem_w=sub("model.semantic_tokenizer."),
        ac_conn=sub("model.acoustic_connector."), sem_conn=sub("model.semantic_connector."),
        ratios=cfg["acoustic_tokenizer_config"]["encoder_ratios"], enc_depths=depths,
        dec_depths=list(reversed(depths)), sem_depths=depths, scaling=0.2, bias=-0.05,
        max_position_embeddings=d["max_position_embeddings"])
    T = synthetic.TOKENS
    tok = ogen.TokenIds(T.speech_start_id, T.speech_end_id, T.speech_diffusion_id, T.eos_token_id, None, T.pad_token_id)
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, 151000, (1, 48), generator=g)
    ids[0, -1] = T.speech_start_id
    stamps = []

    class _Budget(Exception):
        pass

    def noise_fn(step, n2):
        stamps.append(time.perf_counter())
        if len(stamps) >= 2 and stamps[-1] - t_start > t_budget:
            raise _Budget()
        return torch.randn(n2, 64, generator=g)
    forced = [[T.speech_diffusion_id] * (n_frames + 1)]
    try:
        with torch.no_grad():
            ogen.oracle_generate(m, tok, ids, torch.ones_like(ids), cfg_scale=cfg_scale, num_steps=n_solver,
                                 max_new_tokens=n_frames + 1, noise_fn=noise_fn, forced_tokens=forced)
    except _Budget:
        pass
    per_frame = (stamps[-1] - stamps[0]) / max(1, len(stamps) - 1)
    return {"value": round(FRAME_SEC / per_frame, 4), "unit": "audio-s/wall-s", "cores": ncpu, "kind": "port",
            "sample": f"{len(stamps) - 1} decode frames after a 48-token text-only prompt, same model shapes/weights, "
                      f"fp32, {n_solver} solver steps, oracle loop (CPU restatement of the reference)",
            "ms_per_step": round(per_frame * 1e3, 2)}


if __name__ == "__main__":
    main()
