#!/usr/bin/env python
"""bench.py -- audio-seconds per wall-second of the VibeVoice hot loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload north-star|1p5b|streaming] [--model ...]

One "step" = one iteration of the reference's hot loop emitting <speech_diffusion>
(modeling_vibevoice_inference.py:432-675): positive + CFG-negative LM decode,
restricted lm_head, N-step CFG DPM-Solver++ diffusion head, acoustic codec decode
of the 3200-sample frame, semantic re-encode, connectors, token read-back.

Default workload = the north-star configuration, BASELINE.json configs[2]: VibeVoice-7B shapes, 2 speakers, a
10,922-token script prompt PREFILLED THROUGH THE ENGINE (MFMA tile GEMM + prefill attention), the full 20-step diffusion
schedule, decode timed at a KV length of ~32K (the positions between the end of the prompt and the measurement point hold
random bf16 K/V written with vv_kv_import_at -- stepping 21 K frames to get there would take minutes; the prompt part of
the cache is the real prefill output), bf16 weights, synthetic seeded weights and forced token schedule.
The same JSON line carries, under extra.configs, BASELINE configs[1] (VibeVoice-1.5B, 1 speaker, 1p_abs.txt-sized prompt,
10 solver steps -- the round-1 headline) and configs[4] (Streaming-0.5B, p50 first-audio latency), measured in the same
run with the same code.

With --gpus N>1 every rank decodes its own utterance (weak scaling, no collective inside the step loop); weights are
generated on rank 0 and broadcast as one packed blob over RCCL/xGMI at start-up.  `python bench.py --gpus N` without
torchrun spawns the N ranks itself (torch.distributed.run, 127.0.0.1 rendezvous).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_SEC = 3200 / 24000.0
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

# BASELINE.json configs -> bench parameters
WORKLOADS = {
    # configs[2]: 7B, 2 speakers, 32K-token script (L0 = 10,922 prompt tokens -> cap 2*L0 generated -> 32,766), N = 20
    "north-star": dict(model="7b", speakers=2, text_tokens=10731, voice_frames=75, solver_steps=20, kv_target=32000,
                       prefill_rows=11264, baseline_config="configs[2]"),
    # configs[1]: 1.5B, 1 speaker, prompt sized like demo/text_examples/1p_abs.txt, N = 10 (the file demo's default)
    "1p5b": dict(model="1.5b", speakers=1, text_tokens=220, voice_frames=75, solver_steps=10, kv_target=0,
                 prefill_rows=512, baseline_config="configs[1]"),
    "streaming": dict(model="0.5b-streaming", baseline_config="configs[4]"),
}


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="north-star", choices=sorted(WORKLOADS))
    ap.add_argument("--model", default=None, help="override the workload's model (1.5b | 7b | 0.5b-streaming)")
    ap.add_argument("--solver-steps", type=int, default=None)
    ap.add_argument("--text-tokens", type=int, default=None)
    ap.add_argument("--voice-frames", type=int, default=None)
    ap.add_argument("--speakers", type=int, default=None)
    ap.add_argument("--kv-start", type=int, default=None,
                    help="KV length at which decode is measured; positions past the prefilled prompt hold random K/V")
    ap.add_argument("--prefill-rows", type=int, default=None, help="prompt rows per LM weight pass (engine max_rows)")
    ap.add_argument("--cfg-scale", type=float, default=1.3)
    ap.add_argument("--xsplit", type=int, default=int(os.environ.get("VVHIP_XSPLIT", "1")))
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the PyTorch-ROCm eager (oracle loop on the GPU, bf16) leg")
    ap.add_argument("--no-parity", action="store_true", help="skip the full-depth parity leg (HIP engine vs the oracle legs, teacher-forced)")
    ap.add_argument("--no-parity-long", action="store_true", help="skip the parity leg through the prompt pass at the timed prompt length")
    ap.add_argument("--no-roofline", action="store_true", help="skip the GEMV launch-duration passes (for rocprof --pmc runs)")
    ap.add_argument("--skip-extra", action="store_true", help="main workload only (no extra.configs lines)")
    ap.add_argument("--no-config3", action="store_true", help="skip the configs[3] per-GPU leg (7B, 4 speakers, 8 utterances at 32K on each GPU)")
    ap.add_argument("--cpu-frames", type=int, default=3)
    ap.add_argument("--cpu-windows", action="store_true", help="SURVEY 8(d)'s CPU baseline as specified, nothing else: --cpu-frames decode frames "
                                                               "of the oracle loop at three KV lengths (minutes of host time)")
    ap.add_argument("--batch", type=int, default=1, help="utterances decoded together on each GPU (up to 8 share every LM / "
                    "diffusion-head weight pass)")
    ap.add_argument("--continuous", type=int, default=0, help="queue this many utterances through generate_continuous() "
                    "(slots = --batch) instead of one synchronous batch")
    ap.add_argument("--lanes", type=int, default=1, help="--continuous: split the queue over this many engine contexts sharing ONE weight upload "
                                                         "(generate_interleaved: one host thread + stream per lane)")
    ap.add_argument("--host-delay-us", type=float, default=0.0, help="diagnostic: busy-wait this long on the HOST at the top of every step; the "
                    "largest delay that leaves ms_per_step unchanged is the host's slack per step (how far off the critical path it is)")
    ap.add_argument("--full-utterance", action="store_true", help="one whole utterance of the workload, KV really growing; the metric over the "
                    "wall clock of the whole generate() (prefill and voice encode included)")
    ap.add_argument("--utterance-frames", type=int, default=0, help="--full-utterance: cap the generated tokens (0: the reference's own cap, 2 x prompt)")
    ap.add_argument("--max-ctx", type=int, default=0)
    ap.add_argument("--enc-frames", type=int, default=75, help="voice-prompt frames per acoustic-encoder pass (the engine default)")
    return ap.parse_args(argv)


CHECKPOINT_NAMES = {"1.5b": ("VibeVoice-1.5B", "vibevoice-1.5b", "1.5b"),
                    "7b": ("VibeVoice-7B", "VibeVoice-Large", "vibevoice-7b", "7b"),
                    "0.5b-streaming": ("VibeVoice-Realtime-0.5B", "VibeVoice-Streaming-0.5B", "vibevoice-streaming-0.5b", "0.5b-streaming")}


def find_checkpoint(model_key, root=None):
    """"if $VIBEVOICE_MODEL_DIR holds real checkpoints they are used instead" of the seeded synthetic weights.
    A checkpoint is a directory with config.json + *.safetensors (what the reference's converter writes and from_pretrained
    reads): $VIBEVOICE_MODEL_DIR/<name> for the names the released models go by, or $VIBEVOICE_MODEL_DIR itself when its
    config.json has this model's decoder width and depth.  Returns the directory or None."""
    from vibevoice_amd.configs import CONFIGS
    root = root if root is not None else os.environ.get("VIBEVOICE_MODEL_DIR")
    if not root or not os.path.isdir(root):
        return None

    def is_ckpt(d):
        return os.path.isfile(os.path.join(d, "config.json")) and any(f.endswith(".safetensors") for f in os.listdir(d))
    for name in CHECKPOINT_NAMES.get(model_key, ()):
        d = os.path.join(root, name)
        if os.path.isdir(d) and is_ckpt(d):
            return d
    if is_ckpt(root):
        try:
            with open(os.path.join(root, "config.json")) as f:
                dc = json.load(f)["decoder_config"]
            want = CONFIGS[model_key]["decoder_config"]
            if all(dc.get(k) == want.get(k) for k in ("hidden_size", "num_hidden_layers", "num_attention_heads")):
                return root
        except Exception:
            return None
    return None


def checkpoint_tensors(path):
    """(key, tensor) over every *.safetensors shard of a checkpoint directory, one tensor in memory at a time"""
    from safetensors import safe_open
    for fn in sorted(f for f in os.listdir(path) if f.endswith(".safetensors")):
        with safe_open(os.path.join(path, fn), framework="pt", device="cpu") as sf:
            for k in sf.keys():
                yield k, sf.get_tensor(k)


def pmc_traffic(model_key, kernel, bytes_per_launch=None):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE in
    its own run, x2 on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -- only when that pass ran on the binary
    that is loaded now (the file records the library's build id); otherwise null, with the reason."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            doc = json.load(f)
        ent = doc.get(model_key, {})
        fam = (ent.get("kernels") or {}).get(kernel)
        if fam is None:
            return None, f"profiles/pmc_traffic.json has no {kernel} entry for {model_key}"
        if ent.get("libvvhip_build_id") != _build_id():
            return None, (f"profiles/pmc_traffic.json was measured on build {ent.get('libvvhip_build_id')}, this run loads {_build_id()}: "
                          "not carried over (re-run tools/pmc_refresh.sh)")
        alg = fam.get("algorithmic_bytes_per_launch")
        if bytes_per_launch and alg and abs(bytes_per_launch - alg) > 0.05 * alg:
            return None, (f"profiles/pmc_traffic.json holds a pass of another launch geometry for {kernel} ({alg:.0f} algorithmic bytes per launch "
                          f"there, {bytes_per_launch:.0f} here): not comparable")
        return fam["hbm_bytes_per_launch"], ("profiles/pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE (own pass, x2 gfx950 correction) on this "
                                             f"build ({ent.get('libvvhip_build_id')}); algorithmic bytes of that pass {fam.get('algorithmic_bytes_per_launch')}")
    except Exception as ex:
        return None, f"profiles/pmc_traffic.json unreadable: {ex!r}"


def first_audio_trials(fn, n_trials):
    """first-audio latency = generate() entry -> the first chunk an AudioStreamer hands to a consumer on the host.
    fn(streamer) runs one generate() with that streamer; a consumer thread blocks on sample 0's stream and stamps the arrival
    of the first chunk.  Returns the latencies in ms."""
    import threading
    from vibevoice_amd.streamer import AudioStreamer
    out = []
    for _ in range(n_trials):
        st = AudioStreamer(batch_size=1)
        got = {}

        def consume():
            for _chunk in st.get_stream(0):
                got.setdefault("t", time.perf_counter())
        th = threading.Thread(target=consume, daemon=True)
        th.start()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(st)
        th.join(timeout=30)
        st.close()
        if "t" in got:
            out.append((got["t"] - t0) * 1e3)
    return out


def _build_id():
    """sha256[:16] of the kernel sources the loaded libvvhip.so was compiled from (vv_build_id)."""
    try:
        from vibevoice_amd import _lib
        return _lib.load().vv_build_id().decode().split("=", 1)[1]
    except Exception:
        return None


def respawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks (one process per GPU, RCCL rendezvous on 127.0.0.1)."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and os.environ.get("VVHIP_BENCH_SHARED_GPU") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(respawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the measured path)")
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): reporting n_gpus={world}", file=sys.stderr)
    # VVHIP_BENCH_SHARED_GPU=1: harness check of the N > 1 control flow on a ONE-GPU box (spawn, packed weight broadcast, sharding,
    # max-over-ranks timing, per-rank aggregation, the JSON line): every rank drives cuda:0 and the group is gloo.  The ranks
    # time-share the GPU, so the line says so and its numbers are not measurements.
    shared_gpu = os.environ.get("VVHIP_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    import torch.distributed as dist
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)   # torchrun, even with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner to STDOUT when its communicator comes up; stdout carries the one JSON line only, so the
        # banner is sent to stderr (fd level: the library writes past Python's sys.stdout)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if shared_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from vibevoice_amd import build as vbuild
    if rank == 0 and vbuild.stale():
        vbuild.build()
    if use_dist:
        dist.barrier()

    spec = dict(WORKLOADS[args.workload])
    for k, a in (("model", args.model), ("solver_steps", args.solver_steps), ("text_tokens", args.text_tokens),
                 ("voice_frames", args.voice_frames), ("speakers", args.speakers), ("kv_target", args.kv_start),
                 ("prefill_rows", args.prefill_rows)):
        if a is not None:
            spec[k] = a
    ctx = dict(rank=rank, world=world, device=device, use_dist=use_dist)
    if args.cpu_windows:
        print(json.dumps(cpu_baseline_windows(args, spec, device)), flush=True)
        return
    if "streaming" in spec["model"]:
        res = bench_streaming(args, spec, ctx)
    elif args.full_utterance:
        if world != 1:
            raise SystemExit("--full-utterance is a one-GPU measurement")
        res = bench_full_utterance(args, spec, ctx)
    else:
        res = bench_decode(args, spec, ctx, with_cpu=(world == 1 and not args.no_cpu_baseline and args.batch == 1), with_roofline=not args.no_roofline,
                           with_parity=(world == 1 and not args.no_cpu_baseline and not args.continuous))
        if rank == 0 and world == 1 and not args.skip_extra and args.workload == "north-star" and args.batch == 1 and not args.continuous:
            # the other single-GPU BASELINE configs, same run, same code
            extra = {}
            for name in ("1p5b", "streaming"):
                sp = dict(WORKLOADS[name])
                try:
                    a2 = parse_args([])
                    a2.xsplit, a2.no_graph, a2.cfg_scale = args.xsplit, args.no_graph, args.cfg_scale
                    if name == "streaming":
                        a2.steps = 60
                        r = bench_streaming(a2, sp, ctx)
                    else:
                        a2.steps, a2.warmup = 60, 10
                        a2.no_parity = args.no_parity or args.no_cpu_baseline
                        a2.no_parity_long = args.no_parity_long
                        r = bench_decode(a2, sp, ctx, with_cpu=False, with_roofline=not args.no_roofline, with_parity=True)
                    keep = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup")}
                    if r.get("parity_long"):
                        keep["parity_long"] = r["parity_long"]
                    if r.get("parity"):
                        keep["parity"] = r["parity"]
                        keep["gpu_eager_baseline"] = r.get("gpu_eager_baseline")
                    keep["workload"] = r["config"]["workload"]
                    if r.get("roofline"):
                        keep["roofline"] = {k: r["roofline"].get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_us", "launches_per_step",
                                                                            "bytes_per_launch", "whole_step_frac", "whole_step_achieved_frac", "traffic", "attention")}
                    for k in ("p50_first_audio_ms", "p90_first_audio_ms", "p50_first_chunk_on_device_ms", "prefill_plus_first_frame_s", "first_audio"):
                        if k in r.get("extra", {}):
                            keep[k] = r["extra"][k]
                    extra[sp["baseline_config"]] = keep
                except Exception as ex:          # an extra line must never take the main line down
                    extra[sp["baseline_config"]] = {"error": repr(ex)[:200]}
            # the small model with MORE THAN ONE decode chain on the GPU: the same 16-utterance queue through one engine context (8 slots,
            # continuous admission) and through two contexts sharing one weight copy (generate_interleaved: a host thread + stream each)
            try:
                lanes_res = {}
                for n_l in (1, 2):
                    a3 = parse_args([])
                    a3.xsplit, a3.no_graph, a3.cfg_scale = args.xsplit, args.no_graph, args.cfg_scale
                    a3.steps, a3.warmup, a3.batch, a3.continuous, a3.lanes = 100, 5, 8, 16, n_l
                    r3 = bench_decode(a3, dict(WORKLOADS["1p5b"]), ctx, with_cpu=False, with_roofline=False, with_parity=False)
                    lanes_res[f"{n_l}_context{'s' if n_l > 1 else ''}"] = {"audio_s_per_wall_s": r3["value"], "queue_wall_s": r3["extra"]["utterance_wall_s"],
                                                                            "capture_fallbacks": (r3["extra"]["continuous"] or {}).get("capture_fallbacks")}
                lanes_res["ratio"] = round(lanes_res["2_contexts"]["audio_s_per_wall_s"] / lanes_res["1_context"]["audio_s_per_wall_s"], 3)
                lanes_res["workload"] = ("BASELINE configs[1] shapes, 16 queued utterances of ~107 frames, 8 slots per engine context; whole queue incl. every prefill; "
                                         "2 contexts = vv_create_shared over ONE weight copy, one host thread + stream each")
                extra["configs[1] queue over engine contexts"] = lanes_res
            except Exception as ex:
                extra["configs[1] queue over engine contexts"] = {"error": repr(ex)[:200]}
            res["extra"]["configs"] = extra
        if not args.skip_extra and args.workload == "north-star" and args.batch == 1 and not args.continuous and not args.no_config3:
            # BASELINE configs[3]'s per-GPU unit, the configuration north_star's ">= 5x real-time" sentence names: 7B, 4 speakers, EIGHT utterances in
            # lock-step on each GPU, every one at a 32K context, N = 20 -- on EVERY rank (weak scaling: 8 utterances per GPU, the value is the
            # whole job's), with the batch kernels' own roofline (vv_gemv16p_kernel + the attention unit) and a batch-8 parity block on rank 0
            try:
                a4 = parse_args([])
                a4.xsplit, a4.no_graph, a4.cfg_scale, a4.gpus = args.xsplit, args.no_graph, args.cfg_scale, args.gpus
                a4.batch, a4.steps, a4.warmup = 8, 20, 5
                a4.no_eager_baseline, a4.no_parity_long, a4.no_cpu_baseline = True, True, True
                a4.no_parity = args.no_parity or args.no_cpu_baseline
                sp4 = dict(WORKLOADS["north-star"], speakers=4, text_tokens=10569, baseline_config="configs[3] per-GPU unit")
                r4 = bench_decode(a4, sp4, ctx, with_cpu=False, with_roofline=not args.no_roofline, with_parity=True)
                if rank == 0:
                    keep4 = {k: r4[k] for k in ("value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "scaling", "roofline", "parity")}
                    keep4["workload"] = r4["config"]["workload"]
                    keep4["utterances_per_gpu"] = 8
                    keep4["audio_s_per_wall_s_per_utterance"] = round(r4["value"] / (8 * world), 3)
                    keep4["per_rank_ms_per_step"] = r4["extra"]["per_rank_ms_per_step"]
                    keep4["prefill_phases"] = r4["extra"].get("prefill_phases")
                    res["extra"].setdefault("configs", {})["configs[3] per GPU"] = keep4
            except Exception as ex:          # an extra line must never take the main line down
                if rank == 0:
                    res["extra"].setdefault("configs", {})["configs[3] per GPU"] = {"error": repr(ex)[:300]}
    if rank == 0:
        if shared_gpu:
            res["harness"] = ("VVHIP_BENCH_SHARED_GPU=1: every rank time-shares cuda:0 over a gloo group -- a control-flow check of the "
                              "N > 1 path on a one-GPU box, NOT a measurement")
        print(json.dumps(res), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def bench_decode(args, spec, ctx, with_cpu, with_roofline, with_parity=False):
    """One multi-speaker model workload: prefill, W warm-up steps, K timed steps; returns the JSON-line dict (rank 0).
    with_cpu: the two oracle baselines (fp32 on the host cores, bf16 eager on this GPU); with_parity: the HIP engine of THIS run
    (full depth, xsplit / hipGraph as timed) is fed the prompt, forced schedule and noise of those oracle runs, teacher-forced
    per step, and the line carries the worst per-step differences (oracle/parity.py; SURVEY 8d's tolerances).  with_parity
    without with_cpu runs the bf16 eager leg only (no CPU minute)."""
    import torch.distributed as dist
    from vibevoice_amd import parallel, synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.engine import Engine, map_param_name
    from vibevoice_amd.modeling import BenchHooks, VibeVoiceForConditionalGenerationInference, engine_config_from_reference
    rank, world, device, use_dist = ctx["rank"], ctx["world"], ctx["device"], ctx["use_dist"]
    model_key = spec["model"]
    cfg = CONFIGS[model_key]
    ckpt = find_checkpoint(model_key)
    if ckpt:                                     # real weights ($VIBEVOICE_MODEL_DIR): the checkpoint's own config decides the shapes
        with open(os.path.join(ckpt, "config.json")) as f:
            cfg = json.load(f)
    K, W, NS = args.steps, max(1, args.warmup), spec["solver_steps"]
    B = max(1, min(8, args.batch))
    n_utt = max(B, args.continuous) if args.continuous else B
    inputs = synthetic.synthetic_inputs(cfg, n_speakers=spec["speakers"], text_tokens=spec["text_tokens"],
                                        voice_frames=spec["voice_frames"], seed=100 + rank, batch=n_utt)
    L0 = inputs["input_ids"].shape[1]
    total_steps = W + K + 2
    d = cfg["decoder_config"]
    model_ctx = d.get("max_position_embeddings", 32768)
    kv_target = int(spec.get("kv_target") or 0)
    if kv_target:
        kv_target = max(L0, min(kv_target, model_ctx - total_steps - 8))       # the timed window must fit the model's context
    max_ctx = args.max_ctx or ((max(L0, kv_target) + total_steps + 256 + 127) // 128 * 128)
    ecfg = engine_config_from_reference(cfg, n_slots=B, max_ctx=max_ctx, xsplit=args.xsplit, use_graph=not args.no_graph,
                                        enc_frames=args.enc_frames, max_rows=max(2 * B, spec["prefill_rows"]))
    t_load0 = time.time()
    eng = Engine(ecfg, device)
    exp = eng.expected_weights()
    with_parity = with_parity and rank == 0 and not args.no_parity
    keep_cpu = rank == 0 and (with_cpu or with_parity)
    cpu_sd = {}
    # rank 0 draws the weights; ONE packed-blob broadcast over RCCL/xGMI at start-up, no collective later
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    make = lambda k, shape: synthetic.random_tensor(k, shape, gen, device, torch.bfloat16)
    bc = {}
    scaling, sbias = 0.2, -0.05
    if ckpt:                                     # every rank reads the shards itself (page cache shared on one node): no collective
        source = ((k, t) for k, t in checkpoint_tensors(ckpt))
    else:
        source = parallel.broadcast_packed(synthetic.param_shapes(cfg).items(), make, device, torch.bfloat16, stats=bc)
    for k, t in source:
        if k == "model.speech_scaling_factor":
            scaling = float(t)
            continue
        if k == "model.speech_bias_factor":
            sbias = float(t)
            continue
        name = map_param_name(k)
        if name in exp:
            eng.upload(name, t)
        if keep_cpu:
            cpu_sd[k] = t.to("cpu", torch.bfloat16)
        del t
    miss = eng.missing_weights()
    if miss:
        raise SystemExit(f"engine parameters not provided: {miss[:5]}")
    model = VibeVoiceForConditionalGenerationInference(cfg, eng, model_dtype=torch.bfloat16)
    model.set_speech_factors(scaling, sbias)
    model.set_ddpm_inference_steps(NS)
    load_s = time.time() - t_load0

    forced = [synthetic.forced_schedule(total_steps, turn=150) for _ in range(n_utt)]
    g = torch.Generator(device=device)
    g.manual_seed(1234 + rank)
    noise_bank = torch.randn(total_steps + 1, 2 * B, cfg["acoustic_vae_dim"], generator=g, device=device)
    kvh, hd, n_layers = d["num_key_value_heads"], d["hidden_size"] // d["num_attention_heads"], d["num_hidden_layers"]

    def kv_fill(e, cache, p0, p1):
        """positions [p0, p1) of every layer of `cache`: random bf16 K/V at the scale of real keys/values (not zeros: the
        softmax over 32K positions and the P.V accumulation then run on ordinary data)"""
        gk = torch.Generator(device=device)
        gk.manual_seed(4321 + cache)
        for layer in range(n_layers):
            k = (torch.randn(kvh, p1 - p0, hd, generator=gk, device=device) * 0.5).to(torch.bfloat16)
            v = (torch.randn(kvh, p1 - p0, hd, generator=gk, device=device) * 0.5).to(torch.bfloat16)
            e.kv_import_at(cache, layer, p0, k, v)
    torch.cuda.synchronize()

    marks = {}

    def step_cb(step):
        if step == W or step == W + K:
            eng.sync()
            torch.cuda.synchronize()
            # ranks meet at the edges of the timed region -- only where every rank is certain to get there: a continuous queue
            # may drain on one rank before step W + K (its region then ends with its generate()), and a barrier the others
            # still wait at would hang the job; the collectives AFTER generate() (same count on every rank) do the aggregation
            if use_dist and not args.continuous:
                dist.barrier()
                torch.cuda.synchronize()
            marks[step] = time.perf_counter()
        if step == 1:
            eng.sync()
            marks["prefill_done"] = time.perf_counter()
        if args.host_delay_us > 0 and step > W:
            t_end = time.perf_counter() + args.host_delay_us * 1e-6
            while time.perf_counter() < t_end:
                pass

    os.environ.setdefault("VVHIP_TIME_PREFILL", "1")        # sync + time the two prefill phases (outside the timed region)
    if not os.environ.get("VVHIP_COLD_PREFILL"):
        # the reported times are a warm serving process's: model.warmup() (what from_pretrained runs) touches every kernel the
        # request needs -- voice-prompt encoder, a prompt pass of the same row count, decode frames; the real prefill overwrites
        # the cache positions it touched
        n_prompt = int(inputs["attention_mask"][0].sum())
        t_w0 = time.perf_counter()
        model.warmup(prompt_rows=[min(n_prompt, eng.cfg.max_rows)], voice_frames=spec["voice_frames"])
        warm_s = time.perf_counter() - t_w0
    else:
        warm_s = 0.0
    t_gen0 = time.perf_counter()
    if args.continuous:
        reqs = []
        for i in range(n_utt):
            r = {k: (v[i:i + 1] if k in ("input_ids", "attention_mask", "speech_input_mask") else v) for k, v in inputs.items()}
            ns = spec["speakers"]
            r["speech_tensors"] = inputs["speech_tensors"][i * ns:(i + 1) * ns]
            r["speech_masks"] = inputs["speech_masks"][i * ns:(i + 1) * ns]
            r["_forced_tokens"] = forced[i][:W + K + 1 + 7 * (i % 3)] + [synthetic.TOKENS.eos_token_id]     # staggered ends
            reqs.append(r)
        if args.lanes > 1:
            # the queue over `lanes` engine contexts sharing this model's weights (vv_create_shared), one host thread + stream per lane:
            # the lanes exist before the clock starts (a serving process creates them once), one untimed queue warms their graphs
            model.generate_interleaved(reqs[:args.lanes * 2], lanes=args.lanes, tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale,
                                       generation_config={"do_sample": False}, max_concurrent=B)
            eng.sync(); torch.cuda.synchronize()
            t_gen0 = time.perf_counter()
            outs = model.generate_interleaved(reqs, lanes=args.lanes, tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale,
                                              generation_config={"do_sample": False}, max_concurrent=B)
            torch.cuda.synchronize()
            marks[W] = t_gen0
        else:
            outs = model.generate_continuous(reqs, tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale,
                                             generation_config={"do_sample": False}, max_concurrent=B, _bench_hooks=BenchHooks(step_callback=step_cb))
        out = outs[0]
    else:
        out = model.generate(tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale, generation_config={"do_sample": False},
                             max_new_tokens=total_steps, show_progress_bar=False, _forced_tokens=forced,
                             _noise_fn=lambda step, n2: noise_bank[step],
                             _bench_hooks=BenchHooks(step_callback=step_cb, kv_start=kv_target, kv_fill_fn=kv_fill if kv_target else None), **inputs)
    eng.sync()
    t_gen1 = time.perf_counter()
    if W + K not in marks:                         # continuous mode may finish early
        marks[W + K] = t_gen1
    wall = marks[W + K] - marks[W]
    # steps W..W+K-1: count the <speech_diffusion> frames among them (the schedule inserts 2 control tokens per 150)
    frames = B * sum(1 for t in forced[0][W:W + K] if t == synthetic.TOKENS.speech_diffusion_id)
    frames_all, wall_max = parallel.aggregate_throughput(frames, wall, device)   # sum over ranks / max over ranks
    value = frames_all * FRAME_SEC / wall_max
    if args.continuous:            # whole queue: prefill of every admitted utterance + all decode iterations
        frames_all, wall_max = parallel.aggregate_throughput(model.last_stats["frames"], t_gen1 - t_gen0, device)
        value = frames_all * FRAME_SEC / wall_max
    audio_total = out.speech_outputs[0].shape[-1] / 24000.0
    prefill_phases = getattr(model, "last_prefill", None)
    cont_stats = dict(model.last_stats) if args.continuous else None
    # every rank's own step time next to the max: an imbalance (a slow GPU, a straggling host loop) shows in the first SCALE run
    per_rank_ms = [round(v, 4) for v in parallel.per_rank_values(wall / K * 1e3, device)]
    # the result phase of an N-GPU job, outside the timed region: every rank's finished utterances -- token sequences and
    # waveforms of the generate() above -- travel to rank 0 as device tensors through parallel.gather_outputs (RCCL gather; with one
    # rank under torchrun it is the same collective on a one-rank communicator).  Rank 0 checks what arrived against its own rows.
    rccl = None
    if use_dist:
        from vibevoice_amd.modeling import VibeVoiceGenerationOutput
        per = list(outs) if args.continuous else [
            VibeVoiceGenerationOutput(sequences=out.sequences[b:b + 1], speech_outputs=[out.speech_outputs[b]],
                                      reach_max_step_sample=out.reach_max_step_sample[b:b + 1]) for b in range(out.sequences.shape[0])]
        n_per = len(per)
        gst = {}
        got = parallel.gather_outputs(per, [list(range(r * n_per, (r + 1) * n_per)) for r in range(world)], world * n_per, device,
                                      torch.bfloat16, gather_to=0, stats=gst)
        if rank == 0:
            ok = len(got) == world * n_per and all(g is not None and g.speech_outputs[0] is not None for g in got)
            ok = ok and all(torch.equal(got[j].sequences, per[j].sequences.cpu()) and
                            torch.equal(got[j].speech_outputs[0], per[j].speech_outputs[0].reshape(1, -1).float().cpu()) for j in range(n_per))
            rccl = {"backend": dist.get_backend(), "ranks": world,
                    "weights_broadcast": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bc.items()},
                    "result_gather": gst.get("gather"), "utterances_received_on_rank0": len(got),
                    "audio_seconds_received": round(sum(g.speech_outputs[0].shape[-1] for g in got if g is not None) / 24000.0, 2),
                    "rank0_rows_identical_after_the_round_trip": bool(ok),
                    "collectives_inside_the_step_loop": 0}
    # first-audio latency as SURVEY 8d defines it: generate() entry (voice-prompt encode + prompt prefill + first frame) -> the
    # first chunk an AudioStreamer consumer receives on the host; 5 trials of the same request
    first_audio = None
    if rank == 0 and world == 1 and B == 1 and not args.continuous and not os.environ.get("VVHIP_NO_TTFA"):
        try:
            one_in = {k: v for k, v in inputs.items()}
            os.environ.pop("VVHIP_TIME_PREFILL", None)          # the phase timer's syncs do not belong in a latency measurement
            lat = first_audio_trials(lambda st: model.generate(
                tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale, generation_config={"do_sample": False}, max_new_tokens=3,
                show_progress_bar=False, _forced_tokens=forced, _noise_fn=lambda step, n2: noise_bank[step], audio_streamer=st, **one_in), 5)
            if lat:
                first_audio = {"p50_ms": round(sorted(lat)[len(lat) // 2], 2), "trials_ms": [round(x, 2) for x in lat],
                               "definition": "generate() entry -> first chunk delivered to an AudioStreamer consumer thread on the host "
                                             "(voice-prompt encode + prompt prefill + first frame + D2H + queue hand-off), warm process"}
        except Exception as ex:
            first_audio = {"error": repr(ex)[:200]}

    if os.environ.get("VVHIP_TIMELINE") and rank == 0:      # timing builds only: tools/step_timeline.py
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import step_timeline
        step_timeline.dump(eng, os.environ["VVHIP_TIMELINE"])

    kv_mid = max(L0, kv_target) + W + K // 2
    # algorithmic bytes of one step: every weight byte once (the B utterances in flight share each weight pass) + each
    # utterance's own KV traffic
    one = algorithmic_bytes_per_frame(cfg, NS, kv_mid, 1 + min(150, K) // 2)
    kv_only = one - algorithmic_bytes_per_frame(cfg, NS, 0, 0)
    formula = one + (B - 1) * kv_only

    # ---- roofline of the dominant kernel (vv_gemv_kernel) ----
    roof = None
    if rank == 0 and with_roofline and not args.continuous:
        kprof = 6
        forced_p = [synthetic.forced_schedule(kprof + 3, turn=150) for _ in range(B)]
        prof = {}

        def prof_cb(step):
            if step == 2:
                eng.sync()
                eng.profile_begin()
            if step == 2 + kprof:
                prof["res"] = eng.profile_end()
        inp2 = synthetic.synthetic_inputs(cfg, n_speakers=spec["speakers"], text_tokens=min(spec["text_tokens"], 220),
                                          voice_frames=spec["voice_frames"], seed=100, batch=B)
        # the window is recorded at the timed KV length (the attention launches' bytes depend on it; the GEMV launches do not)
        model.generate(tokenizer=synthetic.TOKENS, cfg_scale=args.cfg_scale, generation_config={"do_sample": False},
                       max_new_tokens=kprof + 3, show_progress_bar=False, _forced_tokens=forced_p,
                       _noise_fn=lambda step, n2: noise_bank[step],
                       _bench_hooks=BenchHooks(step_callback=prof_cb, kv_start=kv_target, kv_fill_fn=kv_fill if kv_target else None), **inp2)
        (n_l, ms_cal, by), (n_o, ms_o, by_o) = prof["res"]       # [decode GEMV kernel], [other GEMM kernels]
        # (1) launch duration in the execution mode of the timed region: the recorded GEMV launches replayed as ONE dependent
        # hipGraph chain between two events (vv_profile_replay) = start-to-start period of a launch = what rocprofv3
        # --kernel-trace reports per kernel under graph replay (profiles/r02_*_kernel_stats.csv)
        n_rep, ms_rep, by_rep = eng.profile_replay(reps=3)
        us_graph = ms_rep * 1e3 / max(1, n_rep)
        ach = by_rep / 1e9 / (ms_rep / 1e3) if ms_rep > 0 else 0.0                      # GB/s
        # (2) secondary: hipEvent pair around each EAGER launch minus an in-stream empty pair (a lower bound on the duration)
        ach_pair = by / 1e9 / (ms_cal / 1e3) if ms_cal > 0 else 0.0
        pmc_key = model_key if B == 1 else f"{model_key}-batch{B}"       # the batch leg has its own FETCH_SIZE pass (other launch geometry)
        traffic, traffic_src = pmc_traffic(pmc_key, "vv_gemv_kernel", by_rep / max(1, n_rep))
        roof = {"bound": "hbm", "kernel": "vv_gemv_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "method": "recorded vv_gemv_kernel launches of 6 live steps replayed as one dependent hipGraph chain, hipEvents around 3 replays "
                          "(vv_profile_replay): launch period = kernel + boundary, the quantity rocprofv3 --kernel-trace reports",
                "launches_per_step": round(n_l / kprof, 1), "avg_launch_us": round(us_graph, 3),
                "bytes_per_launch": round(by_rep / max(1, n_rep), 1),
                "gemv_bytes_per_step": round(by / kprof, 1),
                "event_pair": {"avg_launch_us": round(ms_cal * 1e3 / max(1, n_l), 3), "achieved": round(ach_pair, 1),
                               "frac": round(ach_pair / HBM_PEAK_GBS, 4), "raw_pair_us": round(eng.stat(2) / 1e3 / max(1, n_l), 3),
                               "empty_pair_us": round(eng.stat(3) / 1e3, 3), "note": "eager launches, per-launch event pair minus an empty pair"},
                "other_gemm": {"launches_per_step": round(n_o / kprof, 1), "bytes_per_step": round(by_o / kprof, 1),
                               "GBps": round(by_o / 1e9 / (ms_o / 1e3), 1) if ms_o > 0 else None},
                "formula_bytes_per_step": round(formula, 1), "formula_kv_bytes_per_utterance": round(kv_only, 1),
                "whole_step_GBps": round(formula / 1e9 / (wall_max / K), 1),
                "whole_step_frac": round(formula / 1e9 / (wall_max / K) / HBM_PEAK_GBS, 4),
                "whole_step_frac_basis": "SURVEY 8(d) formula bytes per step (every head weight incl. the adaLN matrices once per solver step) / step time / 8 TB/s"}
        # the other timed kernel families of the same window, each replayed as its own dependent chain
        def family(fid, name):
            n_f, ms_f, by_f = eng.profile_replay(reps=3, family=fid)
            if not n_f or ms_f <= 0:
                return None
            a_f = by_f / 1e9 / (ms_f / 1e3)
            tr, tsrc = pmc_traffic(pmc_key, name.split(" ")[0], by_f / n_f)
            return {"kernel": name, "achieved": round(a_f, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a_f / HBM_PEAK_GBS, 4),
                    "launches_per_step": round(n_f / 3 / kprof, 1), "avg_launch_us": round(ms_f * 1e3 / n_f, 3),
                    "bytes_per_launch": round(by_f / n_f, 1), "traffic": tr, "traffic_source": tsrc}
        roof["attention"] = family(2, "vv_attn_fused_kernel (+ vv_attn_merge2_kernel): one unit per layer, KV bytes of every row")
        p16 = family(1, "vv_gemv16p_kernel")
        # bytes the implementation actually moves per step (the adaLN matrices are hoisted out of the solver loop, so this is
        # below the formula): every recorded weight-streaming launch's own byte count + the attention units' KV bytes
        moved = (by + by_o) / kprof + sum((f["bytes_per_launch"] * f["launches_per_step"]) for f in (roof["attention"], p16) if f)
        roof["whole_step_moved_bytes"] = round(moved, 1)
        roof["whole_step_achieved_GBps"] = round(moved / 1e9 / (wall_max / K), 1)
        roof["whole_step_achieved_frac"] = round(moved / 1e9 / (wall_max / K) / HBM_PEAK_GBS, 4)
        if p16 is not None:
            # batch decode (5..16 rows): the LM / head projections run in vv_gemv16p_kernel (pre-packed activations) -- THAT is the
            # dominant kernel of the step; the vv_gemv_kernel figures (the remaining 16-row tokenizer / sampler launches) move aside
            roof["gemv_other"] = {k: roof[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "avg_launch_us", "bytes_per_launch", "traffic")}
            for k in ("kernel", "achieved", "frac", "launches_per_step", "avg_launch_us", "bytes_per_launch", "traffic", "traffic_source"):
                roof[k] = p16[k]
            roof["method"] = roof["method"].replace("vv_gemv_kernel", "vv_gemv16p_kernel")

    # ---- CPU baseline: the oracle loop on the host cores, bounded sample ----
    cpu = None
    eager = None
    parity = None
    legs = {}
    if keep_cpu and with_cpu:
        try:
            # on the configuration the GPU number is quoted on: the window runs at the timed KV length (kv_target > 0); its leg is then
            # not a parity leg (the parity block below runs its own fp32 oracle on the GPU)
            leg_cpu, cpu = cpu_baseline(cfg, cpu_sd, NS, args.cfg_scale, args.cpu_frames, model_key, kv_len=(max(L0, kv_target) + W if kv_target else 0))
            if leg_cpu is not None:
                legs["vs_fp32"] = leg_cpu
        except Exception as ex:   # the baseline is a reported number, never the product path
            cpu = {"value": None, "error": repr(ex)[:200]}
    if keep_cpu and not args.no_eager_baseline:
        try:
            _free_run_leg, eager = gpu_eager_baseline(cfg, cpu_sd, NS, args.cfg_scale, 8, model_key, device,
                                                      kv_len=(max(L0, kv_target) + W if kv_target else 0))     # timing only
            del _free_run_leg
            eager["speedup_of_this_path"] = round(value / world / eager["value"], 2) if eager["value"] else None
        except Exception as ex:   # a reported number, never the product path
            eager = {"value": None, "error": repr(ex)[:200]}
        torch.cuda.empty_cache()
    if with_parity and B > 1:
        # ---- the batch that was just timed, full depth, against the oracle: B rows in lock-step (16-row LM / head projections over
        # pre-packed activations, batch attention, slot-batched tokenizer chains), two DIFFERENT requests alternating over the rows,
        # every row teacher-forced per step by its own fp32 oracle leg (oracle/parity.py::compare_engine_batch) ----
        from oracle import parity as oparity
        parity = {"model": f"VibeVoice-{model_key}", "lm_layers": d["num_hidden_layers"], "head_layers": cfg["diffusion_head_config"].get("head_layers", 4),
                  "solver_steps": NS, "rows": B, "engine_mode": {"xsplit": args.xsplit, "hipgraph": not args.no_graph, "dtype": "bf16"}, "prompt_tokens": 48,
                  "definition": "the timed engine (same weights, batch kernels) decoding B rows in lock-step: rows alternate over two requests (different "
                                "prompt ids and noise), each row teacher-forced per step by the fp32 oracle run of ITS request (oracle/generate.py as fp32 "
                                "eager ops on this GPU); worst row and step; rel-L2 unless marked dB; bounds = the stated tolerance vs fp32 (SURVEY 8d)"}
        try:
            t_pb = time.perf_counter()
            blegs = [oparity.oracle_leg(cfg, cpu_sd, synthetic.TOKENS, NS, args.cfg_scale, 3, device, torch.float32, 20.0, seed=sd_) for sd_ in (7, 11)]
            rb = oparity.compare_engine_batch(model, blegs, synthetic.TOKENS, B)
            vb = oparity.verdict("vs_fp32", rb)
            vb["oracle"] = "float32 eager ops on this GPU, one run per distinct request"
            parity["vs_fp32"] = vb
            parity["within_bounds"] = bool(vb["within_bounds"] and vb["greedy_pick_equal"] and not vb["nonfinite_steps"])
            parity["seconds"] = round(time.perf_counter() - t_pb, 1)
            del blegs
        except Exception as ex:
            parity["error"] = repr(ex)[:300]
            parity["within_bounds"] = False
        model.set_ddpm_inference_steps(NS)
        torch.cuda.empty_cache()
    elif with_parity:
        # ---- full-depth parity of the engine that was just timed (same weights, same execution mode) against the oracle legs ----
        from oracle import parity as oparity
        parity = {"model": f"VibeVoice-{model_key}", "lm_layers": d["num_hidden_layers"], "head_layers": cfg["diffusion_head_config"].get("head_layers", 4),
                  "solver_steps": NS, "engine_mode": {"xsplit": args.xsplit, "hipgraph": not args.no_graph, "dtype": "bf16"},
                  "prompt_tokens": 48, "weights": "the timed run's (synthetic, seeded)" if not ckpt else "checkpoint",
                  "definition": "HIP engine vs the oracle loop (oracle/generate.py, the restatement of the reference's generate()) on the same prompt, "
                                "forced <speech_diffusion> schedule and noise, teacher-forced per step; worst step; rel-L2 unless marked dB; "
                                "bounds = SURVEY 8(d).  Three runs on IDENTICAL inputs: the fp32 oracle (free-running, defines the trajectory), the oracle as "
                                "bf16 eager ops on this GPU (the reference's GPU dtype) and the engine, both teacher-forced by the fp32 run.  "
                                "reference_bf16_vs_fp32 = the reference path's own rounding noise at this depth"}
        try:
            if "vs_fp32" not in legs:      # no CPU leg in this run (the extra configs): the fp32 oracle as eager ops on this GPU
                legs["vs_fp32"] = oparity.oracle_leg(cfg, cpu_sd, synthetic.TOKENS, NS, args.cfg_scale, 4, device, torch.float32, 20.0)
            fp32_leg = legs["vs_fp32"]
            # the reference's GPU dtype on the SAME inputs: the oracle as bf16 eager ops, teacher-forced by the fp32 run
            bf16_leg = oparity.oracle_leg(cfg, cpu_sd, synthetic.TOKENS, NS, args.cfg_scale, 8, device, torch.bfloat16, 20.0, teacher=fp32_leg)
            floor = oparity.compare_legs(bf16_leg, fp32_leg)
            parity["reference_bf16_vs_fp32"] = floor
            both = oparity.compare_engine(model, fp32_leg, synthetic.TOKENS, also={"bf16": bf16_leg})
            r32 = oparity.verdict("vs_fp32", {k: v for k, v in both.items() if k != "also"})
            r32["oracle"] = f"{fp32_leg.dtype} on {fp32_leg.device}".replace("torch.", "")
            r16 = oparity.verdict("vs_bf16_eager", both["also"]["bf16"], floor=floor, vs_fp32=r32)
            r16["oracle"] = f"{bf16_leg.dtype} on {bf16_leg.device}, teacher-forced by the fp32 run (identical inputs)".replace("torch.", "")
            parity["vs_fp32"], parity["vs_bf16_eager"] = r32, r16
            # the engine's bf16 mode (fp32 residual stream, bf16 only at the matrix-unit inputs) against the reference's bf16 path
            parity["engine_at_least_as_close_to_fp32_as_reference_bf16"] = {k: bool(r32[k] <= 1.1 * floor[k] + 1e-3) for k in ("latent", "pos_hidden", "neg_hidden")}
            parity["within_bounds"] = bool(r32["within_bounds"] and r16["within_bounds"]
                                           and all(parity["engine_at_least_as_close_to_fp32_as_reference_bf16"].values()))
            del bf16_leg
        except Exception as ex:
            parity["error"] = repr(ex)[:300]
            parity["within_bounds"] = False
        model.set_ddpm_inference_steps(NS)
    parity_long = None
    if with_parity and B == 1 and not args.no_parity_long:
        # ---- the same comparison THROUGH THE PROMPT PASS, at the timed length: the run's own request (voice prompts + the whole L0-token
        # prompt) through the engine's prefill chain and through the oracle as fp32 eager ops on this GPU with the same weights; step 0 =
        # the prompt's last position; then teacher-forced decode frames on the KV cache the prefill kernels wrote ----
        from oracle import parity as oparity
        try:
            one = {k: (v[:1] if k in ("input_ids", "attention_mask", "speech_input_mask") else v[:spec["speakers"]]) for k, v in inputs.items()}
            t_pl = time.perf_counter()
            lf32 = oparity.oracle_leg(cfg, cpu_sd, synthetic.TOKENS, NS, args.cfg_scale, 4, device, torch.float32, 60.0, inputs=one, attn_rows=1024)
            lbf16 = oparity.oracle_leg(cfg, cpu_sd, synthetic.TOKENS, NS, args.cfg_scale, 4, device, torch.bfloat16, 60.0, teacher=lf32, attn_rows=1024)
            lfloor = oparity.compare_legs(lbf16, lf32)
            lboth = oparity.compare_engine(model, lf32, synthetic.TOKENS, also={"bf16": lbf16})
            l32 = oparity.verdict("vs_fp32", {k: v for k, v in lboth.items() if k != "also"})
            l16 = oparity.verdict("vs_bf16_eager", lboth["also"]["bf16"], floor=lfloor, vs_fp32=l32)
            closer = {k: bool(l32[k] <= 1.1 * lfloor[k] + 1e-3) for k in ("latent", "pos_hidden", "neg_hidden")}
            parity_long = {"model": f"VibeVoice-{model_key}", "lm_layers": d["num_hidden_layers"], "prompt_tokens": L0, "speakers": spec["speakers"],
                           "voice_frames_per_speaker": spec["voice_frames"], "solver_steps": NS, "frames": l32["frames"],
                           "engine_mode": {"xsplit": args.xsplit, "hipgraph": not args.no_graph, "dtype": "bf16"},
                           "definition": "the timed run's own request: voice prompts through the acoustic encoder + connector, the whole prompt through "
                                         "vv_pack_rows -> vv_gemm4 (QKV + bias + RoPE + KV append) -> vv_attn_prefill4 -> vv_gemm4 at full depth, vs the oracle as "
                                         "fp32 eager ops on this GPU (same weights, same sampling draws); step0 = the hidden state at the prompt's last "
                                         "position / negative condition / first latent; then teacher-forced decode frames on the KV the prefill kernels wrote; "
                                         "worst step; bounds = SURVEY 8(d)",
                           "step0": oparity.compare_engine_first_step(lboth), "vs_fp32": l32, "vs_bf16_eager": l16,
                           "reference_bf16_vs_fp32": lfloor, "engine_at_least_as_close_to_fp32_as_reference_bf16": closer,
                           "oracle_prompt_phase_s": {"fp32_eager": round(lf32.prompt_s, 3), "bf16_eager": round(lbf16.prompt_s, 3)},
                           "within_bounds": bool(l32["within_bounds"] and l16["within_bounds"] and all(closer.values())),
                           "seconds": round(time.perf_counter() - t_pl, 1)}
            del lf32, lbf16
        except Exception as ex:
            parity_long = {"error": repr(ex)[:300], "within_bounds": False}
        model.set_ddpm_inference_steps(NS)
        torch.cuda.empty_cache()
    legs.clear()
    cpu_sd.clear()
    res = {
        "metric": "audio-sec/wall-sec", "value": round(value, 3), "unit": "audio-s/wall-s", "n_gpus": world,
        "steps": K, "warmup": W, "ms_per_step": round((marks[W + K] - marks[W]) / K * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic" if not ckpt else "synthetic inputs, checkpoint weights",
        "config": {"workload": f"BASELINE {spec.get('baseline_config', '?')}: VibeVoice-{model_key.upper()} shapes, {spec['speakers']} speaker(s), "
                               f"{L0}-token prompt ({spec['text_tokens']} text + {spec['speakers']}x{spec['voice_frames']}-frame voice) prefilled through the engine, "
                               f"{NS} solver steps, cfg {args.cfg_scale}, decode timed at KV length {max(L0, kv_target) + W}..{max(L0, kv_target) + W + K}"
                               + (" (positions past the prompt: random bf16 K/V)" if kv_target > L0 else "")
                               + f", {B} utterance{'s' if B > 1 else ''} per GPU"
                               + (f" ({n_utt} queued, continuous admission)" if args.continuous else "")
                               + (f" over {args.lanes} engine contexts sharing one weight copy" if args.continuous and args.lanes > 1 else "") + ", forced token schedule",
                   "model": f"VibeVoice-{model_key}", "solver_steps": NS, "prompt_tokens": L0, "speakers": spec["speakers"],
                   "xsplit": args.xsplit, "hipgraph": not args.no_graph, "kv_len_timed": max(L0, kv_target) + W,
                   **({"host_delay_us": args.host_delay_us} if args.host_delay_us > 0 else {}),
                   "parallelism": f"utterance-dp{world}"},
        "roofline": roof, "cpu_baseline": cpu, "gpu_eager_baseline": eager, "parity": parity, "parity_long": parity_long,
        "extra": {"frames_timed": frames, "weights_load_s": round(load_s, 2), "libvvhip_build_id": _build_id(),
                  # hipGraph executables this engine holds, captures that fell back to eager runs, nodes of the captured graphs that are not
                  # kernel launches (0 by construction: DESIGN.md section 8, the memset node of a replayed graph)
                  "captured_graphs": {"executables": eng.stat(1), "capture_fallbacks": eng.stat(4), "non_kernel_nodes": eng.stat(5)},
                  "weights_broadcast": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in bc.items()},
                  "weights_source": (f"checkpoint {ckpt}" if ckpt else "synthetic (seeded N(0, 0.02^2) at the config's shapes)"),
                  "per_rank_ms_per_step": per_rank_ms, "rccl": rccl, "warmup_s": round(warm_s, 3), "first_audio": first_audio,
                  "sharding": parallel.shard_report([L0] * (world * n_utt), [list(range(r * n_utt, (r + 1) * n_utt)) for r in range(world)]),
                  "prefill_plus_first_frame_s": round(marks.get("prefill_done", t_gen0) - t_gen0, 4),
                  "prefill_phases": prefill_phases,
                  "utterance_audio_s": round(audio_total, 2), "utterance_wall_s": round(t_gen1 - t_gen0, 3),
                  "continuous": cont_stats},
    }
    if prefill_phases and prefill_phases.get("lm_prefill_s"):
        import math
        sh = synthetic.param_shapes(cfg)
        n_lm = sum(math.prod(s) for k, s in sh.items() if k.startswith("model.language_model.layers."))
        hq = d["num_attention_heads"]
        flops = 2.0 * n_lm * L0 * B + 4.0 * hq * hd * (L0 * L0 / 2.0) * n_layers * B
        res["extra"]["prefill_tflops"] = round(flops / prefill_phases["lm_prefill_s"] / 1e12, 1)
        res["extra"]["prefill_frac_of_2p5PF"] = round(flops / prefill_phases["lm_prefill_s"] / 2.5e15, 4)
    eng.close()
    del model, eng
    torch.cuda.empty_cache()
    return res


def bench_full_utterance(args, spec, ctx):
    """SURVEY 8(d)'s metric on ONE WHOLE UTTERANCE of the workload (default: BASELINE configs[2]): generate() from the prompt to the
    reference's own length cap (max_length_times = 2 -> 2 * L0 generated tokens, modeling_vibevoice_inference.py:421; --utterance-frames
    caps it lower), the KV cache REALLY growing step by step (no imported K/V), the forced schedule's speaker turns included.
    value = audio seconds / wall(generate()) with the wall clock around the whole call: voice-prompt encode + prompt prefill + every
    decode step + output assembly -- the reference demo's own figure (demo/inference_from_file.py:388-410, inverted).  Also reports
    ms/step in windows at several context lengths against the 8(d) bytes of that length, the device memory in use, and the size
    of the hipGraph cache (the attention geometry, hence the step graph, changes every 1024 positions)."""
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.engine import Engine, map_param_name
    from vibevoice_amd.modeling import BenchHooks, VibeVoiceForConditionalGenerationInference, engine_config_from_reference
    device = ctx["device"]
    model_key = spec["model"]
    cfg = CONFIGS[model_key]
    ckpt = find_checkpoint(model_key)
    if ckpt:
        with open(os.path.join(ckpt, "config.json")) as f:
            cfg = json.load(f)
    NS = spec["solver_steps"]
    inputs = synthetic.synthetic_inputs(cfg, n_speakers=spec["speakers"], text_tokens=spec["text_tokens"],
                                        voice_frames=spec["voice_frames"], seed=100, batch=1)
    L0 = inputs["input_ids"].shape[1]
    d = cfg["decoder_config"]
    model_ctx = d.get("max_position_embeddings", 32768)
    cap = min(model_ctx - L0, 2 * L0)                              # :421 -- the loop's own length
    n_steps = min(cap, args.utterance_frames) if args.utterance_frames else cap
    max_ctx = (L0 + n_steps + 256 + 127) // 128 * 128
    free0, total_mem = torch.cuda.mem_get_info(device)
    ecfg = engine_config_from_reference(cfg, n_slots=1, max_ctx=max_ctx, xsplit=args.xsplit, use_graph=not args.no_graph,
                                        enc_frames=args.enc_frames, max_rows=max(2, spec["prefill_rows"]))
    eng = Engine(ecfg, device)
    exp = eng.expected_weights()
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    scaling, sbias = 0.2, -0.05
    source = checkpoint_tensors(ckpt) if ckpt else ((k, synthetic.random_tensor(k, shp, gen, device, torch.bfloat16))
                                                     for k, shp in synthetic.param_shapes(cfg).items())
    for k, t in source:
        if k == "model.speech_scaling_factor":
            scaling = float(t)
        elif k == "model.speech_bias_factor":
            sbias = float(t)
        else:
            name = map_param_name(k)
            if name in exp:
                eng.upload(name, t)
        del t
    model = VibeVoiceForConditionalGenerationInference(cfg, eng, model_dtype=torch.bfloat16)
    model.set_speech_factors(scaling, sbias)
    model.set_ddpm_inference_steps(NS)
    T = synthetic.TOKENS
    forced = [synthetic.forced_schedule(n_steps - 1, turn=150) + [T.eos_token_id]]
    n_frames_plan = sum(1 for t in forced[0] if t == T.speech_diffusion_id)
    g = torch.Generator(device=device)
    g.manual_seed(1234)
    noise_bank = torch.randn(n_steps + 1, 2, cfg["acoustic_vae_dim"], generator=g, device=device)
    n_prompt = int(inputs["attention_mask"][0].sum())
    model.warmup(prompt_rows=[min(n_prompt, eng.cfg.max_rows)], voice_frames=spec["voice_frames"])
    os.environ.pop("VVHIP_TIME_PREFILL", None)                     # no extra syncs inside the measured call
    # ms/step in windows of WIN steps starting at these KV lengths (those the utterance reaches)
    WIN = 200
    targets = [L for L in (L0 + 64, 16384, 22000, 28000, L0 + n_steps - WIN - 8) if L0 + 8 <= L <= L0 + n_steps - WIN - 4]
    edges = {}
    for L in targets:
        edges[L - L0] = ("a", L)
        edges[L - L0 + WIN] = ("b", L)
    stamps = {}

    def step_cb(step):
        if step in edges:
            eng.sync()
            stamps[(edges[step][0], edges[step][1])] = time.perf_counter()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(tokenizer=T, cfg_scale=args.cfg_scale, generation_config={"do_sample": False}, max_new_tokens=n_steps,
                         show_progress_bar=False, _forced_tokens=forced, _noise_fn=lambda step, n2: noise_bank[step],
                         _bench_hooks=BenchHooks(step_callback=step_cb), **inputs)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    audio_s = out.speech_outputs[0].shape[-1] / 24000.0
    free1, _ = torch.cuda.mem_get_info(device)
    run_stats = dict(model.last_stats)                              # of THIS call (the first-audio trials below overwrite it)
    n_graphs, n_launch_last = eng.stat(1), eng.stat(0)
    blocks_kept = len([b for b in model._audio_blocks if b is not None])
    windows = []
    for L in targets:
        if ("a", L) in stamps and ("b", L) in stamps:
            ms = (stamps[("b", L)] - stamps[("a", L)]) / WIN * 1e3
            by = algorithmic_bytes_per_frame(cfg, NS, L + WIN // 2, 75)
            windows.append({"kv_len": L, "steps": WIN, "ms_per_step": round(ms, 4), "formula_bytes_per_step": round(by, 1),
                            "whole_step_frac_of_8TBps": round(by / 1e9 / (ms / 1e3) / HBM_PEAK_GBS, 4)})
    # first-audio latency of the same request (warm process), as in the default line
    lat = first_audio_trials(lambda st: model.generate(
        tokenizer=T, cfg_scale=args.cfg_scale, generation_config={"do_sample": False}, max_new_tokens=3, show_progress_bar=False,
        _forced_tokens=forced, _noise_fn=lambda step, n2: noise_bank[step], audio_streamer=st, **inputs), 3)
    res = {"metric": "audio-sec/wall-sec", "value": round(audio_s / wall, 3), "unit": "audio-s/wall-s", "n_gpus": 1,
           "steps": int(run_stats.get("steps", n_steps)), "warmup": 0, "ms_per_step": round(wall / max(1, run_stats.get("steps", n_steps)) * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE {spec.get('baseline_config', '?')}, ONE WHOLE UTTERANCE: VibeVoice-{model_key.upper()} shapes, {spec['speakers']} speaker(s), "
                                  f"{L0}-token prompt, {n_steps} generated tokens ({n_frames_plan} speech frames, turns of 150), KV grows {L0} -> {L0 + n_steps}, "
                                  f"{NS} solver steps, cfg {args.cfg_scale}; wall = whole generate() incl. voice-prompt encode + prompt prefill",
                      "model": f"VibeVoice-{model_key}", "solver_steps": NS, "prompt_tokens": L0, "generated_tokens": n_steps,
                      "xsplit": args.xsplit, "hipgraph": not args.no_graph},
           "roofline": None, "cpu_baseline": None,
           "extra": {"utterance_audio_s": round(audio_s, 2), "utterance_wall_s": round(wall, 3), "frames": int(run_stats.get("frames", 0)),
                     "ms_per_step_by_context": windows,
                     "first_audio_ms": [round(x, 2) for x in lat],
                     "device_memory_in_use_GB": round((total_mem - free1) / 2 ** 30, 2),
                     "device_memory_before_engine_GB": round((total_mem - free0) / 2 ** 30, 2),
                     "torch_peak_allocated_GB": round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 2),
                     "kv_cache_GB": round(2 * 2 * d["num_hidden_layers"] * d["num_key_value_heads"] * (d["hidden_size"] // d["num_attention_heads"]) * 2 * max_ctx / 2 ** 30, 2),
                     "hipgraph_cache_entries": n_graphs, "launches_last_engine_call": n_launch_last,
                     "frame_store_blocks_kept_after_the_call": blocks_kept,
                     "reach_max_step_sample": bool(out.reach_max_step_sample[0]), "libvvhip_build_id": _build_id()}}
    eng.close()
    return res


def streaming_bytes_per_frame(cfg, n_solver, kv_len):
    """SURVEY 8(d), Streaming-0.5B: the TTS LM's layers once per frame (positive + negative rows share the pass), the head once per
    solver step, the acoustic decoder, connector and EOS classifier once, one text window of the text LM + TTS LM per 6 frames, and
    the TTS KV of both branches."""
    from vibevoice_amd.synthetic import streaming_param_shapes
    import math
    sh = streaming_param_shapes(cfg)
    cnt = lambda pre: sum(math.prod(v) for k, v in sh.items() if k.startswith(pre))
    d = cfg["decoder_config"]
    H = d["hidden_size"]
    p_tts = cnt("model.tts_language_model.layers.") + H
    p_lm = cnt("model.language_model.layers.")
    p_head = cnt("model.prediction_head.")
    p_dec = cnt("model.acoustic_tokenizer.decoder.")
    p_small = cnt("model.acoustic_connector.") + cnt("tts_eos_classifier.")
    n_tts = cfg["tts_backbone_num_hidden_layers"]
    kv_tok = 2 * n_tts * d["num_key_value_heads"] * (H // d["num_attention_heads"]) * 2
    return float(2 * p_tts + 2 * (p_head - H * H) * n_solver + 2 * H * H + 2 * p_dec + 2 * p_small + 2 * (p_lm + p_tts) / 6.0 + kv_tok * (kv_len + 1 + 6))


def bench_streaming(args, spec, ctx):
    """BASELINE.json configs[4]: Streaming-0.5B, hipGraph-captured decode+diffusion step, p50 first-audio latency.
    Synthetic weights and an Emma-shaped synthetic preset (lm 74 / tts_lm 251 cached positions)."""
    import statistics
    import types
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    from vibevoice_amd.modeling_streaming import VibeVoiceStreamingForConditionalGenerationInference
    device = ctx["device"]
    cfg = CONFIGS[spec["model"]]
    NS = spec.get("solver_steps") or 5                                 # the streaming demo default is 5
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    ckpt = find_checkpoint(spec["model"])
    if ckpt:
        with open(os.path.join(ckpt, "config.json")) as f:
            cfg = json.load(f)
        sd = checkpoint_tensors(ckpt)
    else:
        sd = ((k, synthetic.random_tensor(k, shp, gen, device, torch.bfloat16))
              for k, shp in synthetic.streaming_param_shapes(cfg).items())
    model = VibeVoiceStreamingForConditionalGenerationInference.from_state_dict(
        cfg, sd, torch.bfloat16, device, xsplit=args.xsplit, use_graph=not args.no_graph, max_ctx=2048, n_slots=2)
    if not ckpt:
        model.set_speech_factors(0.2, -0.05)
    model.set_ddpm_inference_steps(NS)
    model.engine.upload("eos.fc2.bias", torch.tensor([-30.0]))       # fixed-length runs: keep the EOS head from firing
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
    n_frames = audio_s / FRAME_SEC
    # two concurrent sessions over ONE weight copy (model.fork(): vv_create_shared), one host thread + stream each: the same steady-state
    # request in both at once; aggregate = both sessions' audio / the wall clock around both
    two = None
    if not os.environ.get("VVHIP_NO_TWO_SESSIONS"):
        try:
            import threading
            m2 = model.fork()
            sess = [model, m2]
            texts = [text, torch.randint(0, 151000, (1, n_text), generator=g)]
            res2 = [None, None]

            def run_session(k):
                torch.cuda.set_device(device)
                res2[k] = sess[k].generate(tts_text_ids=texts[k], all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok,
                                           max_new_tokens=n_text + (n_text // 5) * 6)
            m2.generate(tts_text_ids=text[:, :5], all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok, max_new_tokens=5 + 6)   # its graphs
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            th = threading.Thread(target=run_session, args=(1,))
            th.start()
            run_session(0)
            th.join()
            torch.cuda.synchronize()
            wall2 = time.perf_counter() - t2
            a2 = sum(r.speech_outputs[0].shape[-1] for r in res2) / 24000.0
            two = {"sessions": 2, "aggregate_audio_s_per_wall_s": round(a2 / wall2, 3), "single_session": round(audio_s / wall, 3),
                   "ratio": round((a2 / wall2) / (audio_s / wall), 3), "ms_per_frame_per_session": round(wall2 / (a2 / 2 / FRAME_SEC) * 1e3, 4),
                   "weights": "one copy (vv_create_shared)"}
            m2.engine.close()
            del m2
        except Exception as ex:
            two = {"error": repr(ex)[:200]}
    # SURVEY 8d's first-audio latency: generate() entry -> first chunk handed to an AudioStreamer consumer on the host
    text5 = torch.randint(0, 151000, (1, 5), generator=g)
    lat_host = first_audio_trials(lambda st: model.generate(tts_text_ids=text5, all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok,
                                                            max_new_tokens=5 + 6, audio_streamer=st), 30)
    # roofline: the GEMV launches and the attention units of one text window + one speech window, replayed as dependent chains
    roof = None
    if not args.no_roofline:
        eng = model.engine
        eng.sync()
        eng.profile_begin()
        model.generate(tts_text_ids=text5, all_prefilled_outputs=preset, cfg_scale=1.5, tokenizer=tok, max_new_tokens=5 + 6)
        (n_l, ms_cal, by), _ = eng.profile_end()
        n_rep, ms_rep, by_rep = eng.profile_replay(reps=3)
        ach = by_rep / 1e9 / (ms_rep / 1e3) if ms_rep > 0 else 0.0
        formula = streaming_bytes_per_frame(cfg, NS, 251 + int(n_frames) // 2)
        roof = {"bound": "hbm", "kernel": "vv_gemv_kernel", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(spec["model"], "vv_gemv_kernel", by_rep / max(1, n_rep))[0],
                "traffic_source": pmc_traffic(spec["model"], "vv_gemv_kernel", by_rep / max(1, n_rep))[1],
                "launches_per_step": round(n_l / 6.0, 1), "avg_launch_us": round(ms_rep * 1e3 / max(1, n_rep), 3),
                "bytes_per_launch": round(by_rep / max(1, n_rep), 1),
                "method": "vv_gemv_kernel launches of one generate() (text window of 5 + speech window of 6) replayed as one dependent hipGraph chain",
                "formula_bytes_per_step": round(formula, 1), "whole_step_GBps": round(formula / 1e9 / (wall / n_frames), 1),
                "whole_step_frac": round(formula / 1e9 / (wall / n_frames) / HBM_PEAK_GBS, 4)}
        n_a, ms_a, by_a = eng.profile_replay(reps=3, family=2)
        if n_a and ms_a > 0:
            roof["attention"] = {"kernel": "vv_attn_fused_kernel", "achieved": round(by_a / 1e9 / (ms_a / 1e3), 1), "frac": round(by_a / 1e9 / (ms_a / 1e3) / HBM_PEAK_GBS, 4),
                                 "avg_launch_us": round(ms_a * 1e3 / n_a, 3), "bytes_per_launch": round(by_a / n_a, 1)}
    res = {"metric": "audio-sec/wall-sec", "value": round(audio_s / wall, 3), "unit": "audio-s/wall-s", "n_gpus": 1,
           "steps": int(audio_s / FRAME_SEC + 0.5), "warmup": 33, "ms_per_step": round(wall / (audio_s / FRAME_SEC) * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
           "config": {"workload": f"BASELINE configs[4]: VibeVoice-Streaming-0.5B shapes, Emma-shaped synthetic preset (lm 74 / tts 251), {NS} solver steps, "
                                  "whole generate() incl. preset import, text windows of 5 / speech windows of 6",
                      "model": "VibeVoice-Streaming-0.5B", "solver_steps": NS, "hipgraph": not args.no_graph},
           "roofline": roof, "cpu_baseline": None,
           "extra": {"p50_first_audio_ms": round(statistics.median(lat_host), 3) if lat_host else None,
                     "p90_first_audio_ms": round(sorted(lat_host)[int(0.9 * len(lat_host))], 3) if lat_host else None,
                     "trials": len(lat_host), "first_audio_definition": "generate() entry -> first 3200-sample chunk delivered to an AudioStreamer "
                     "consumer thread on the host (preset KV import + first text window + first frame + D2H + queue hand-off); SURVEY 8d",
                     "p50_first_chunk_on_device_ms": round(statistics.median(lat), 3),
                     "weights_source": (f"checkpoint {ckpt}" if ckpt else "synthetic"), "libvvhip_build_id": _build_id(),
                     "two_sessions": two, "finished_by_eos_or_cap": True}}
    model.engine.close()
    del model
    torch.cuda.empty_cache()
    return res


def _oracle_leg(cfg, sd, n_solver, cfg_scale, n_frames, device, dtype, t_budget):
    """`n_frames` decode frames of oracle/ (the restatement of the reference loop, plain PyTorch ops) after a 48-token text-only
    prompt, on `device` in `dtype`, weights from the state dict `sd` (oracle/parity.py).  Returns the leg: seconds per frame,
    frames timed AND the trace (hidden states, latents, next-step embeddings, decoded frames) the parity check compares the
    HIP engine against."""
    from oracle import parity
    from vibevoice_amd import synthetic
    return parity.oracle_leg(cfg, sd, synthetic.TOKENS, n_solver, cfg_scale, n_frames, device, dtype, t_budget)


def _oracle_window(cfg, sd, n_solver, cfg_scale, n_frames, device, dtype, t_budget, kv_len):
    """Timing only: `n_frames` decode frames of the oracle loop on `device` in `dtype` AT A KV LENGTH OF `kv_len` -- a 48-token text-only
    prompt pass, then the positive branch's cache is padded with noise (N(0, 0.5^2), the scale bench_decode's kv_fill uses) up to kv_len, so
    every frame's attention, its softmax over kv_len positions and the DynamicCache-style torch.cat run at the length the GPU leg is timed at
    (the 10,922-token prompt pass itself would cost ~150 s of CPU time and is not what the window measures; the negative branch's cache is
    short in the real run too: it restarts at every <speech_start>).  Stops after t_budget seconds once two whole frames are in.
    Returns (seconds per frame, frames timed, first-interval seconds)."""
    from oracle import generate as ogen
    from oracle import parity as oparity
    from vibevoice_amd import synthetic
    d = cfg["decoder_config"]
    kvh, hd = d["num_key_value_heads"], d["hidden_size"] // d["num_attention_heads"]
    on_gpu = torch.device(device).type == "cuda"
    T = synthetic.TOKENS
    with torch.device(device):
        m = oparity.oracle_model(cfg, sd, device, dtype)
        m.t_cast_dtype = torch.bfloat16
        tok = ogen.TokenIds(T.speech_start_id, T.speech_end_id, T.speech_diffusion_id, T.eos_token_id, None, T.pad_token_id)
        plain_forward = m.lm.forward

        def forward(embeds, cache, final_norm=True):
            out = plain_forward(embeds, cache, final_norm)
            if embeds.shape[0] > 1 and cache.length < kv_len:        # the prompt pass of the positive branch: pad its cache with noise
                pad = kv_len - cache.length
                gk = torch.Generator(device="cpu").manual_seed(4321)
                for i in range(len(cache.k)):
                    cache.k[i] = torch.cat([cache.k[i], (torch.randn(kvh, pad, hd, generator=gk, device="cpu") * 0.5).to(device=device, dtype=dtype)], dim=1)
                    cache.v[i] = torch.cat([cache.v[i], (torch.randn(kvh, pad, hd, generator=gk, device="cpu") * 0.5).to(device=device, dtype=dtype)], dim=1)
                cache.length = kv_len
            return out
        m.lm.forward = forward
        g = torch.Generator(device="cpu").manual_seed(7)
        ids = torch.randint(0, min(151000, d["vocab_size"] - 64), (1, 48), generator=g, device="cpu")
        ids[0, -1] = T.speech_start_id
        stamps = []

        class _Budget(Exception):
            pass

        def noise_fn(step, n2):
            if on_gpu:
                torch.cuda.synchronize()
            stamps.append(time.perf_counter())
            if len(stamps) >= 3 and stamps[-1] - stamps[0] > t_budget:
                raise _Budget()
            return torch.randn(n2, 64, generator=g, device="cpu").to(device=device, dtype=dtype)
        ids_d = ids.to(device)
        try:
            with torch.no_grad():
                ogen.oracle_generate(m, tok, ids_d, torch.ones_like(ids_d), cfg_scale=cfg_scale, num_steps=n_solver, max_new_tokens=n_frames + 1,
                                     noise_fn=noise_fn, forced_tokens=[[T.speech_diffusion_id] * (n_frames + 1)])
        except _Budget:
            pass
        if on_gpu:
            torch.cuda.synchronize()
    first = 1 if (on_gpu and len(stamps) >= 4) else 0        # a GPU's first interval carries one-off costs (kernel selection, allocator growth)
    n = len(stamps) - 1 - first
    per = (stamps[-1] - stamps[first]) / max(1, n)
    del m
    return per, n, (stamps[1] - stamps[0]) if len(stamps) > 1 else 0.0


def cpu_baseline(cfg, cpu_sd, n_solver, cfg_scale, n_frames, model_key, kv_len=0):
    """Times oracle/ (the CPU restatement of the reference loop) on this host: `n_frames` decode
    frames, fp32, a bounded number of host threads.  kind = "port".  kv_len > 0: the window runs AT THAT KV LENGTH (the timed GPU
    leg's: _oracle_window) and the returned leg is None -- the parity block then takes its fp32 oracle run on the GPU."""
    # the GPU box advertises hundreds of logical CPUs but the job may be cgroup-limited; a modest
    # thread count keeps torch's intra-op pool from thrashing.  Measured on the box, ms per 7B frame: 16 threads 3150-3635,
    # 32 threads 3657, 64 threads 5958, 256 threads ~200,000: 16 is at the optimum, more cores do not make this baseline faster
    host_cpus = os.cpu_count() or 1
    ncpu = min(int(os.environ.get("VVHIP_CPU_THREADS", "16")), host_cpus)
    torch.set_num_threads(ncpu)
    t_budget = float(os.environ.get("VVHIP_CPU_BUDGET_S", "30"))
    if kv_len > 48:
        per_frame, n, _ = _oracle_window(cfg, cpu_sd, n_solver, cfg_scale, n_frames, "cpu", torch.float32, t_budget, kv_len)
        return None, {"value": round(FRAME_SEC / per_frame, 5), "unit": "audio-s/wall-s", "cores": ncpu, "kind": "port", "host_logical_cpus": host_cpus,
                      "kv_length": kv_len,
                      "sample": f"{n} decode frames AT A KV LENGTH OF {kv_len} (the GPU leg's timed context: the positive cache holds {kv_len} positions -- a 48-token "
                                f"prompt pass + noise K/V, as the GPU leg's window past its prompt does; cache growth by torch.cat as the reference's DynamicCache) of the "
                                f"same model shapes and weights (VibeVoice-{model_key}, fp32 = the reference's CPU dtype, {n_solver} solver steps, CFG pos+neg passes); "
                                f"oracle loop = CPU restatement of the reference's generate(), torch intra-op threads capped at {ncpu} of {host_cpus} logical CPUs "
                                f"(16 is the measured optimum on the box); the 16-frame windows at three KV lengths: bench.py --cpu-windows, profiles/r05_cpu_baseline.json",
                      "ms_per_step": round(per_frame * 1e3, 2)}
    leg = _oracle_leg(cfg, cpu_sd, n_solver, cfg_scale, n_frames, "cpu", torch.float32, t_budget)
    per_frame, n = leg.per_frame_s, leg.frames_timed
    return leg, {"value": round(FRAME_SEC / per_frame, 4), "unit": "audio-s/wall-s", "cores": ncpu, "kind": "port",
            "host_logical_cpus": host_cpus,
            "sample": f"{n} decode frames of the same model shapes and weights (VibeVoice-{model_key}, fp32 = the reference's CPU "
                      f"dtype, {n_solver} solver steps, CFG pos+neg passes) after a 48-token text-only prompt -- the GPU leg's 32K-token context is "
                      f"NOT reproduced here, which flatters the CPU: 16-frame windows of the same loop at KV 10.9K / 21.8K / 32.7K measure 4.7 / 6.0 / 8.1 s per "
                      f"7B frame (bench.py --cpu-windows, profiles/r05_cpu_baseline.json); oracle loop = CPU restatement of the reference's "
                      f"generate(), torch intra-op threads capped at {ncpu} of {host_cpus} logical CPUs",
            "ms_per_step": round(per_frame * 1e3, 2)}


def cpu_baseline_windows(args, spec, device):
    """SURVEY 8(d)'s CPU baseline as specified, run once per round (minutes of host time: not part of the default line): the oracle loop
    (CPU restatement of the reference's generate(), fp32 = the reference's CPU dtype) on the workload's model shapes and weights, a
    fixed window of `--cpu-frames` decode frames at three KV lengths -- the prompt length L0, the middle and the end of the utterance --
    with the positive cache pre-filled with noise up to that length right after a short prompt pass (the 10,922-token prompt pass itself
    would be ~150 s of CPU time per window and is not what the window measures).  The cache grows by torch.cat per step, as the
    reference's DynamicCache does.  Prints / returns one JSON object; profiles/r05_cpu_baseline.json is this output."""
    from oracle import generate as ogen
    from oracle import parity as oparity
    from vibevoice_amd import synthetic
    from vibevoice_amd.configs import CONFIGS
    model_key = spec["model"]
    cfg = CONFIGS[model_key]
    NS = spec["solver_steps"]
    d = cfg["decoder_config"]
    kvh, hd = d["num_key_value_heads"], d["hidden_size"] // d["num_attention_heads"]
    inputs = synthetic.synthetic_inputs(cfg, n_speakers=spec["speakers"], text_tokens=spec["text_tokens"], voice_frames=spec["voice_frames"], seed=100)
    L0 = inputs["input_ids"].shape[1]
    L_end = min(3 * L0, d.get("max_position_embeddings", 32768)) - args.cpu_frames - 2
    lengths = [L0, (L0 + L_end) // 2, L_end]
    host_cpus = os.cpu_count() or 1
    ncpu = min(int(os.environ.get("VVHIP_CPU_THREADS", "16")), host_cpus)
    torch.set_num_threads(ncpu)
    gen = torch.Generator(device=device)
    gen.manual_seed(0)
    sd = {k: synthetic.random_tensor(k, shp, gen, device, torch.bfloat16).to("cpu") for k, shp in synthetic.param_shapes(cfg).items()}
    m = oparity.oracle_model(cfg, sd, "cpu", torch.float32)
    sd.clear()
    m.t_cast_dtype = torch.bfloat16
    T = synthetic.TOKENS
    tok = ogen.TokenIds(T.speech_start_id, T.speech_end_id, T.speech_diffusion_id, T.eos_token_id, None, T.pad_token_id)
    target = {"L": 0}
    plain_forward = m.lm.forward

    def forward(embeds, cache, final_norm=True):
        out = plain_forward(embeds, cache, final_norm)
        if embeds.shape[0] > 1 and cache.length < target["L"]:        # the prompt pass of the positive branch: pad its cache with noise
            pad = target["L"] - cache.length
            g = torch.Generator().manual_seed(4321)
            for i in range(len(cache.k)):
                cache.k[i] = torch.cat([cache.k[i], torch.randn(kvh, pad, hd, generator=g) * 0.5], dim=1)
                cache.v[i] = torch.cat([cache.v[i], torch.randn(kvh, pad, hd, generator=g) * 0.5], dim=1)
            cache.length = target["L"]
        return out
    m.lm.forward = forward
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(0, 151000, (1, 48), generator=g)
    ids[0, -1] = T.speech_start_id
    windows = []
    for L in lengths:
        target["L"] = L
        stamps = []

        def noise_fn(step, n2):
            stamps.append(time.perf_counter())
            return torch.randn(n2, 64, generator=g)
        n = args.cpu_frames
        with torch.no_grad():
            ogen.oracle_generate(m, tok, ids, torch.ones_like(ids), cfg_scale=args.cfg_scale, num_steps=NS, max_new_tokens=n + 1,
                                 noise_fn=noise_fn, forced_tokens=[[T.speech_diffusion_id] * (n + 1)])
        per = (stamps[-1] - stamps[0]) / (len(stamps) - 1)
        windows.append({"kv_length": L, "frames": len(stamps) - 1, "s_per_frame": round(per, 3), "audio_s_per_wall_s": round(FRAME_SEC / per, 5)})
        print(f"[cpu window] KV {L}: {per:.3f} s/frame over {len(stamps) - 1} frames", file=sys.stderr, flush=True)
    return {"metric": "audio-sec/wall-sec", "kind": "port", "cores": ncpu, "host_logical_cpus": host_cpus, "dtype": "f32",
            "model": f"VibeVoice-{model_key}", "solver_steps": NS, "cfg_scale": args.cfg_scale, "windows": windows,
            "mean_audio_s_per_wall_s": round(sum(w["audio_s_per_wall_s"] for w in windows) / len(windows), 5),
            "sample": f"oracle loop (CPU restatement of the reference's generate()), synthetic weights at the model's shapes, {args.cpu_frames}-frame "
                      f"windows at KV lengths {lengths} (positive cache pre-filled with noise after a 48-token prompt pass), CFG pos + neg passes, "
                      f"torch intra-op threads {ncpu} of {host_cpus}"}


def gpu_eager_baseline(cfg, dev_sd, n_solver, cfg_scale, n_frames, model_key, device, kv_len=0):
    """SURVEY 8(d)'s "GPU before": the same oracle loop as plain PyTorch-ROCm eager ops in bf16 on the SAME GPU (what the
    reference's generate() issues per frame: ~2-3 k library kernels, a second weight pass for the CFG-negative row, the
    full-vocabulary lm_head, DynamicCache-style torch.cat of the KV cache).  A reported baseline like cpu_baseline: it is the
    restatement under eager PyTorch, not the reference's own classes (those do not travel to the GPU box)."""
    t_budget = float(os.environ.get("VVHIP_EAGER_BUDGET_S", "15"))
    # Two identical passes, the SECOND is the baseline.  The KV cache grows by torch.cat, so every frame presents new attention
    # shapes to the BLAS libraries, and in the first process of a fresh box each of them pulls code objects from a cold disk:
    # measured 212 ms/frame for the first pass against 104 for the same pass repeated (a second process in the same box also
    # runs at 104).  The warm figure is the honest "GPU before".
    if kv_len > 48:
        # at the timed KV length: the eager loop's attention over kv_len positions and its torch.cat of the whole cache per layer and step
        # are part of the frame, as they are in the reference on a GPU
        cold, _, _ = _oracle_window(cfg, dev_sd, n_solver, cfg_scale, n_frames, device, torch.bfloat16, t_budget, kv_len)
        per_frame, n, _ = _oracle_window(cfg, dev_sd, n_solver, cfg_scale, n_frames, device, torch.bfloat16, t_budget, kv_len)
        return None, {"value": round(FRAME_SEC / per_frame, 4), "unit": "audio-s/wall-s", "kind": "port, PyTorch-ROCm eager bf16, same GPU", "kv_length": kv_len,
                      "sample": f"{n} decode frames (after one untimed frame) AT A KV LENGTH OF {kv_len} (48-token prompt pass + noise K/V in the positive cache, "
                                f"grown by torch.cat per step) of the same model shapes and weights (VibeVoice-{model_key}, bf16, {n_solver} solver steps, CFG "
                                f"pos+neg passes), the second of two identical passes (the first one, which also loads the libraries' code objects: "
                                f"{cold * 1e3:.0f} ms/frame)",
                      "ms_per_step": round(per_frame * 1e3, 3), "first_pass_ms_per_step": round(cold * 1e3, 3)}
    cold = _oracle_leg(cfg, dev_sd, n_solver, cfg_scale, n_frames, device, torch.bfloat16, t_budget).per_frame_s
    leg = _oracle_leg(cfg, dev_sd, n_solver, cfg_scale, n_frames, device, torch.bfloat16, t_budget)
    per_frame, n = leg.per_frame_s, leg.frames_timed
    return leg, {"value": round(FRAME_SEC / per_frame, 4), "unit": "audio-s/wall-s", "kind": "port, PyTorch-ROCm eager bf16, same GPU",
            "sample": f"{n} decode frames (after one untimed frame) of the same model shapes and weights (VibeVoice-{model_key}, bf16, "
                      f"{n_solver} solver steps, CFG pos+neg passes) after a 48-token text-only prompt, the second of two identical "
                      f"passes (the first one, which also loads the libraries' code objects: {cold * 1e3:.0f} ms/frame): the timed leg's "
                      f"32K-token context is NOT reproduced (its attention + KV torch.cat would add to the eager frame, so this "
                      f"baseline is on the fast side)",
            "ms_per_step": round(per_frame * 1e3, 3), "first_pass_ms_per_step": round(cold * 1e3, 3)}


if __name__ == "__main__":
    main()
