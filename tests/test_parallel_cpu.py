"""world_size-2 gloo tests of the multi-GPU host path: packed-blob weight broadcast, utterance sharding,
per-rank continuous decoding through the product's host loop (oracle-backed fake engine), audio gathered on rank 0."""
import socket
import types

import torch

TOK = types.SimpleNamespace(speech_start_id=301, speech_end_id=302, speech_diffusion_id=303, eos_token_id=304,
                            bos_token_id=None, pad_token_id=305)
CFGD = {"decoder_config": {"max_position_embeddings": 4096}, "diffusion_head_config": {"ddpm_num_inference_steps": 5},
        "acoustic_tokenizer_config": {"fix_std": 0.5, "std_dist_type": "gaussian"}}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model(n_slots):
    import fake_engine
    from test_oracle_golden import _oracle_small
    from vibevoice_amd.modeling import VibeVoiceForConditionalGenerationInference
    m = VibeVoiceForConditionalGenerationInference(CFGD, fake_engine.FakeEngine(_oracle_small(), n_slots=n_slots), model_dtype=torch.float32)
    m.set_speech_factors(0.2, -0.05)
    m.set_ddpm_inference_steps(5)
    m.concurrent_codecs = False
    return m


def _worker(rank, world, port, q):
    import contextlib
    import torch.distributed as dist
    import fake_engine
    from test_dropin_cpu import _requests
    from vibevoice_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    # ---- packed-blob broadcast: 3 tensors, two buckets forced by a tiny bucket size ----
    shapes = [("a.weight", (4, 6)), ("b.bias", (5,)), ("c.weight", (3, 2, 2)), ("d.weight", (300,))]

    def make(name, shape):
        assert dist.get_rank() == 0                      # only the source rank materialises weights
        return torch.randn(shape, generator=torch.Generator().manual_seed(len(name) + shape[0]))
    st = {}
    got = {k: v.clone() for k, v in parallel.broadcast_packed(shapes, make, "cpu", torch.float32, bucket_bytes=1024, stats=st)}
    checksum = float(sum(v.double().sum() for v in got.values()))
    # ---- sharded generation: 5 utterances over 2 ranks, each rank 2 slots ----
    reqs = _requests(5, 3)

    class MP:                                            # monkeypatch stand-in for cpu_cuda_shims
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    with fake_engine.cpu_cuda_shims(MP()):
        shst = {}
        res = parallel.generate_sharded(_model(2), reqs, gather_to=0, stats=shst, tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
    assert shst["utterances_per_rank"] == [3, 2] and sum(shst["load_per_rank"]) == sum(int(r["input_ids"].shape[-1]) for r in reqs)
    payload = None
    if res is not None:
        payload = [(r.sequences.tolist(), None if r.speech_outputs[0] is None else r.speech_outputs[0].double().sum().item(),
                    None if r.speech_outputs[0] is None else r.speech_outputs[0].shape[-1]) for r in res]
    q.put((rank, checksum, st["collectives"], st["bytes"], [tuple(v.shape) for v in got.values()], payload))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_packed_broadcast_and_sharded_generation():
    import pytest
    import torch.multiprocessing as mp
    import fake_engine
    from test_dropin_cpu import _requests
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, n0, b0, sh0, pay0), (r1, c1, n1, b1, sh1, pay1) = res
    assert c0 == c1 and sh0 == sh1 == [(4, 6), (5,), (3, 2, 2), (300,)]            # identical weights on both ranks
    assert n0 == n1 == 2 and b0 == b1                                                # two buckets = two collectives, not four
    assert pay1 is None and pay0 is not None and len(pay0) == 5                      # audio comes back to rank 0 from BOTH ranks
    # reference: the same 5 utterances decoded in this process, one engine
    reqs = _requests(5, 3)

    class MP:
        def setattr(self, obj, name, val):
            setattr(obj, name, val)
    import contextlib
    saved = {n: getattr(torch.cuda, n) for n in ("Event", "Stream", "stream")}
    saved_pin = torch.Tensor.pin_memory
    try:
        with fake_engine.cpu_cuda_shims(MP()):
            solo = _model(2).generate_continuous(reqs, tokenizer=TOK, generation_config={"do_sample": False}, cfg_scale=1.3)
    finally:
        for n, v in saved.items():
            setattr(torch.cuda, n, v)
        torch.Tensor.pin_memory = saved_pin
    from vibevoice_amd.parallel import shard_utterances
    shards = shard_utterances([int(r["input_ids"].shape[-1]) for r in reqs], 2)
    assert all(len(s) >= 2 for s in shards)                                          # both ranks really decoded something
    for (seq, asum, alen), s in zip(pay0, solo):
        assert seq == s.sequences.tolist()
        a = s.speech_outputs[0]
        assert (alen is None) == (a is None)
        if a is not None:
            assert alen == a.shape[-1] and abs(asum - a.double().sum().item()) <= 1e-3 * max(1.0, abs(asum))


def test_sharding_cost_model_with_unequal_prompts():
    """Longest-prompt-first over ranks (cost = prompt length, since max_steps = 2 x prompt length, modeling_vibevoice_inference.py:421):
    every utterance exactly once, the most loaded rank within Graham's LPT bound (4/3 - 1/(3m)) of the best possible makespan,
    and shard_report() states the loads / imbalance the job's scaling efficiency is bounded by."""
    from vibevoice_amd.parallel import shard_report, shard_utterances
    g = torch.Generator().manual_seed(8)
    for world in (2, 4, 8):
        for n in (world, 3 * world + 1, 40):
            costs = torch.randint(200, 11000, (n,), generator=g).tolist()
            shards = shard_utterances(costs, world)
            assert sorted(i for sh in shards for i in sh) == list(range(n))
            rep = shard_report(costs, shards)
            assert rep["load_per_rank"] == [float(sum(costs[i] for i in sh)) for sh in shards]
            lower = max(sum(costs) / world, max(costs))                  # no schedule ends before this
            assert max(rep["load_per_rank"]) <= lower * (4.0 / 3.0 - 1.0 / (3.0 * world)) + 1e-9
            assert abs(rep["imbalance_max_over_mean"] - max(rep["load_per_rank"]) / (sum(costs) / world)) < 1e-12
    # equal prompts on every rank (bench.py --gpus N): perfectly balanced
    assert shard_report([5000] * 8, shard_utterances([5000] * 8, 8))["imbalance_max_over_mean"] == 1.0


def test_bench_gpus_flag_spawns_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` outside torchrun must start N ranks itself (the flag was parsed and ignored):
    the command it runs is torch.distributed.run with one process per GPU and a 127.0.0.1 rendezvous, forwarding its own flags."""
    import importlib.util
    import os
    import subprocess
    import sys
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: (seen.update(cmd=cmd, env=env), 0)[1])
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    monkeypatch.delenv("RANK", raising=False)
    args = bench.parse_args(["--gpus", "4", "--steps", "7"])
    assert bench.respawn_ranks(args) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "7"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.respawn_ranks(args)


def _agg_worker(rank, world, port, q):
    """rank 1 finishes its shard early (fewer frames, shorter wall): the job's figure is sum(units) / max(wall), every rank's own
    step time arrives in rank order on every rank, and nobody waits for a collective the early rank never issues"""
    import time
    import torch.distributed as dist
    from vibevoice_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    frames, wall = (20, 0.40) if rank == 0 else (7, 0.10)            # rank 1: a short queue that drained early
    if rank == 0:
        time.sleep(0.3)                                              # ... and it reaches the aggregation long before rank 0
    tot, wmax = parallel.aggregate_throughput(frames, wall, "cpu")
    per_rank = parallel.per_rank_values(wall / max(1, frames) * 1e3, "cpu")
    st = {}
    list(parallel.broadcast_packed([("w", (8,))], lambda n, s: torch.ones(s), "cpu", torch.float32, stats=st))
    q.put((rank, tot, wmax, per_rank, st["collectives"], st["world"]))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregation_survives_a_rank_that_finishes_early():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agg_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, tot, wmax, per_rank, n_coll, world in res:
        assert tot == 27.0 and wmax == 0.40                          # whole job: all frames over the slowest rank's wall
        assert [round(v, 4) for v in per_rank] == [20.0, round(0.10 / 7 * 1e3, 4)]
        assert n_coll == 1 and world == 2
