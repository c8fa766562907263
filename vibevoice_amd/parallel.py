"""Multi-GPU host logic: one process per GPU, utterances sharded across ranks, ONE
collective phase at start-up (weight broadcast from rank 0 over RCCL/xGMI) and none in
the step loop (utterances are independent -- own KV caches, conv states,
RNG and token stream; speaker turns inside an utterance share KV and are never split).
Works with any torch.distributed backend ("nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
from typing import Callable, Iterable, Iterator, List, Sequence, Tuple

import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_utterances(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of utterance indices to ranks
    (cost ~ prompt length, since max_steps is proportional to it:
    modeling_vibevoice_inference.py:421).  Deterministic; every index appears once."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(i)
        load[r] += costs[i]
    for r in range(world):
        out[r].sort()
    return out


def broadcast_params(shapes: Iterable[Tuple[str, tuple]], make: Callable[[str, tuple], torch.Tensor],
                     device, dtype=torch.bfloat16, src: int = 0) -> Iterator[Tuple[str, torch.Tensor]]:
    """Yields (name, tensor) on every rank; rank `src` materialises each tensor with
    make(name, shape) and the others receive it by broadcast (one tensor in flight at a
    time, so the 18.7 GB 7B bundle never needs a second copy)."""
    rank, world = world_info()
    for name, shape in shapes:
        if rank == src:
            t = make(name, shape).to(device=device, dtype=dtype).contiguous()
        else:
            t = torch.empty(shape, dtype=dtype, device=device)
        if dist.is_available() and dist.is_initialized():     # also with a single rank: exercises the RCCL path
            dist.broadcast(t, src=src)
        yield name, t


def aggregate_throughput(units_local: float, wall_local: float, device) -> Tuple[float, float]:
    """Whole-job (sum of units over ranks) / (max wall over ranks)."""
    rank, world = world_info()
    t = torch.tensor([units_local, wall_local], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        s = t.clone()
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        m = t.clone()
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        return float(s[0]), float(m[1])
    return float(t[0]), float(t[1])


def per_rank_values(x: float, device) -> List[float]:
    """every rank's own scalar (its ms per step, its frame count), in rank order, on every rank: one small all_gather"""
    rank, world = world_info()
    if not (dist.is_available() and dist.is_initialized()):
        return [float(x)]
    parts = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(parts, torch.tensor([float(x)], dtype=torch.float64, device=device))
    return [float(t[0]) for t in parts]


def broadcast_packed(shapes: Iterable[Tuple[str, tuple]], make: Callable[[str, tuple], torch.Tensor], device,
                     dtype=torch.bfloat16, src: int = 0, bucket_bytes: int = 2 << 30, stats: dict = None
                     ) -> Iterator[Tuple[str, torch.Tensor]]:
    """The start-up collective of SURVEY 8(e): the parameter bundle travels as packed blobs (the 7B bundle, 18.7 GB in
    bf16, is ten 2 GiB ncclBroadcasts over the xGMI links of rank 0) instead of ~1000 per-tensor collectives, each of which
    pays the RCCL launch + ring set-up latency.  Rank `src` writes make(name, shape) into its slice of the blob; every rank
    then yields (name, view-into-the-blob).  A view is valid until the generator advances to the next bucket (the engine's
    vv_upload re-packs it into its own storage at once).  `stats` receives bytes / seconds / number of collectives.
    bucket_bytes bounds the transient copy and keeps every collective's element count below 2^31 (default 2 GiB: 2^30 bf16
    elements; a tensor larger than a bucket travels alone)."""
    import time
    rank, world = world_info()
    esz = torch.empty(0, dtype=dtype).element_size()
    items = [(n, tuple(s)) for n, s in shapes]
    use_dist = dist.is_available() and dist.is_initialized()
    is_cuda = torch.device(device).type == "cuda"
    tot_bytes, tot_s, n_coll = 0, 0.0, 0
    i = 0
    while i < len(items):
        # ---- one bucket: consecutive tensors, each slice 256-byte aligned ----
        offs, j, cur = [], i, 0
        while j < len(items):
            ne = 1
            for d in items[j][1]:
                ne *= d
            nb = (ne * esz + 255) // 256 * 256
            if offs and cur + nb > bucket_bytes:
                break
            offs.append((cur // esz, ne))
            cur += nb
            j += 1
        blob = torch.empty(cur // esz, dtype=dtype, device=device)
        if rank == src:
            for (name, shape), (o, ne) in zip(items[i:j], offs):
                blob[o:o + ne].view(shape).copy_(make(name, shape))
        if use_dist:
            if is_cuda:
                torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            dist.broadcast(blob, src=src)
            if is_cuda:
                torch.cuda.synchronize(device)
            tot_s += time.perf_counter() - t0
            n_coll += 1
        tot_bytes += cur
        for (name, shape), (o, ne) in zip(items[i:j], offs):
            yield name, blob[o:o + ne].view(shape)
        del blob
        i = j
    if stats is not None:
        stats.update(bytes=tot_bytes, seconds=tot_s, collectives=n_coll, world=world,
                     GBps=(tot_bytes / 1e9 / tot_s) if tot_s > 0 else None)


def shard_report(costs: Sequence[float], shards: Sequence[Sequence[int]]) -> dict:
    """What the assignment costs: per-rank load under the cost model (prompt length), and the imbalance max / mean that bounds
    the scaling efficiency of the whole job (every rank decodes at the same rate; the job ends with the most loaded rank)."""
    load = [float(sum(costs[i] for i in sh)) for sh in shards]
    mean = sum(load) / max(1, len(load))
    return {"utterances_per_rank": [len(sh) for sh in shards], "load_per_rank": load,
            "imbalance_max_over_mean": (max(load) / mean) if mean > 0 else 1.0}


def _gather_rows(rows: List[torch.Tensor], dtype, device, gather_to):
    """Variable-length 1-D tensors of every rank -> (on the receiving ranks) the list of every rank's rows, through ONE tensor
    collective per call: lengths travel as a small int64 all_gather, payloads as one padded buffer per rank (dist.gather /
    all_gather of tensors -- no pickling, no per-object CPU staging; on RCCL the buffers stay on the GPU)."""
    rank, world = world_info()
    n_max = torch.tensor([len(rows)], dtype=torch.int64, device=device)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    if int(n_max) == 0:                 # nothing anywhere (an empty request list): no zero-length collective
        return [[] for _ in range(world)] if (gather_to is None or rank == gather_to) else None
    lens_cpu = torch.zeros(int(n_max), dtype=torch.int64)
    for j, r in enumerate(rows):
        lens_cpu[j] = r.numel()
    lens = lens_cpu.to(device)         # one copy, not one tiny device write per row
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    width = max(1, max(int(l.sum()) for l in all_lens))
    buf = torch.zeros(width, dtype=dtype, device=device)
    o = 0
    for r in rows:
        buf[o:o + r.numel()] = r.reshape(-1).to(device=device, dtype=dtype)
        o += r.numel()
    if gather_to is None:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
    else:
        parts = [torch.empty_like(buf) for _ in range(world)] if rank == gather_to else None
        dist.gather(buf, parts, dst=gather_to)
        if rank != gather_to:
            return None
    out = []
    for p, l in zip(parts, all_lens):
        o, rr = 0, []
        for n in l.tolist():
            rr.append(p[o:o + n])
            o += n
        out.append(rr)
    return out


def generate_sharded(model, requests: Sequence[dict], gather_to: int = 0, stats: dict = None, lanes: int = 1, **gen_kwargs):
    """Utterance-parallel generation over the ranks of the current process group (BASELINE config 4): the
    requests (single-utterance processor outputs, as for model.generate_continuous) are assigned to ranks by
    longest-prompt-first (max_steps ~ prompt length, modeling_vibevoice_inference.py:421), every rank decodes its own shard
    with continuous batching on ITS GPU -- no collective inside the step loop -- and the finished utterances come back ONCE at
    the end as tensor collectives: one padded int64 buffer of token sequences and one padded buffer of waveforms per rank (a
    45-minute utterance is 65 M samples: gathered as a tensor it never passes through pickle or a CPU staging copy; on RCCL it
    goes GPU to GPU over xGMI).  Returns, on rank `gather_to` (every rank if gather_to is None), the list of
    VibeVoiceGenerationOutput in request order; None elsewhere.  `stats` receives the sharding report (shard_report).
    lanes > 1: every rank decodes its shard through model.generate_interleaved (that many engine contexts over the rank's ONE
    weight copy, a host thread and stream each): 1.6 x per GPU at the small models' shapes; at 7B only once a context's 8 slots are full (+13 %)."""
    from .modeling import VibeVoiceGenerationOutput
    rank, world = world_info()
    costs = [int(r["input_ids"].shape[-1]) for r in requests]
    shards = shard_utterances(costs, world)
    if stats is not None:
        stats.update(shard_report(costs, shards))
    mine = shards[rank]
    if mine and lanes > 1:
        outs = model.generate_interleaved([requests[i] for i in mine], lanes=lanes, **gen_kwargs)
    else:
        outs = model.generate_continuous([requests[i] for i in mine], **gen_kwargs) if mine else []
    if not (dist.is_available() and dist.is_initialized()):
        # no process group at all: a plain single-process call.  With a group -- a ONE-rank group included -- the results always go
        # through the collectives below: the path an 8-GPU job takes is the path every test and every 1-GPU run takes
        res = [None] * len(requests)
        for i, o in zip(mine, outs):
            audio = o.speech_outputs[0] if o.speech_outputs else None
            res[i] = VibeVoiceGenerationOutput(sequences=o.sequences.cpu(), speech_outputs=[None if audio is None else audio.float().cpu()],
                                               reach_max_step_sample=o.reach_max_step_sample.cpu())
        return res
    return gather_outputs(outs, shards, len(requests), model.device, model.dtype, gather_to=gather_to, stats=stats)


def gather_outputs(outs, shards: Sequence[Sequence[int]], n_requests: int, device, model_dtype, gather_to: int = 0, stats: dict = None):
    """The result phase of generate_sharded, on its own (bench.py --gpus N runs it after the timed region): every rank hands in the
    VibeVoiceGenerationOutput objects of ITS shard (in shard order); rank `gather_to` (every rank if None) gets the list of all
    n_requests outputs in request order, the others None.  Three tensor collectives per call -- token sequences (int64), flags (int64),
    waveforms (the model dtype on RCCL, device to device; fp32 on gloo) -- each preceded by the small all_gather of the row lengths.
    Requires an initialised process group (a one-rank group included)."""
    import time
    from .modeling import VibeVoiceGenerationOutput
    rank, world = world_info()
    on_gpu = dist.get_backend() == "nccl"
    device = device if on_gpu else torch.device("cpu")
    seqs, auds, flags = [], [], []
    for o in outs:
        audio = o.speech_outputs[0] if o.speech_outputs else None
        seqs.append(o.sequences.reshape(-1).to(torch.int64))
        auds.append(torch.zeros(0) if audio is None else audio.reshape(-1))
        flags.append(torch.tensor([int(bool(o.reach_max_step_sample.reshape(-1)[0])), 0 if audio is None else 1], dtype=torch.int64))
    if on_gpu:
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    aud_dtype = torch.float32 if not on_gpu else model_dtype
    g_seq = _gather_rows(seqs, torch.int64, device, gather_to)
    g_flag = _gather_rows(flags, torch.int64, device, gather_to)
    g_aud = _gather_rows(auds, aud_dtype, device, gather_to)
    if on_gpu:
        torch.cuda.synchronize(device)
    if stats is not None:
        payload = sum(int(t.numel()) * 8 for t in seqs + flags) + sum(int(t.numel()) for t in auds) * torch.empty(0, dtype=aud_dtype).element_size()
        stats.update(gather={"backend": dist.get_backend(), "ranks": world, "collective": "all_gather" if gather_to is None else "gather",
                             "payload_bytes_this_rank": payload, "seconds": round(time.perf_counter() - t0, 6), "on_device": bool(on_gpu),
                             "audio_dtype": str(aud_dtype).replace("torch.", "")})
    if g_seq is None:
        return None
    res = [None] * n_requests
    for r in range(world):
        for j, i in enumerate(shards[r]):
            has_audio = bool(g_flag[r][j][1])
            res[i] = VibeVoiceGenerationOutput(
                sequences=g_seq[r][j].reshape(1, -1).cpu(),
                speech_outputs=[g_aud[r][j].reshape(1, -1).float().cpu() if has_audio else None],
                reach_max_step_sample=torch.tensor([bool(g_flag[r][j][0])]))
    return res
