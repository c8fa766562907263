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
