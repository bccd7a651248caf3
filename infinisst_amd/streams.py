"""Stream-parallel partitioning across the GPUs of one node (replicas only).

Audio streams are independent; nothing inside one stream shards, so there is no data-path collective: every rank
holds a full weight replica and steps its own streams.  `torch.distributed` is used for rendezvous, the timing
barrier and the max-over-ranks reduction of the elapsed time only.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def assign_streams(n_streams: int, rank: int, world_size: int) -> List[int]:
    """gpu = stream_id mod n_gpu."""
    return [s for s in range(n_streams) if s % world_size == rank]


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a python float (gloo on CPU tensors, RCCL on GPU tensors)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_floats(values: Sequence[float], device=None) -> List[float]:
    """All ranks' per-step latencies on every rank (equal lengths required)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64, device=device or "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(x) for o in out for x in o.tolist()]


def whole_job_xrt(audio_seconds_this_rank: float, elapsed_this_rank: float, device=None) -> float:
    """audio seconds processed by ALL ranks / max-over-ranks wall time."""
    return sum_over_ranks(audio_seconds_this_rank, device) / max_over_ranks(elapsed_this_rank, device)
