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


class TimingGroup:
    """The only cross-rank traffic of a run: the barrier around the timed region and the max / sum / gather of python floats.
    RCCL (backend "nccl") on device tensors is the normal form; if an RCCL call fails on a node (IPC / topology trouble has nothing to do with
    the stream-parallel data path, which has no collective), the same reductions go over a gloo group on CPU tensors and the failure is REPORTED
    (`describe()`), not hidden."""

    def __init__(self, device=None):
        self.device = device
        self.gloo = None
        self.rccl_error = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and device is not None:
            try:
                self.gloo = dist.new_group(backend="gloo")
            except Exception as e:  # pragma: no cover
                self.rccl_error = None if self.gloo else f"gloo side group unavailable: {e}"

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _reduce(self, values: Sequence[float], op) -> List[float]:
        if not self._active():
            return [float(v) for v in values]
        if self.device is not None and self.rccl_error is None:
            try:
                t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
                dist.all_reduce(t, op=op)
                return [float(x) for x in t.tolist()]
            except Exception as e:
                if self.gloo is None:
                    raise
                self.rccl_error = f"{type(e).__name__}: {e}"
        t = torch.tensor(list(values), dtype=torch.float64)
        dist.all_reduce(t, op=op, group=self.gloo)
        return [float(x) for x in t.tolist()]

    def barrier(self):
        self._reduce([0.0], dist.ReduceOp.SUM)

    def max(self, value: float) -> float:
        return self._reduce([value], dist.ReduceOp.MAX)[0]

    def sum(self, value: float) -> float:
        return self._reduce([value], dist.ReduceOp.SUM)[0]

    def gather(self, values: Sequence[float]) -> List[float]:
        """all ranks' values on every rank (equal lengths): a SUM of one-hot-placed vectors."""
        if not self._active():
            return list(values)
        n, w, r = len(values), dist.get_world_size(), dist.get_rank()
        flat = [0.0] * (n * w)
        flat[r * n:(r + 1) * n] = list(values)
        return self._reduce(flat, dist.ReduceOp.SUM)

    def describe(self) -> str:
        if not self._active():
            return "single rank"
        if self.device is None:
            return "gloo"
        return "rccl" if self.rccl_error is None else f"gloo (an RCCL call failed: {self.rccl_error})"


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
