"""Many concurrent streams: the batch driver of one GPU (`StreamBatch`) and the stream-parallel partitioning across the GPUs of
one node (replicas only).

Audio streams are independent; nothing inside one stream shards, so there is no data-path collective: every rank
holds a full weight replica and steps its own streams.  `torch.distributed` is used for rendezvous, the timing
barrier and the max-over-ranks reduction of the elapsed time only.

`StreamBatch` is what BASELINE.json configs[2] / [3] run: the per-chunk body of the reference's `policy()`
(agents/infinisst.py:287-367) for MANY independent streams with ONE `isst_generate` call per tick.  The reference has no such
thing -- its only "batching" replicates one stream N times (`pseudo_batch_size`, :291-301) -- so the class restates, per stream,
exactly the state one single-stream agent keeps: `speech_cache` / `past_key_values` (inside the library, by stream id),
`target_ids`, the `cache_checkpoints` list and the whole-chunk eviction (:337-361).
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


def assign_streams(n_streams: int, rank: int, world_size: int) -> List[int]:
    """gpu = stream_id mod n_gpu."""
    return [s for s in range(n_streams) if s % world_size == rank]


def _parse_cpulist(text: str) -> List[int]:
    cpus: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def _kfd_gpu_order(kfd: str) -> Optional[List[str]]:
    """PCI addresses (`dddd:bb:dd.f`) of the GPU agents in KFD topology order -- the order ROCr enumerates them in, hence the HIP device index
    of an unmasked process (the amdgpu cards sorted by PCI address need not be in that order).  Read from
    `<kfd>/topology/nodes/<n>/properties` (`simd_count` > 0: a GPU; `domain` and `location_id` = bus << 8 | device << 3 | function).
    None when the tree is absent or unreadable.  Touches no GPU."""
    import os
    nodes = os.path.join(kfd, "topology", "nodes")
    try:
        names = sorted((n for n in os.listdir(nodes) if n.isdigit()), key=int)
    except OSError:
        return None
    out: List[str] = []
    for n in names:
        props = {}
        try:
            with open(os.path.join(nodes, n, "properties")) as f:
                for line in f:
                    k, _, v = line.strip().partition(" ")
                    if v.strip().lstrip("-").isdigit():
                        props[k] = int(v)
        except OSError:
            return None  # (a node the process may not read: the order of the others is not the device order)
        if props.get("simd_count", 0) <= 0:
            continue
        loc = props.get("location_id")
        if loc is None:
            return None
        out.append("%04x:%02x:%02x.%x" % (props.get("domain", 0) & 0xFFFF, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
    return out or None


def local_cores_of_rank(local_rank: int, local_world: int, sysfs: str = "/sys/class/drm", allowed: Optional[Sequence[int]] = None,
                        kfd: str = "/sys/class/kfd/kfd") -> List[int]:
    """Host cores rank `local_rank` of `local_world` ranks on this node should run on: the cores local to its GPU's NUMA node (sysfs:
    the amdgpu cards' `local_cpulist`), shared evenly with the other ranks on that node; an even split of the allowed cores when sysfs
    does not answer (no GPUs here, a visibility mask, fewer cards than ranks).  HIP device i is the i-th GPU agent of the KFD topology
    (`_kfd_gpu_order`), so the cards are taken in THAT order; when the KFD tree is absent they are taken in PCI address order, and when
    it names a GPU that has no amdgpu card here the two views disagree and the even split is used.  The reference runs one GPU per SLURM
    array task (scripts/infer/infinisst.sh:5-13) and leaves placement to the scheduler; one process per GPU on one node has to do
    it itself -- eight Python hosts on the cores of one socket would time each other, not the GPUs.  Touches no GPU."""
    import os
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))
    local_world = max(1, local_world)

    def even_split(cpus: Sequence[int], idx: int, n: int) -> List[int]:
        per = max(1, len(cpus) // max(1, n))
        part = list(cpus[idx * per:(idx + 1) * per]) if idx < n - 1 else list(cpus[idx * per:])
        return part or list(cpus)

    cards = []
    masked = any(os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    try:
        for name in sorted(os.listdir(sysfs)) if not masked else []:
            dev = os.path.join(sysfs, name, "device")
            if not name.startswith("card") or "-" in name or not os.path.exists(os.path.join(dev, "local_cpulist")):
                continue
            with open(os.path.join(dev, "vendor")) as f:
                if f.read().strip() != "0x1002":
                    continue
            if not os.path.exists(os.path.join(dev, "mem_info_vram_total")):
                continue
            with open(os.path.join(dev, "local_cpulist")) as f:
                cards.append((os.path.basename(os.path.realpath(dev)), [c for c in _parse_cpulist(f.read()) if c in set(allowed)]))
    except OSError:
        cards = []
    cards.sort()
    order = _kfd_gpu_order(kfd) if cards else None
    if order is not None:
        by_bdf = dict(cards)
        cards = [(b, by_bdf[b]) for b in order] if all(b in by_bdf for b in order) else []
    if len(cards) >= local_world and 0 <= local_rank < len(cards) and cards[local_rank][1]:
        mine = cards[local_rank][1]
        peers = [i for i in range(local_world) if cards[i][1] == mine]
        return even_split(mine, peers.index(local_rank), len(peers))
    return even_split(allowed, local_rank % local_world, local_world)


def pin_rank_to_local_cores(local_rank: int, local_world: int) -> List[int]:
    """Set this process's affinity (before anything touches the GPU) and return the cores chosen."""
    import os
    # (ISST_SYSFS_DRM / ISST_SYSFS_KFD: other roots for the two sysfs trees -- containers that mount the host's elsewhere, and the launcher's CPU test)
    cores = local_cores_of_rank(local_rank, local_world, sysfs=os.environ.get("ISST_SYSFS_DRM", "/sys/class/drm"),
                                kfd=os.environ.get("ISST_SYSFS_KFD", "/sys/class/kfd/kfd"))
    try:
        os.sched_setaffinity(0, cores)
    except OSError:
        return sorted(os.sched_getaffinity(0))
    return cores


@dataclass
class _Slot:
    sid: int
    started: bool = False  # reference: states.speech_cache is not None
    target_ids: List[int] = field(default_factory=list)
    ckpts: List[int] = field(default_factory=list)  # cache_checkpoints of this stream's agent (:106, never reset per utterance)
    chunks: int = 0
    evictions: int = 0
    last_generated: List[int] = field(default_factory=list)  # every id sampled in the last chunk, the never-fed final one included
    # target_ids once more as int32 in a buffer of a few windows, so that the per-tick window handed to the library is a view, not a
    # list -> array conversion per stream (64 streams: that conversion was a fifth of the host time of a step)
    hist: Optional[np.ndarray] = None
    hist_n: int = 0

    def push_targets(self, ids: Sequence[int], lookback: int):
        k = len(ids)
        if self.hist is None:
            self.hist = np.zeros(4 * max(1, lookback) + 64, dtype=np.int32)
        if self.hist_n + k > self.hist.size:  # compact: only the last `lookback` ids are ever read
            keep = min(self.hist_n, lookback)
            tail = self.hist[self.hist_n - keep:self.hist_n].copy()
            if keep + k > self.hist.size:
                self.hist = np.zeros(keep + k + 4 * max(1, lookback), dtype=np.int32)
            self.hist[:keep] = tail
            self.hist_n = keep
        if k:
            self.hist[self.hist_n:self.hist_n + k] = ids
            self.hist_n += k

    def window(self, lookback: int) -> Optional[np.ndarray]:
        # lookback <= 0: NO encoder ids.  A deliberate deviation: the reference's `target_ids[-0:]` (agents/infinisst.py:298-300) is the WHOLE history --
        # an unbounded id list per stream and tick; every script of the reference passes 100, and the single-stream agent (agent.py) keeps the slice as is.
        if self.hist is None or self.hist_n == 0 or lookback <= 0:
            return None
        return self.hist[max(0, self.hist_n - lookback):self.hist_n]


class StreamBatch:
    """One GPU's concurrent streams, stepped together.

        batch = StreamBatch(engine, gen, system_prompt_size, prompt_fn)
        a = batch.open(); b = batch.open()                 # one library stream + one agent-side state each
        outs = batch.step([seg_a, None])                   # stream b brought no new chunk this tick: it is skipped
        outs[0] -> output_ids of stream a (generated[:-1], agents/infinisst.py:363); outs[1] is None

    * ONE `isst_generate` call per tick for every stream that brought a chunk (ragged prompts: a stream's first chunk carries the
      system prompt, later chunks the 22-token turn; the system prompt is pinned for fresh streams only);
    * per stream: `target_ids` (the last `no_repeat_ngram_lookback` feed the encoder-n-gram processor, :298-300), the
      checkpoint list and the whole-chunk eviction -- the same code path as `agent.InfiniSST.policy` (`evict_whole_chunks`);
    * all streams of a tick must bring the same number of samples (one latency multiplier per batch: `gen.latency_multiplier`);
    * `host_seconds` / `device_call_seconds`: wall time spent outside / inside the library call since `reset_timers()`."""

    def __init__(self, engine, gen, system_prompt_size: int, prompt_fn: Callable[[bool, int], List[int]]):
        self.engine, self.gen = engine, gen
        self.system_prompt_size = system_prompt_size
        self.prompt_fn = prompt_fn
        self.slots: List[Optional[_Slot]] = []
        self._prompt_cache = {}  # (first chunk?, multiplier) -> int32 array: prompt_fn is a pure function of the two
        self.reset_timers()

    def reset_timers(self):
        self.host_seconds = 0.0
        self.device_call_seconds = 0.0
        self.ticks = 0

    # ------------------------------------------------------------------ stream lifetime
    def open(self) -> int:
        slot = _Slot(sid=self.engine.open_stream())
        for i, s in enumerate(self.slots):
            if s is None:
                self.slots[i] = slot
                return i
        self.slots.append(slot)
        return len(self.slots) - 1

    def close(self, idx: int):
        self.engine.close_stream(self._slot(idx).sid)
        self.slots[idx] = None

    def new_utterance(self, idx: int):
        """S2TAgentStates.reset (:60-67): fresh speech cache / KV cache / target ids; the checkpoint list is agent-level and stays."""
        s = self._slot(idx)
        self.engine.reset_stream(s.sid)
        s.started, s.target_ids, s.hist_n = False, [], 0

    def _slot(self, idx: int) -> _Slot:
        if idx < 0 or idx >= len(self.slots) or self.slots[idx] is None:
            raise KeyError(f"no open stream {idx}")
        return self.slots[idx]

    def stream_id(self, idx: int) -> int:
        return self._slot(idx).sid

    def cache_len(self, idx: int) -> int:
        return self.engine.stream_info(self._slot(idx).sid)["llm_cache_len"]

    @property
    def evictions(self) -> int:
        return sum(s.evictions for s in self.slots if s is not None)

    def adopt_state(self, idx: int, ckpts: Sequence[int], started: bool = True):
        """A stream whose caches were imported (isst_stream_import_*: resume, steady-state set-up) also needs the agent-side half."""
        s = self._slot(idx)
        s.ckpts, s.started = list(ckpts), started

    def _prompt(self, first: bool, multiplier: int) -> np.ndarray:
        key = (bool(first), int(multiplier))
        got = self._prompt_cache.get(key)
        if got is None:
            got = self._prompt_cache[key] = np.asarray(self.prompt_fn(first, multiplier), dtype=np.int32)
        return got

    # ------------------------------------------------------------------ one tick
    def step(self, audio: Sequence, forced_tokens=None, return_logits: bool = False):
        """`audio[i]`: the new samples of open stream i (fp32 numpy array, or a contiguous fp32 CUDA tensor: resident audio), already
        padded to whole chunks (`agent._prepare_speech`), or None when stream i has nothing new.  Returns a list with, per stream,
        the output ids of this chunk (`sequences[0, len(prompt):-1]`) or None; with `return_logits` a pair (that list, logits of the
        active streams in call order)."""
        t0 = time.perf_counter()
        if len(audio) != len(self.slots):
            raise ValueError(f"{len(audio)} audio entries for {len(self.slots)} stream slots")
        active = [i for i, a in enumerate(audio) if a is not None and self.slots[i] is not None]
        outs: List[Optional[List[int]]] = [None] * len(self.slots)
        if not active:
            self.host_seconds += time.perf_counter() - t0
            return (outs, None) if return_logits else outs
        gen = self.gen
        slots = [self.slots[i] for i in active]
        prompts = [self._prompt(not s.started, gen.latency_multiplier) for s in slots]
        prevs = [s.window(gen.no_repeat_ngram_lookback) for s in slots]
        pin = self.system_prompt_size if gen.always_cache_system_prompt else 0
        forced = None if forced_tokens is None else [forced_tokens[i] for i in active]
        gens, logits = self.engine.generate(gen, [s.sid for s in slots], [audio[i] for i in active], prompts, prevs,
                                            system_prompt_size=pin, forced_tokens=forced, return_logits=return_logits)
        self.device_call_seconds += self.engine.last_call_seconds
        lens = self.engine.stream_cache_lens([s.sid for s in slots])
        keep = self.system_prompt_size if gen.always_cache_system_prompt else 0
        for i, s, g, cur in zip(active, slots, gens, lens):
            s.started = True
            s.chunks += 1
            s.last_generated = list(g)
            s.ckpts, new_size = evict_whole_chunks(s.ckpts, cur, gen.max_llm_cache_size, keep)
            if new_size is not None:
                self.engine.kv_evict(s.sid, new_size, keep)
                s.evictions += 1
            out = g[:-1]
            s.target_ids.extend(out)
            s.push_targets(out, gen.no_repeat_ngram_lookback)
            if len(s.target_ids) > 4 * max(1, gen.no_repeat_ngram_lookback):  # only the last `lookback` ids are ever read
                s.target_ids = s.target_ids[-gen.no_repeat_ngram_lookback:]
            outs[i] = out
        self.ticks += 1
        self.host_seconds += time.perf_counter() - t0 - self.engine.last_call_seconds
        return (outs, logits) if return_logits else outs


def evict_whole_chunks(ckpts: List[int], cur: int, max_llm_cache_size: int, keep_prefix: int):
    """The checkpoint walk of agents/infinisst.py:337-352 for one stream: append the cache length after this chunk; when it exceeds
    the budget find the first checkpoint c with cur - c <= budget, drop the checkpoints up to it and rebase the rest (the pinned
    system prompt is not counted as trimmed).  Returns (new checkpoint list, tail size for isst_kv_evict or None)."""
    ckpts = list(ckpts) + [cur]
    if cur <= max_llm_cache_size:
        return ckpts, None
    new_size = 0
    for i, c in enumerate(ckpts):
        new_size = cur - c
        if new_size <= max_llm_cache_size:
            n_trimmed = c - keep_prefix
            ckpts = [x - n_trimmed for x in ckpts[i + 1:]]
            break
    return ckpts, effective_new_cache_size(new_size, cur, keep_prefix)


def effective_new_cache_size(new_size: int, cur: int, keep_prefix: int) -> int:
    """Entries of the tail that survive `k[:, :, -new_size:]` (reference agents/infinisst.py:354-361) for ANY integer the checkpoint
    loop can produce.  `cache_checkpoints` is agent-level and never reset (:106), so after a new utterance starts the list holds
    stale, possibly larger-than-`cur` entries and `new_size` can leave [0, cur - keep_prefix]:
      * 0 < new_size <= cur - keep_prefix: the ordinary case;
      * new_size < 0: Python slicing `-new_size:` = `[|new_size|:]` keeps the last cur - |new_size| entries;
      * a tail that would overlap the pinned prefix (the reference then DUPLICATES prefix entries behind the prefix -- the cache
        grows with repeated keys) is clamped to the evictable range: nothing is evicted, nothing is duplicated (deliberate);
      * new_size == 0 (one chunk longer than the whole budget): `-0:` keeps everything in the reference, again duplicating the
        prefix; here the tail is dropped, as the budget asks (deliberate; same as oracle/agent.py)."""
    evictable = max(0, cur - keep_prefix)
    if new_size < 0:
        new_size = max(0, cur + new_size)
    return min(new_size, evictable)


class TimingGroup:
    """The only cross-rank traffic of a run: the barrier around the timed region and the max / sum / gather of python floats.
    RCCL (backend "nccl") on device tensors is the normal form.  Whether it works on this node is decided ONCE, COLLECTIVELY, at
    construction: every rank tries one RCCL all-reduce, the ranks then agree over a gloo side group (MIN of their "it worked" flags),
    and if any rank failed ALL ranks use gloo on CPU tensors from then on -- the ranks can never sit in different collectives -- and the
    failure is REPORTED (`describe()`), not hidden.  (IPC / topology trouble has nothing to do with the stream-parallel data path, which
    has no collective.)  Without a gloo side group RCCL stays the only path and an error propagates."""

    def __init__(self, device=None, probe_timeout_s: float = 120.0):
        self.device = device
        self.gloo = None
        self.gloo_error = None
        self.rccl_error = None
        self.use_rccl = device is not None
        if self._active() and device is not None:
            try:
                self.gloo = dist.new_group(backend="gloo")
            except Exception as e:  # pragma: no cover
                self.gloo_error = f"{type(e).__name__}: {e}"
            # The probe must not be able to hang the ranks in DIFFERENT collectives: a rank whose all-reduce raises at launch (communicator / IPC trouble)
            # would otherwise go on to the gloo agreement while its peers sit inside the RCCL collective forever.  So (1) the all-reduce is launched
            # asynchronously and nobody waits for it yet, (2) the ranks agree over gloo whether every LAUNCH worked -- all of them reach this point --
            # and only then (3) wait for completion, on the host, with a deadline, and (4) agree on the outcome.  A pending RCCL operation that is
            # given up is never waited for.
            work, t, launched = None, None, 1.0
            try:
                t = torch.ones(1, dtype=torch.float64, device=device)
                work = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
            except Exception as e:
                if self.gloo is None:
                    raise
                launched, self.rccl_error = 0.0, f"{type(e).__name__}: {e}"
            if self.gloo is not None and not self._agree(launched, "an RCCL all-reduce could not be launched on another rank"):
                return
            ok = 1.0
            try:
                deadline = time.monotonic() + probe_timeout_s
                while work is not None and not work.is_completed() and time.monotonic() < deadline:
                    time.sleep(0.005)
                if work is not None and not work.is_completed():
                    if self.gloo is None:
                        raise RuntimeError(f"the RCCL probe all-reduce did not complete within {probe_timeout_s:.0f} s")
                    ok, self.rccl_error = 0.0, f"the RCCL probe all-reduce did not complete within {probe_timeout_s:.0f} s"
                else:
                    if work is not None:
                        work.wait()
                    ok = 1.0 if float(t.item()) == float(dist.get_world_size()) else 0.0
                    if ok < 1.0:
                        self.rccl_error = "the RCCL probe all-reduce returned a wrong sum"
            except Exception as e:
                if self.gloo is None:
                    raise
                ok, self.rccl_error = 0.0, f"{type(e).__name__}: {e}"
            if self.gloo is not None:
                self._agree(ok, "an RCCL all-reduce failed on another rank")

    def _agree(self, mine: float, why_other: str) -> bool:
        """MIN of the ranks' flags over the gloo side group; any 0 switches EVERY rank to gloo.  Returns whether RCCL stays in use."""
        flag = torch.tensor([mine], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.gloo)
        if float(flag.item()) < 1.0:
            self.use_rccl = False
            self.rccl_error = self.rccl_error or why_other
            return False
        return True

    def _active(self) -> bool:
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _reduce(self, values: Sequence[float], op) -> List[float]:
        if not self._active():
            return [float(v) for v in values]
        if self.use_rccl:
            t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
            dist.all_reduce(t, op=op)
            return [float(x) for x in t.tolist()]
        t = torch.tensor(list(values), dtype=torch.float64)
        dist.all_reduce(t, op=op, group=self.gloo)  # (device None: the default group is the CPU one)
        return [float(x) for x in t.tolist()]

    def barrier(self):
        self._reduce([0.0], dist.ReduceOp.SUM)

    def all_ok(self, ok: bool) -> bool:
        """Whether EVERY rank says ok (a MIN over the ranks, on the host: the gloo side group when there is one).  A leg whose set-up can fail on one
        rank (engine build, import, warm-up) asks this before its first timed barrier, so that a local failure makes all ranks skip the leg together
        instead of leaving the healthy ones inside a collective until the launcher's timeout."""
        if not self._active():
            return bool(ok)
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
        if self.gloo is not None or self.device is None:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.gloo)
            return float(flag.item()) >= 1.0
        return self._reduce([float(flag.item())], dist.ReduceOp.MIN)[0] >= 1.0

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
        if self.use_rccl:
            return "rccl" + (f" (no gloo side group: {self.gloo_error})" if self.gloo_error else "")
        return f"gloo (RCCL unusable on this node: {self.rccl_error})"


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
