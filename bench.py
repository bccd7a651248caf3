#!/usr/bin/env python3
"""Benchmark of the InfiniSST hot path on MI355X: xRT (audio-seconds per wall-second), whole job.

    python bench.py                                   # 1 GPU
    python bench.py --gpus N --steps K --warmup W     # N GPUs: starts the N ranks itself (torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # ... or is started as the ranks (the driver's form)

A "step" is one 960 ms chunk of every stream of this rank through the whole per-chunk path (the body of the
reference's policy(), agents/infinisst.py:287-361): H2D of the chunk, conv extractor, streaming encoder, shrink +
projector, Llama-3.1-8B prefill of the 22-token chunk prompt and greedy decode, then the chunk-wise KV eviction.
Workload = BASELINE.json configs[1]: InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 1 stream
per GPU, synthetic 16 kHz audio, random-init weights; EOS is disabled so that every chunk
runs the worst case max_new_tokens = 10 forward passes (1 prefill + 9 decode steps).

Steady state by construction, whatever --warmup is: before the first step every stream's state is IMPORTED
(isst_stream_import_*): LLM KV = pinned system prompt + 31 chunks' worth of entries with the matching checkpoint list, so
every step runs over >= 1000 cached entries and ends with a whole-chunk eviction; encoder rings hold the full 576-frame
window (K = 624 per step); both rings start close to their physical end so they wrap inside the timed region.

N ranks = N independent replicas (stream-parallel, no collective: SURVEY 8(e)); global stream ids are dealt to the ranks
by streams.assign_streams; the barrier and the max/sum reductions of the timing are the only cross-rank traffic.

The streams of a rank are stepped by the PRODUCT's multi-stream driver (infinisst_amd/streams.py::StreamBatch: one isst_generate per tick, per-stream
target ids / checkpoints / whole-chunk eviction); this file only adds the synthetic audio, the imported steady state and the clock.

Rank 0 prints ONE JSON line.  `value` has the chunks' audio resident in HBM before the timed region (the task's contract); `host_audio` is the same loop
with every chunk handed over as host arrays (H2D inside the step: SURVEY 8(d)'s latency definition) measured right after it.  Extra objects: `roofline`
for the dominant kernel (the packed-weight skinny GEMM streaming Llama weights, HBM-bound) measured live with HIP events on the launch stream, with
`roofline.whole_step` = the whole chunk's algorithmic bytes over the step time; `host_ms_per_step` (wall time outside isst_generate); `cpu_baseline`
(the CPU oracle timed on this box's host cores over 4 steady-state chunks, N=1 only); and, N=1 only, two more configurations run in this process after the
timed region: `streams64` = 64 concurrent streams on the GPU (BASELINE.json configs[2]) with its own `host_audio`, `host_ms_per_step` and an `mfma` block
(the prefill's widest dense GEMM against the 2.5 PFLOP/s dense bf16 peak), and `beam4` = the reference's production decoding (`--beam 4`).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--spinup", type=int, default=24, help="untimed chunks before the warm-up steps (a fresh box's first second of GPU work is noisy)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent streams per GPU (configs[1] = 1)")
    ap.add_argument("--gen-tokens", type=int, default=10, help="max_new_tokens per chunk (production: 10 x multiplier)")
    ap.add_argument("--beam", type=int, default=1, help="num_beams (1 = greedy, the north-star mode; 4 = the reference's production setting)")
    ap.add_argument("--toy", action="store_true", help="toy dimensions (plumbing check, not a valid benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--host-audio", action="store_true", help="hand every chunk's samples over as host arrays (uploaded inside the step: the PCIe-inclusive "
                    "rate); default: the audio is resident in HBM before the timed region")
    ap.add_argument("--no-streams64", action="store_true", help="skip the 64-streams-per-GPU leg (configs[2]) that follows the timed region at N=1")
    ap.add_argument("--streams64-steps", type=int, default=16)
    ap.add_argument("--no-beam4", action="store_true", help="skip the num_beams = 4 leg (the reference's production decoding) that follows the timed region at N=1")
    ap.add_argument("--beam4-steps", type=int, default=16)
    ap.add_argument("--no-multipliers", action="store_true", help="skip the latency-multiplier table (m = 2, 3, 4 at num_beams 4) at N=1")
    ap.add_argument("--multiplier-steps", type=int, default=8)
    ap.add_argument("--no-streams64-beam4", action="store_true", help="skip the 64 streams x num_beams 4 leg (the reference's production decoding on configs[2]) at N=1")
    ap.add_argument("--streams64-beam4-steps", type=int, default=8)
    ap.add_argument("--host-audio-steps", type=int, default=16, help="steps of the PCIe-inclusive leg (chunks handed over as host arrays) after the timed region")
    ap.add_argument("--cold-start", action="store_true", help="do NOT import the steady state: streams start empty (first-chunk behaviour; then use --warmup >= 40)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = the fastest of 64 / 128 / all logical cores on a one-second GEMV probe)")
    ap.add_argument("--cpu-chunks", type=int, default=4, help="steady-state chunks the CPU baseline runs")
    ap.add_argument("--attn-target-wgs", type=int, default=0, help="profiling aid: isst_op_set_attn_tuning (0 = library default)")
    ap.add_argument("--gemm-tuning", type=int, default=0, help="profiling aid: isst_op_set_gemm_tuning(w, 0) (0 = library default; e.g. 900010 + variant: gemm_wide.hip's A/B variants)")
    ap.add_argument("--cpu-layers", type=int, default=32, help="Llama layers actually run by the CPU baseline (time is scaled to all layers)")
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself (0 = pick a free one)")
    ap.add_argument("--dry-run", action="store_true",
                    help="LAUNCHER SELF-TEST, no compute and no GPU: every rank sleeps instead of stepping the engine and the ranks meet over gloo; "
                         "the JSON line is marked dry_run and is not a measurement (tests/test_streams_gloo.py)")
    ap.add_argument("--dry-fail-step-rank", type=int, default=-1, help="--dry-run only: this rank raises inside the TIMED steps of the 64-streams-per-GPU leg (after the barrier): "
                    "every rank must leave the leg together and the job must end non-zero")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="--dry-run only: this rank reports a failed set-up of the 64-streams-per-GPU leg (every rank must then skip it)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: this process -- which never touches HIP -- starts
    `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` as a CHILD process (never exec: a process image
    swap after GPU initialisation takes the node down on this pool, and a child keeps the rule trivially true), lets the ranks
    write to the inherited stdout/stderr (rank 0 prints the JSON line) and returns the launcher's exit code."""
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _a = parse()
    if _a.gpus > 1:  # before torch is imported: the parent only waits for its children
        sys.exit(spawn_ranks(_a))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from infinisst_amd import synth  # noqa: E402
from infinisst_amd import streams as S  # noqa: E402
from infinisst_amd.config import GenConfig, full_config, toy_config  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured with a float4 copy)


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def max_prompt_len(sys_n, multiplier=1):
    return sys_n + 20 + 12 * multiplier  # the first chunk's prompt (system prompt + a 9 + 12 m token turn) with slack; sys_n + 32 at m = 1


def build_engine(cfg, n_streams, gen_tokens, device, beams=1, weights=None, multiplier=1):
    from infinisst_amd.engine import Engine
    sys_n = len(synth.system_prompt_ids(cfg))
    eng = Engine(cfg, max_streams=n_streams, max_multiplier=multiplier, max_prompt_len=max_prompt_len(sys_n, multiplier), max_new_tokens=max(gen_tokens, 10),
                 max_llm_cache_size=1000, max_system_prompt=sys_n, max_beams=beams)
    log(f"engine created ({n_streams} stream slots)")
    if weights is None:
        weights = synth.random_weights_device(cfg, device)
        torch.cuda.synchronize()
        log("random weights drawn on device")
    eng.load_weights(weights)
    log("weights packed into the library")
    return eng, weights, sys_n


STEADY_CHUNKS = 31  # chunks' worth of LLM KV behind the system prompt in the imported steady state: 31 x (22 + 9) = 961 entries


class ChunkLoop:
    """The streams of one rank over `streams.StreamBatch` (the product's multi-stream driver: one isst_generate per tick, per-stream
    target ids / checkpoints / whole-chunk eviction); this class only adds the synthetic audio and the imported steady state."""

    def __init__(self, eng, cfg, gen, stream_ids, sys_n, host_audio=False):
        """`stream_ids`: GLOBAL stream ids of this rank (streams.assign_streams); they seed the audio.
        The audio of every stream is resident in HBM before the timed region (isst_gen_params.pcm_on_device) unless `host_audio`:
        then every chunk's samples are handed over as host arrays and uploaded inside the step, as the reference agent does."""
        self.eng, self.cfg, self.gen, self.sys_n = eng, cfg, gen, sys_n
        self.m = gen.latency_multiplier  # a step hands every stream m x 960 ms of audio (agents/infinisst.py:125-128)
        self.batch = S.StreamBatch(eng, gen, sys_n, lambda first, m: synth.chunk_prompt_ids(cfg, m, first=first))
        self.idx = [self.batch.open() for _ in stream_ids]
        self.sids = [self.batch.stream_id(i) for i in self.idx]
        n_chunks = 64 if self.m == 1 else 16
        self.audio = [synth.synthetic_audio(cfg.chunk_samples * self.m * n_chunks, stream_id=g) for g in stream_ids]
        self.audio_dev = None
        self.set_host_audio(host_audio)
        self.n_chunks = n_chunks
        self.c = 0

    def set_host_audio(self, host_audio: bool):
        """The chunks of every stream as ready-made segment objects (a served stream hands its chunk over as an array of its own: cutting
        64 views out of the synthetic clips is the benchmark's bookkeeping, not the path's, and stays out of the timed region)."""
        self.host_audio = host_audio
        if not host_audio and self.audio_dev is None:
            self.audio_dev = [torch.from_numpy(a).to("cuda") for a in self.audio]
        cs = self.cfg.chunk_samples * self.m
        src = self.audio if host_audio else self.audio_dev
        self.segs = [[a[k * cs:(k + 1) * cs] for a in src] for k in range(len(self.audio[0]) // cs)]

    @property
    def evictions(self):
        return self.batch.evictions

    def import_steady_state(self, device):
        """Every stream starts where a long-running stream is ("warm up >= 40 chunks (or pre-fill)"): LLM KV = system prompt +
        STEADY_CHUNKS chunks of (22 prompt + 9 fed) entries with the checkpoint list the agent would hold, encoder rings = the full
        window, audio history = real samples; ring starts near the physical end of both rings so that they wrap within a few steps.
        The contents are random bf16 of the scale the model produces (they do not influence the timing)."""
        cfg, eng = self.cfg, self.eng
        per_chunk = len(synth.chunk_prompt_ids(cfg, self.m, first=False)) + self.gen.max_new_tokens - 1
        n_steady = max(1, (STEADY_CHUNKS * 31) // per_chunk)  # as many whole chunks as fit the same ~961 entries (31 at m = 1 with 10 tokens)
        total = self.sys_n + n_steady * per_chunk
        g = torch.Generator(device=device)
        g.manual_seed(7)
        kv = [[torch.randn((cfg.llm_kv_heads, total, cfg.llm_head_dim), device=device, generator=g).bfloat16().cpu() for _ in range(2)]
              for _ in range(cfg.llm_layers)]
        enc = [[(0.6 * torch.randn((cfg.enc_heads, cfg.max_cache_size, cfg.enc_head_dim), device=device, generator=g)).bfloat16().cpu() for _ in range(2)]
               for _ in range(cfg.enc_layers)]
        ring_cap = 64 * ((1000 + max_prompt_len(self.sys_n, self.m) + max(self.gen.max_new_tokens, 10) + 8 + 63) // 64)  # engine_core.hip isst_create
        enc_cap = 64 * ((cfg.max_cache_size + cfg.block_size * self.m + 63) // 64)
        for i, sid in enumerate(self.sids):
            eng.import_llm_kv(sid, kv, sys_len=self.sys_n, ring_start=(ring_cap - 200 + 13 * i) % ring_cap)
            tail = torch.from_numpy(self.audio[i][-cfg.first_chunk_offset:].copy())
            eng.import_speech_cache(sid, enc, n_steps=cfg.block_size * 40, audio_tail=tail, ring_start=(enc_cap - 100 + 7 * i) % enc_cap)
            first_ck = self.sys_n + per_chunk  # cache length after the first chunk (system prompt + its turn), then one more turn each
            ckpts = [first_ck + k * per_chunk for k in range(n_steady)]
            assert ckpts[-1] == total
            self.batch.adopt_state(self.idx[i], ckpts)

    def step(self):
        self.batch.step(self.segs[self.c % self.n_chunks])
        self.c += 1


def chunk_algorithmic_bytes(cfg, n_streams, passes, kv_len):
    """HBM bytes one chunk of `n_streams` batched streams has to move -- every pass streams the Llama weights once (shared by
    the batch) and every stream's KV; the speech encoder's weights once plus its KV window per stream."""
    llm_w = 2 * (cfg.llm_layers * (cfg.llm_dim * (cfg.llm_heads + 2 * cfg.llm_kv_heads) * cfg.llm_head_dim + cfg.llm_heads * cfg.llm_head_dim * cfg.llm_dim +
                                   3 * cfg.llm_dim * cfg.llm_ffn) + cfg.vocab * cfg.llm_dim)
    kv_per_entry = 2 * cfg.llm_kv_heads * cfg.llm_head_dim * 2 * cfg.llm_layers
    enc_w = 2 * (cfg.enc_layers * (4 * cfg.enc_dim * cfg.enc_dim + 2 * cfg.enc_dim * cfg.enc_ffn) + cfg.enc_dim * cfg.conv_dim)
    enc_kv = cfg.enc_layers * 2 * (cfg.max_cache_size + cfg.block_size) * cfg.enc_dim * 2
    return passes * (llm_w + n_streams * kv_per_entry * kv_len) + enc_w + n_streams * enc_kv


def timed_steps(loop, steps):
    """`steps` chunks of every stream of `loop`, bracketed by device synchronisation; returns (seconds, per-step latencies, host seconds)."""
    torch.cuda.synchronize()
    loop.batch.reset_timers()
    lat = []
    t0 = time.perf_counter()
    for _ in range(steps):
        s0 = time.perf_counter()
        loop.step()  # returns after the last token id is on the host (isst_generate synchronises the stream)
        lat.append(time.perf_counter() - s0)
    torch.cuda.synchronize()
    return time.perf_counter() - t0, lat, loop.batch.host_seconds


def whole_step_roofline(cfg, n_streams, passes, kv_len, ms):
    algo = chunk_algorithmic_bytes(cfg, n_streams, passes, kv_len)
    achieved = algo / (ms * 1e-3) / 1e9
    return {"kernel": f"whole chunk (all kernels of one {n_streams}-stream step)", "bound": "hbm", "algorithmic_bytes_per_step": algo,
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}


def host_audio_leg(loop, steps, n_streams):
    """The same loop with every chunk handed over as host arrays (61 KB per stream, uploaded inside the step like the reference's
    `_prepare_speech`, agents/infinisst.py:222): the PCIe-inclusive rate, reported beside `value` (which has the audio resident)."""
    loop.set_host_audio(True)
    for _ in range(2):
        loop.step()
    dt, lat, _ = timed_steps(loop, steps)
    loop.set_host_audio(False)
    return {"what": "chunks handed over as host arrays, H2D inside the step (SURVEY 8(d) latency definition)", "steps": steps,
            "ms_per_step": round(1e3 * dt / steps, 3), "xrt": round(0.96 * n_streams * steps / dt, 3),
            "p50_chunk_latency_ms": round(1e3 * float(np.percentile(lat, 50)), 3)}


def run_leg(cfg, gen, weights, device, args, n_streams, steps, workload, extra_env=None):
    """One more configuration in the same process: `n_streams` concurrent streams on this GPU (shared weights, per-stream KV),
    steady state imported, a few warm-up steps, `steps` timed.  Reported next to the one-stream line."""
    saved = {k: os.environ.get(k) for k in (extra_env or {})}
    os.environ.update(extra_env or {})
    try:
        eng, _, sys_n = build_engine(cfg, n_streams, gen.max_new_tokens, device, gen.beam, weights, gen.latency_multiplier)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    loop = ChunkLoop(eng, cfg, gen, list(range(n_streams)), sys_n, host_audio=False)
    loop.import_steady_state(device)
    first = first_step_record(loop, eng, gen)  # (the first warm-up step, with stream 0's logits / candidate lists kept for the self-check below)
    for _ in range(7):
        loop.step()
    dt, lat, host_s = timed_steps(loop, steps)
    info = eng.stream_info(loop.sids[0])
    ms = 1e3 * dt / steps
    out = {"workload": workload, "streams": n_streams, "num_beams": gen.beam, "steps": steps, "ms_per_step": round(ms, 3),
           "xrt": round(0.96 * gen.latency_multiplier * n_streams * steps / dt, 2),
           "p50_chunk_latency_ms": round(1e3 * float(np.percentile(lat, 50)), 3), "p95_chunk_latency_ms": round(1e3 * float(np.percentile(lat, 95)), 3),
           "host_ms_per_step": round(1e3 * host_s / steps, 3),
           "host_ms_per_step_is": "wall time of a step spent in Python / ctypes outside isst_generate (prompt lists, argument marshalling, "
                                  "per-stream checkpoint walk and isst_kv_evict)",
           "llm_kv_entries": info["llm_cache_len"], "encoder_window": info["enc_cache_len"], "evictions_per_stream": loop.evictions // n_streams,
           "roofline": whole_step_roofline(cfg, n_streams, gen.max_new_tokens, info["llm_cache_len"], ms)}
    try:  # untimed: what this leg computed, held against another dispatch path of the library (never against the oracle: that is the tests' job)
        out["self_check"] = leg_self_check(cfg, gen, weights, device, n_streams, first)
        out["checked"] = bool(out["self_check"]["ok"])
    except Exception as e:  # report, never hide
        out["self_check"] = {"failed": f"{type(e).__name__}: {e}"}
        out["checked"] = False
    return out, loop, eng


def first_step_record(loop, eng, gen):
    """One step of `loop` that keeps what stream 0 computed: greedy -- its sampled ids and the raw logits of every pass; beam search -- its winner and the
    per-step candidate lists (isst_debug_beam_trace_*: top log-probs, ids, running beam scores)."""
    if gen.beam > 1:
        eng.beam_trace_begin(gen.beam)
        loop.step()
        trace = eng.beam_trace_end()
        return {"tokens": list(loop.batch.slots[loop.idx[0]].last_generated), "trace": trace}
    outs, logits = loop.batch.step(loop.segs[loop.c % loop.n_chunks], return_logits=True)
    loop.c += 1
    toks = list(loop.batch.slots[loop.idx[0]].last_generated)
    return {"tokens": toks, "logits": np.array(logits[0][:len(toks)], dtype=np.float32)}


# |difference| of bf16-path logits / log-probs between two dispatch paths of the library on the same inputs.  With the plain N(0, 0.02^2) init the residual stream
# reaches |x| ~ 60 and bf16 rounding alone moves a logit by 0.07 on average and 0.4-0.6 at worst (the bf16 oracle against the same math in fp32), so two
# bf16 paths with different summation orders may differ by twice that at their worst element: the check bounds the MEAN tightly and the maximum loosely.
SELF_CHECK_TOL = 1.0
SELF_CHECK_MEAN_TOL = 0.15
SELF_CHECK_MIN_SHARED = 0.5   # many-stream beams against a one-stream beam engine: share of the listed candidate ids both engines must name
SELF_CHECK_ID_TOL = 0.5       # ... and |log-prob difference| of a token both engines list (measured on the plain init: 0.13 at worst, 0.06 on average)


def leg_self_check(cfg, gen, weights, device, n_streams, first):
    """Untimed sanity check of an extra bench leg: stream 0's first step is recomputed on a SECOND engine that takes another dispatch path from the same
    imported state and the same audio, and the two are compared.
      * greedy, many streams: a one-stream engine (GEMV kernels, fused attention + o_proj instead of the batched forms), teacher-forced along the leg's ids:
        every pass's raw logits within SELF_CHECK_TOL;
      * beam search, one stream: the same search with the scorer on the HOST and attention / combine / o_proj as three launches
        (ISST_BEAM_DEVICE=0, ISST_FUSE_ATTN_OPROJ=0): winner and every candidate list must be bit-identical;
      * beam search, many streams: a one-stream beam engine: the candidates of the step behind the prefill (independent of later choices) within
        SELF_CHECK_TOL BY TOKEN ID (beam_candidates_by_id), and whether the winners agree (a near-tie may part them: reported, not required)."""
    import dataclasses
    env = {"ISST_BEAM_DEVICE": "0", "ISST_FUSE_ATTN_OPROJ": "0"} if (gen.beam > 1 and n_streams == 1) else {}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng, _, sys_n = build_engine(cfg, 1, gen.max_new_tokens, device, gen.beam, weights, gen.latency_multiplier)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        loop = ChunkLoop(eng, cfg, gen, [0], sys_n, host_audio=False)
        loop.import_steady_state(device)
        if gen.beam == 1:
            outs, logits = loop.batch.step(loop.segs[0], forced_tokens=[first["tokens"]], return_logits=True)
            toks = list(loop.batch.slots[loop.idx[0]].last_generated)
            k = len(first["tokens"])
            d = np.abs(np.array(logits[0][:k], dtype=np.float32) - first["logits"])
            diff, mean = float(d.max()), float(d.mean())
            ok = toks == first["tokens"] and diff <= SELF_CHECK_TOL and mean <= SELF_CHECK_MEAN_TOL
            return {"what": f"stream 0, first step: {k} passes recomputed by a one-stream engine, teacher-forced along the leg's ids", "passes": k,
                    "max_abs_logit_diff": round(diff, 4), "mean_abs_logit_diff": round(mean, 4), "tolerance_max": SELF_CHECK_TOL, "tolerance_mean": SELF_CHECK_MEAN_TOL,
                    "ok": bool(ok)}
        again = first_step_record(loop, eng, gen)
        t0, t1 = first["trace"], again["trace"]
        if n_streams == 1:
            same = len(t0) == len(t1) and all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) for a, b in zip(t0, t1))
            ok = same and again["tokens"] == first["tokens"]
            return {"what": "first step recomputed with the scorer on the host and attention / combine / o_proj as three launches: bit-identical candidates and winner required",
                    "scorer_steps": len(t0), "candidates_bit_identical": bool(same), "winner_equal": again["tokens"] == first["tokens"], "ok": bool(ok)}
        return beam_candidates_by_id(t0[0], t1[0], again["tokens"] == first["tokens"])
    finally:
        eng.close()


def beam_candidates_by_id(c0, c1, winner_equal):
    """Two engines' candidate lists of ONE scorer step (top log-probs [rows][k], token ids [rows][k]) compared BY TOKEN ID: the log-prob engine B gives a
    token engine A also lists must agree within the bf16 tolerance, and a token only one engine lists must not sit further above the other engine's k-th
    candidate than that tolerance allows (the other engine's value for it is at most its k-th).  With the plain random init the top candidates are near-ties
    (their ORDER legitimately differs between two summation orders), but a wrong kernel moves the values: then either the shared ids disagree or hardly any id
    is shared, and both fail the check.  (Comparing the sorted values alone passes any two rows of near-uniform log-probs: VERDICT r05.)"""
    v0, i0, v1, i1 = np.asarray(c0[0], np.float32), np.asarray(c0[1]), np.asarray(c1[0], np.float32), np.asarray(c1[1])
    rows = min(len(i0), len(i1))
    worst, diffs, shared, listed, worst_excess = 0.0, [], 0, 0, 0.0
    for r in range(rows):
        if not np.isfinite(v0[r]).any() or not np.isfinite(v1[r]).any():
            continue  # (a beam row that does not take part in this step: HF starts beams 1.. at -1e9)
        a = {int(t): float(v) for t, v in zip(i0[r], v0[r]) if np.isfinite(v)}
        b = {int(t): float(v) for t, v in zip(i1[r], v1[r]) if np.isfinite(v)}
        if not a or not b:
            continue
        ka, kb = min(a.values()), min(b.values())
        listed += len(a)
        for t, v in a.items():
            if t in b:
                shared += 1
                diffs.append(abs(v - b[t]))
            else:
                worst_excess = max(worst_excess, v - kb)
        for t, v in b.items():
            if t not in a:
                worst_excess = max(worst_excess, v - ka)
    worst = max(diffs) if diffs else float("inf")
    mean = float(np.mean(diffs)) if diffs else float("inf")
    frac = shared / max(1, listed)
    ok = bool(diffs) and frac >= SELF_CHECK_MIN_SHARED and worst <= SELF_CHECK_ID_TOL and mean <= SELF_CHECK_MEAN_TOL and worst_excess <= SELF_CHECK_ID_TOL
    return {"what": "stream 0, first step recomputed by a one-stream beam engine: the candidates of the step behind the prefill, compared BY TOKEN ID",
            "candidate_ids_shared_fraction": round(frac, 3), "shared_ids": shared, "max_abs_logprob_diff_same_id": round(worst, 4) if diffs else None,
            "mean_abs_logprob_diff_same_id": round(mean, 4) if diffs else None, "max_excess_of_unshared_id_over_other_kth": round(worst_excess, 4),
            "tolerance_max": SELF_CHECK_ID_TOL, "tolerance_mean": SELF_CHECK_MEAN_TOL, "min_shared_fraction": SELF_CHECK_MIN_SHARED,
            "winner_equal": bool(winner_equal), "ok": ok}


def run_streams64(cfg, gen, weights, device, args):
    """BASELINE.json configs[2]: 64 concurrent streams on this GPU."""
    out, loop, eng = run_leg(cfg, gen, weights, device, args, 64, args.streams64_steps,
                             "InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 64 streams on 1 MI355X (BASELINE.json configs[2])")
    out["roofline"]["note"] = ("the 1408-row prefill GEMMs and the encoder are MFMA-bound, so the HBM fraction of the whole step understates them: "
                               "see `mfma`; per-kernel figures: profiles/")
    out["host_audio"] = host_audio_leg(loop, max(4, args.streams64_steps // 2), 64)
    rows = 64 * len(synth.chunk_prompt_ids(cfg, 1, first=False))
    in_situ = None
    try:  # event pairs around the prefill gate/up launch of every layer, inside two more steps of the loop that was just timed
        eng.profile_begin(rows, rows)
        for _ in range(2):
            loop.step()
        in_situ = eng.profile_end()
    except Exception as e:
        log(f"in-situ prefill bracket failed: {type(e).__name__}: {e}")
    eng.close()
    del loop, eng
    try:
        out["mfma"] = mfma_probe(cfg, device, rows, in_situ)
    except Exception as e:  # report, never hide
        out["mfma"] = {"failed": f"{type(e).__name__}: {e}"}
    return out


def run_beam4(cfg, gen, weights, device, args):
    """The reference's production decoding mode (`--beam 4`, scripts/infer/infinisst.sh:48) -- the only setting it publishes a number for
    (RTF 0.382 = 2.6 xRT on an L40S, plots/plot.ipynb:528-531): one stream, 4 beams, steady state imported."""
    import dataclasses
    g4 = dataclasses.replace(gen, beam=4)
    out, loop, eng = run_leg(cfg, g4, weights, device, args, 1, args.beam4_steps,
                             "InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 1 stream, num_beams 4 (the reference's production decoding)")
    out["reference_published"] = {"rtf": 0.382, "xrt": 2.62, "hardware": "1x L40S (inferred)", "source": "plots/plot.ipynb:528-531"}
    eng.close()
    return out


def run_multipliers(cfg, gen, weights, device, args, b4):
    """Latency multipliers 1..4 at the reference's production decoding (num_beams 4, max_new_tokens 10 m; agents/infinisst.py:125-128,245,
    scripts/infer/infinisst.sh:42-48): one stream, steady state imported, a step = m x 960 ms of audio.  The table lines up with the only numbers
    the reference publishes, RTF at m = 1..4 (plots/plot.ipynb:528-531).  m = 1 is the `beam4` leg."""
    import dataclasses
    rows = []
    if b4 and "xrt" in b4:
        rows.append({"multiplier": 1, "chunk_ms": 960, "max_new_tokens": 10, "xrt": b4["xrt"], "ms_per_step": b4["ms_per_step"], "p50_chunk_latency_ms": b4["p50_chunk_latency_ms"],
                     "rtf": round(1.0 / b4["xrt"], 4), "checked": b4.get("checked")})
    for m in (2, 3, 4):
        g = dataclasses.replace(gen, beam=4, latency_multiplier=m, max_new_tokens=10 * m)
        out, loop, eng = run_leg(cfg, g, weights, device, args, 1, args.multiplier_steps, f"1 stream, num_beams 4, latency multiplier {m}")
        eng.close()
        del loop, eng
        rows.append({"multiplier": m, "chunk_ms": 960 * m, "max_new_tokens": 10 * m, "xrt": out["xrt"], "ms_per_step": out["ms_per_step"],
                     "p50_chunk_latency_ms": out["p50_chunk_latency_ms"], "rtf": round(1.0 / out["xrt"], 4), "llm_kv_entries": out["llm_kv_entries"],
                     "evictions_per_stream": out["evictions_per_stream"], "checked": out.get("checked"), "self_check": out.get("self_check")})
    return {"workload": "InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 1 stream on 1 MI355X, num_beams 4, chunks of m x 960 ms, max_new_tokens 10 m",
            "rows": rows, "checked": all(bool(r.get("checked")) for r in rows), "reference_published": {"what": "RTF at m = 1..4 (L40S, inferred)", "source": "plots/plot.ipynb:528-531"}}


def run_streams64_beam4(cfg, gen, weights, device, args):
    """The reference's production decoding (`--beam 4`: agents/infinisst.py:86 asserts beam > 1, scripts/infer/infinisst.sh:48) on BASELINE.json
    configs[2]: 64 concurrent streams x 4 beams = 256 decode rows per pass, steady state imported.  Also brackets the decode gate/up launch of every
    layer (gemm_wide.hip at 256 rows: the dominant weight stream of these passes) with HIP events in two more steps."""
    import dataclasses
    g4 = dataclasses.replace(gen, beam=4)
    out, loop, eng = run_leg(cfg, g4, weights, device, args, 64, args.streams64_beam4_steps,
                             "InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 64 streams x num_beams 4 on 1 MI355X "
                             "(the reference's production decoding on BASELINE.json configs[2])")
    out["roofline"]["note"] = ("whole-step HBM fraction with the KV bytes of ONE arena per stream (the beams share every key below the chunk's first "
                               "generated token); the 1408-row prefill is MFMA-bound")
    try:
        eng.profile_begin(256, 256)
        for _ in range(2):
            loop.step()
        us, n = eng.profile_end()
        wbytes = 2 * cfg.llm_ffn * cfg.llm_dim * 2
        out["decode_gate_up"] = {"kernel": "gemm_wide_kernel<16, 2, 4, EPI_SWIGLU> (256 rows, weights read once; gemm_wide.hip)", "bound": "hbm",
                                 "launch_us_in_step": round(us, 2), "launches_timed": n, "algorithmic_bytes_per_launch": wbytes,
                                 "achieved": round(wbytes / (us * 1e-6) / 1e9, 1) if us > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(wbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None,
                                 "timed_with": "HIP event pairs around that launch in every layer of two more steps (isst_profile_begin_rows; each bracket "
                                               "contains the ~2.5 us dispatch latency)"}
    except Exception as e:
        out["decode_gate_up"] = {"failed": f"{type(e).__name__}: {e}"}
    eng.close()
    return out


def run_streams64_all_ranks(cfg, gen, weights, device, args, tg, world, rank):
    """BASELINE.json configs[3]: 64 concurrent streams on EVERY GPU of the node (512 on 8), independent -- no data-path collective.  Every rank builds its
    64-stream engine, imports the steady state and times the same K steps between barriers; time = max over ranks, audio = sum over ranks.  The
    reference runs one GPU per SLURM array task (scripts/infer/infinisst.sh:5-13)."""
    # set-up (engine, steady state, warm-up) can fail on ONE rank: the ranks agree on the host before the first timed barrier, so that a local failure makes
    # every rank skip the leg instead of leaving the healthy ones inside a collective until the launcher's timeout (TimingGroup.all_ok)
    eng = loop = None
    mine = S.assign_streams(64 * world, rank, world)
    err = None
    try:
        eng, _, sys_n = build_engine(cfg, 64, args.gen_tokens, device, gen.beam, weights)
        loop = ChunkLoop(eng, cfg, gen, mine, sys_n, host_audio=False)
        loop.import_steady_state(device)
        for _ in range(8):
            loop.step()
        torch.cuda.synchronize()
    except Exception as e:  # report, never hide
        err = f"{type(e).__name__}: {e}"
    if not tg.all_ok(err is None):
        if eng is not None:
            eng.close()
        return {"failed": err or "another rank failed while setting the leg up; skipped on every rank"}
    tg.barrier()
    torch.cuda.synchronize()
    steps = args.streams64_steps
    # ... and so can the timed part (a GPU step that raises AFTER the barrier): the healthy ranks would sit in the closing barrier / reductions while the
    # failing one has left the leg.  Every rank therefore catches, and the ranks agree on the host BEFORE the next collective; a failure anywhere fails
    # the leg on every rank and main() ends the job non-zero once the line is printed (tests/test_streams_gloo.py, --dry-fail-step-rank)
    dt_local, lat, host_s = 0.0, [], 0.0
    try:
        dt_local, lat, host_s = timed_steps(loop, steps)
        torch.cuda.synchronize()
    except Exception as e:  # report, never hide
        err = f"{type(e).__name__}: {e}"
    if not tg.all_ok(err is None):
        try:
            eng.close()
        except Exception:
            pass
        return {"failed": err or "another rank failed inside the timed steps of the leg", "failed_in": "timed steps", "fatal": True}
    tg.barrier()
    dt = tg.max(dt_local)
    audio = tg.sum(0.96 * steps * len(mine))
    all_lat = tg.gather(lat)
    host_ms = tg.max(1e3 * host_s / steps)
    info = eng.stream_info(loop.sids[0])
    evictions = tg.sum(loop.evictions)
    eng.close()
    if rank != 0:
        return None
    ms = 1e3 * dt / steps
    return {"workload": f"InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 64 streams on each of {world} MI355X = {64 * world} streams, "
                        "no collective (BASELINE.json configs[3] shape)",
            "streams_per_gpu": 64, "streams_total": 64 * world, "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "steps": steps, "ms_per_step": round(ms, 3), "xrt": round(audio / dt, 2), "xrt_per_gpu": round(audio / dt / world, 2),
            "p50_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 50)), 3), "p95_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 95)), 3),
            "host_ms_per_step_max_over_ranks": round(host_ms, 3), "llm_kv_entries": info["llm_cache_len"], "encoder_window": info["enc_cache_len"],
            "evictions_all_ranks": int(evictions), "timing": "barrier + device synchronisation on both sides, max over ranks; audio summed over ranks",
            "roofline": whole_step_roofline(cfg, 64, args.gen_tokens, info["llm_cache_len"], ms)}


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16 (the 5 PF headline includes 2:1 sparsity)


def mfma_probe(cfg, device, rows, in_situ=None, iters=20):
    """MFMA utilisation of the widest dense contraction of a 64-stream step: the prefill gate/up projection (rows x 2*ffn x dim,
    SwiGLU epilogue) on the hand-written dense kernel (gemm_dense.hip).  Two live measurements with HIP events on the launch stream:
      * `in_situ` = (average us, launches) of the event pairs the engine put around that launch in every layer of real steps
        (isst_profile_begin_rows) -> `achieved`: this is the duration rocprofv3 --kernel-trace reports for the kernel inside a step;
      * `back_to_back_launch_us`: one event pair around `iters` launches of the kernel alone on random operands, weights rotating over
        copies (no launch finds its weights in the Infinity Cache), output preallocated.  A second of nothing but this kernel runs at
        lower clocks than the kernel does between the memory-bound launches of a step, so this figure is the slower of the two."""
    from infinisst_amd import engine as E
    lib = E.load_library()
    N, K = 2 * cfg.llm_ffn, cfg.llm_dim
    g = torch.Generator(device=device)
    g.manual_seed(2)
    packs = []
    for _ in range(3):
        w = torch.empty((N, K), device=device, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w))
        del w
    x = torch.randn(rows, K, device=device, generator=g).bfloat16()
    out = torch.empty((rows, N // 2), device=device, dtype=torch.bfloat16)

    def launch(i):
        rc = lib.isst_op_gemm(E._ptr(x), K, E._ptr(packs[i % 3]), None, None, 0, E._ptr(out), N // 2, rows, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr())
        if rc:
            raise RuntimeError(f"isst_op_gemm -> {rc}")
    for i in range(3):
        launch(i)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(iters):
        launch(i)
    ev1.record()
    torch.cuda.synchronize()
    b2b = 1e3 * ev0.elapsed_time(ev1) / iters
    flop = 2.0 * rows * N * K
    have = in_situ is not None and in_situ[1] > 0
    us = in_situ[0] if have else b2b
    tf = flop / (us * 1e-6) / 1e12
    return {"bound": "mfma", "kernel": E.dense_kernel_name(), "shape": f"M={rows} N={N} K={K} (prefill gate/up of one layer, 64 streams x 22 rows)",
            "launch_us": round(us, 2), "launch_us_is": "in-situ event bracket (kernel + dispatch latency)" if have else "back-to-back launches",
            "in_situ_launches": in_situ[1] if have else 0, "back_to_back_launch_us": round(b2b, 2),
            "back_to_back_tflops": round(flop / (b2b * 1e-6) / 1e12, 1), "flop_per_launch": flop, "achieved": round(tf, 1),
            "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
            "source": "HIP events in this run; rocprofv3 per-kernel durations, the MFMA-busy / fabric-read counters (dense_pmc.json) and the three-stream ablation of the same kernel: profiles/r06/"}


def gemm_roofline(cfg, device, loop=None, eng=None, iters=40, chunks=4):
    """Dominant kernel: gemm_skinny_kernel streaming the gate/up projection of one Llama layer for one token (M=1, N=2*ffn,
    K=dim, fused RMSNorm prologue, SwiGLU epilogue on self-paired tiles: the copy of the weights a one-row pass reads) -- 235 MB of weights per launch at full size, 32 % of the chunk's kernel
    time.  Two live measurements with HIP events on the launch stream:
      * `launch_us` (-> achieved): one event pair around `iters` back-to-back launches of exactly that kernel, cycling enough
        distinct weight copies that no launch finds its weights in the 256 MiB Infinity Cache -- launches pipeline, so this is
        the kernel's duration, the figure rocprofv3 --kernel-trace reports for it (profiles/);
      * `in_situ_event_bracket_us`: `chunks` more steps of the loop that was just timed, with an event pair around EVERY
        decode-pass launch of the kernel inside isst_generate (isst_profile_begin / _end, 288 launches per chunk); a bracket
        also contains that launch's dispatch latency (~2.5 us), which no kernel-duration figure includes.
    Neither runs inside the timed region (event records would perturb it)."""
    from infinisst_amd import engine as E
    N, K = 2 * cfg.llm_ffn, cfg.llm_dim
    w_bytes = N * K * 2
    copies = max(2, min(8, int(np.ceil(600e6 / w_bytes))))
    g = torch.Generator(device=device)
    g.manual_seed(1)
    packs = []
    for _ in range(copies):  # (the form a one-row pass streams: gate / up as self-paired tiles, isst_op_pack_gateup8 + epi swiglu8 -- engine_llm.hip llm_forward)
        w = torch.empty((N, K), device=device, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_gateup8(w[: N // 2], w[N // 2:]))
        del w
    x = torch.randn(1, K, device=device, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, device=device, generator=g)).bfloat16()  # post_attention_layernorm weight (fused RMSNorm)
    for p in packs:
        E.op_gemm(x, p, N, "swiglu8", norm_w=nw, norm_eps=cfg.rms_eps)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(iters):
        E.op_gemm(x, packs[i % copies], N, "swiglu8", norm_w=nw, norm_eps=cfg.rms_eps)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    del packs
    in_situ, launches = None, 0
    if loop is not None and eng is not None:
        eng.profile_begin()
        for _ in range(chunks):
            loop.step()
        in_situ, launches = eng.profile_end()
    algo_bytes = w_bytes + K * 2 + (N // 2) * 2  # weights once + activation row in + bf16 row out
    achieved = algo_bytes / (ms * 1e-3) / 1e9
    traffic, traffic_source = None, None  # HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 --pmc)
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath) and N == 28672 and K == 4096:
        with open(tpath) as f:
            tj = json.load(f)
        traffic = tj.get("hbm_bytes_per_launch")
        traffic_source = ("constant read from profiles/roofline_traffic.json = separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over this "
                          f"kernel ({tj.get('collected', 'see profiles/README.md')}); PMC counters cannot be read inside this run")
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": "gemm_skinny_kernel<1,1,EPI_SWIGLU8,nt,AMODE=2> (gate/up GEMV of a one-row pass with fused RMSNorm: self-paired gate | up tiles, 1792 one-tile workgroups)",
            "launch_us": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": algo_bytes,
            "in_situ_event_bracket_us": None if in_situ is None else round(in_situ, 2), "in_situ_launches": launches,
            "shape": f"M=1 N={N} K={K} (gate/up of one layer, one token)"}


def cpu_baseline(cfg, gen, weights_dev, sys_n, threads=0, llm_layers_run=32, n_chunks=4):
    """The CPU oracle (oracle/, PyTorch eager bf16 -- the reference has no CPU path, agents/infinisst.py:153) on a bounded sample of
    `n_chunks` consecutive steady-state chunks of one stream: the full speech encoder, then prefill + decode passes through
    `llm_layers_run` of the Llama layers (+ final norm and lm_head; the layer-stack time is scaled to all layers when fewer are run),
    with the whole-chunk eviction between the chunks.  LLM KV pre-filled to the steady-state length, encoder window saturated."""
    from oracle import generate as ogen
    from oracle import llm as ollm
    from oracle import speech_encoder as oenc
    cores = os.cpu_count() or 1
    nthreads, thread_note = (threads, "--cpu-threads") if threads > 0 else (min(cores, 64), None)
    torch.set_num_threads(nthreads)
    run_layers = min(llm_layers_run, cfg.llm_layers)
    keep = lambda k: not k.startswith("model.layers.") or int(k.split(".")[2]) < run_layers
    w = {k: v.cpu() for k, v in weights_dev.items() if keep(k)}
    log(f"cpu baseline: weights of encoder + {run_layers} llama layers on the host, {nthreads} threads")
    sub = cfg.replace(llm_layers=run_layers, eos_ids=())
    g = torch.Generator().manual_seed(3)
    prompt = synth.chunk_prompt_ids(cfg, 1, first=False)
    per_chunk = len(prompt) + gen.max_new_tokens - 1
    L = sys_n + STEADY_CHUNKS * per_chunk
    kv = [[(0.5 * torch.randn(1, cfg.llm_kv_heads, L, cfg.llm_head_dim, generator=g)).bfloat16() for _ in range(2)]
          for _ in range(run_layers)]
    sc = oenc.new_cache(cfg)
    sc.n_steps = 48 * 20
    sc.src_len = cfg.block_size
    sc.src = torch.zeros(1, cfg.first_chunk_offset + cfg.chunk_samples).bfloat16()
    for lc in sc.layers:
        lc.k = (0.5 * torch.randn(cfg.enc_heads, cfg.max_cache_size, cfg.enc_head_dim, generator=g)).bfloat16()
        lc.v = (0.5 * torch.randn(cfg.enc_heads, cfg.max_cache_size, cfg.enc_head_dim, generator=g)).bfloat16()
    rope_l = ollm.llm_rope_tables(cfg, L + 256, torch.bfloat16)
    rope_e = oenc.make_rope(cfg)
    if threads <= 0:
        # BASELINE.md's recipe is set_num_threads(os.cpu_count()); on a many-socket host that is far from the fastest setting for one stream (a probe of a
        # lone GEMV liked 128 threads on one box where the whole chunk then ran 2.8 x slower than with 64; a probe of ONE real layer, whose 436 MB of weights
        # stay in the host's caches, liked 128 threads at 4.1 against 9.9 ms where the chunk then took 19.8 s against 9.3), so the thread count is the fastest of
        # {64, 128, all logical cores} on what the baseline actually spends its time in: ONE decode pass through EIGHT consecutive real Llama layers of this
        # oracle (RMSNorm, attention over the steady-state cache, SwiGLU MLP: 3.5 GB of weights, past every cache), two repetitions each, on copies of the layers' KV
        cands = sorted({c for c in (64, 128, cores) if c <= cores}) or [cores]
        probe = {}
        probe_layers = min(8, run_layers)
        xq = torch.randn(1, 1, cfg.llm_dim, generator=g).bfloat16()
        with torch.inference_mode():
            for c_ in cands:
                torch.set_num_threads(c_)
                best = 1e9
                for _ in range(2):
                    kv_p = [[kv[l_][0].clone(), kv[l_][1].clone()] for l_ in range(probe_layers)]
                    t0 = time.perf_counter()
                    y_ = xq
                    for l_ in range(probe_layers):
                        h_ = ollm.rmsnorm(y_, w[f"model.layers.{l_}.input_layernorm.weight"], cfg.rms_eps)
                        y_ = y_ + ollm.attention(w, sub, l_, h_, kv_p, rope_l)
                        h_ = ollm.rmsnorm(y_, w[f"model.layers.{l_}.post_attention_layernorm.weight"], cfg.rms_eps)
                        y_ = y_ + ollm.mlp(w, l_, h_)
                    best = min(best, time.perf_counter() - t0)
                    if probe and best > 3 * min(probe.values()):
                        break  # (far behind a smaller thread count already: no second repetition -- all 256 logical cores of the GPU box's host take 25 s per pass)
                probe[c_] = best
        nthreads = min(probe, key=probe.get)
        thread_note = ("fastest of " + ", ".join(f"{c_} threads: {1e3 * t_:.1f} ms" for c_, t_ in probe.items()) +
                       f" for one decode pass of {probe_layers} consecutive Llama layers of the oracle; host has {cores} logical cores (BASELINE.md: set_num_threads(os.cpu_count()))")
        torch.set_num_threads(nthreads)
        log(f"cpu baseline: {thread_note}")
    audio_all = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=99)
    scale = cfg.llm_layers / run_layers
    per_chunk_s, measured = [], 0.0
    n_pass = 0
    with torch.inference_mode():
        for c in range(n_chunks):
            audio = torch.from_numpy(audio_all[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]).unsqueeze(0).bfloat16()
            t0 = time.perf_counter()
            feats, _ = oenc.encode_speech(w, sub, audio, sc, 1, rope_e)
            t_enc = time.perf_counter() - t0
            # time the layer stack and the head separately so that only the stack is scaled
            seq = list(prompt)
            t_stack = t_head = 0.0
            for step in range(gen.max_new_tokens):
                ids = torch.tensor(seq if step == 0 else seq[-1:])
                t1 = time.perf_counter()
                emb = torch.nn.functional.embedding(ids, w["model.embed_tokens.weight"])
                if step == 0:
                    emb = ollm.splice_speech(sub, ids, emb, feats[0])
                x = emb.unsqueeze(0)
                for i in range(run_layers):
                    p = f"model.layers.{i}."
                    h = ollm.rmsnorm(x, w[p + "input_layernorm.weight"], cfg.rms_eps)
                    x = x + ollm.attention(w, sub, i, h, kv, rope_l)
                    h = ollm.rmsnorm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
                    x = x + ollm.mlp(w, i, h)
                t2 = time.perf_counter()
                x = ollm.rmsnorm(x, w["model.norm.weight"], cfg.rms_eps)
                logits = torch.nn.functional.linear(x[0, -1:], w["lm_head.weight"])[0].float()
                scores = ogen.process_logits(logits, seq, [], gen.repetition_penalty, gen.no_repeat_ngram_size,
                                             gen.no_repeat_ngram_size, ())
                seq.append(int(torch.argmax(scores)))
                t3 = time.perf_counter()
                t_stack += t2 - t1
                t_head += t3 - t2
                n_pass += 1
            # the last sampled token is never fed: the cache grew by prompt + (passes - 1); then evict one chunk (agents/infinisst.py:354-361)
            for layer in kv:
                for j in (0, 1):
                    layer[j] = torch.cat([layer[j][:, :, :sys_n], layer[j][:, :, -(STEADY_CHUNKS * per_chunk):]], dim=2)
            per_chunk_s.append(t_enc + t_stack * scale + t_head)
            measured += t_enc + t_stack + t_head
            log(f"cpu baseline: chunk {c}: encoder {t_enc:.2f} s, layer stack {t_stack:.2f} s, head {t_head:.2f} s")
    dt = float(np.mean(per_chunk_s))
    return {"value": round(0.96 / dt, 4), "unit": "xRT (audio-s/wall-s), 1 stream", "cores": nthreads, "host_cores": cores, "kind": "port",
            "threads_reason": thread_note,
            "sample": f"{n_chunks} consecutive steady-state chunks ({0.96 * n_chunks:.2f} s audio) of one stream: full speech encoder + {len(prompt)}-token "
                      f"prefill + {gen.max_new_tokens - 1} decode passes per chunk through {run_layers} of {cfg.llm_layers} Llama layers (+ norm, lm_head, "
                      f"processors), eviction between chunks; measured {measured:.1f} s"
                      + (f", layer-stack time scaled x{scale:g}" if scale != 1 else "") +
                      f" -> {dt:.2f} s per chunk (p50 {float(np.percentile(per_chunk_s, 50)):.2f} s); LLM KV {L} entries, encoder window "
                      f"{cfg.max_cache_size}; torch {torch.__version__} bf16 eager, {nthreads} threads of {cores} cores",
            "chunk_seconds": round(dt, 3), "p50_chunk_seconds": round(float(np.percentile(per_chunk_s, 50)), 3), "chunks": n_chunks}


class DryEngine:
    """`--dry-run` stand-in for the library (no GPU, no compute): a call sleeps, caches grow and shrink as the real ones would, so the
    product's StreamBatch (prompts, checkpoint walk, evictions) and the launcher / barrier / reduction path run unchanged."""

    def __init__(self, rank, gen_tokens):
        self.rank, self.gen_tokens, self.lens, self.last_call_seconds = rank, gen_tokens, [], 0.0

    def open_stream(self):
        self.lens.append(0)
        return len(self.lens) - 1

    def generate(self, gen, sids, pcm, prompts, prevs, system_prompt_size=0, forced_tokens=None, return_logits=False):
        t0 = time.perf_counter()
        time.sleep(0.002 * (1 + self.rank))
        for sid, pr in zip(sids, prompts):
            self.lens[sid] += len(pr) + self.gen_tokens - 1
        self.last_call_seconds = time.perf_counter() - t0
        return [[7] * self.gen_tokens for _ in sids], None

    def stream_cache_lens(self, sids):
        return [self.lens[s] for s in sids]

    def kv_evict(self, sid, new_size, keep):
        self.lens[sid] = new_size + keep


def dry_run(args, world, rank, cores=None):
    """Launcher self-test (no GPU, no compute): the ranks meet over gloo, deal the global stream ids, step their streams through the
    product's StreamBatch over a sleeping stand-in engine, and go through exactly the barrier / max-over-ranks / gather path of a real run."""
    if world > 1:
        dist.init_process_group("gloo")
    tg = S.TimingGroup(None)
    mine = S.assign_streams(args.streams * world, rank, world)
    cfg = toy_config()
    gen = GenConfig(latency_multiplier=1, max_new_tokens=args.gen_tokens, max_llm_cache_size=200, always_cache_system_prompt=True)
    sys_n = len(synth.system_prompt_ids(cfg))
    batch = S.StreamBatch(DryEngine(rank, args.gen_tokens), gen, sys_n, lambda first, m: synth.chunk_prompt_ids(cfg, m, first=first))
    idx = [batch.open() for _ in mine]
    seg = np.zeros(cfg.chunk_samples, dtype=np.float32)
    tg.barrier()
    t0 = time.perf_counter()
    lat = []
    for _ in range(args.steps):
        s0 = time.perf_counter()
        batch.step([seg] * len(idx))
        lat.append(time.perf_counter() - s0)
    elapsed_local = time.perf_counter() - t0
    tg.barrier()
    elapsed = tg.max(elapsed_local)
    audio_s = tg.sum(0.96 * args.steps * len(mine))
    all_lat = tg.gather(lat)
    evictions = tg.sum(batch.evictions)
    # the configs[3] leg of a real N > 1 run (run_streams64_all_ranks): 64 streams on EVERY rank, the same barrier / max / sum path
    s64 = None
    if world > 1 and not args.no_streams64:
        mine64 = S.assign_streams(64 * world, rank, world)
        b64 = S.StreamBatch(DryEngine(rank, args.gen_tokens), gen, sys_n, lambda first, m: synth.chunk_prompt_ids(cfg, m, first=first))
        idx64 = [b64.open() for _ in mine64]
    if world > 1 and not args.no_streams64 and not tg.all_ok(rank != args.dry_fail_rank):  # (run_streams64_all_ranks: a failed set-up is agreed on before the timed barrier)
        s64 = {"failed": "a rank failed while setting the leg up; skipped on every rank", "ranks_seen": dist.get_world_size()}
    elif world > 1 and not args.no_streams64:
        tg.barrier()
        t1 = time.perf_counter()
        step_err = None
        try:
            for k in range(4):
                if rank == args.dry_fail_step_rank and k == 1:
                    raise RuntimeError("simulated failure of a GPU step inside the timed leg (--dry-fail-step-rank)")
                b64.step([seg] * len(idx64))
        except Exception as e:  # (run_streams64_all_ranks: caught, then agreed on before the next collective)
            step_err = f"{type(e).__name__}: {e}"
        dt64_local = time.perf_counter() - t1
        leg_ok = tg.all_ok(step_err is None)
    if s64 is None and world > 1 and not args.no_streams64 and not leg_ok:
        s64 = {"failed": step_err or "another rank failed inside the timed steps of the leg", "failed_in": "timed steps", "fatal": True, "ranks_seen": dist.get_world_size()}
    elif s64 is None and world > 1 and not args.no_streams64:
        tg.barrier()
        dt64 = tg.max(dt64_local)
        audio64 = tg.sum(0.96 * 4 * len(mine64))
        core_sets = tg.gather([float(c) for c in (cores or [])])
        s64 = {"streams_per_gpu": 64, "streams_total": int(round(tg.sum(len(mine64)))), "ranks_seen": dist.get_world_size(), "steps": 4,
               "ms_per_step": round(1e3 * dt64 / 4, 3), "xrt": round(audio64 / dt64, 2), "cores_pinned_all_ranks": len(core_sets),
               "cores_pinned_distinct": len(set(core_sets))}
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN of the launcher (no compute, not a measurement)", "dry_run": True, "value": round(audio_s / elapsed, 3),
                          "streams64": s64, "host_cores_of_rank0": list(cores or []),
                          "unit": "audio-seconds per wall-second", "n_gpus": world, "ranks_seen": dist.get_world_size() if world > 1 else 1, "timing_collectives": tg.describe(),
                          "steps": args.steps, "warmup": args.warmup, "streams_of_rank0": mine, "latencies_gathered": len(all_lat),
                          "evictions_all_ranks": int(evictions), "host_ms_per_step": round(1e3 * batch.host_seconds / max(1, batch.ticks), 3),
                          "ms_per_step": round(1e3 * elapsed / max(1, args.steps), 3), "higher_is_better": True, "scaling": "weak"}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if isinstance(s64, dict) and s64.get("fatal"):
        sys.exit(3)  # (every rank: the launcher, and bench.py as its parent, then exit non-zero)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    # host cores of this rank: local to its GPU's NUMA node, disjoint from the other ranks' -- set BEFORE anything touches the GPU
    cores = S.pin_rank_to_local_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))) if world > 1 else sorted(os.sched_getaffinity(0))
    if args.dry_run:
        return dry_run(args, world, rank, cores)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:  # (a one-rank launch still forms the group: the RCCL path is then exercised on a one-GPU box too)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
    tg = S.TimingGroup(device if world > 1 else None)

    cfg = (toy_config() if args.toy else full_config()).replace(eos_ids=())  # fixed G: EOS never stops a chunk early
    gen = GenConfig(latency_multiplier=1, max_new_tokens=args.gen_tokens, no_repeat_ngram_size=5, no_repeat_ngram_lookback=100,
                    repetition_penalty=1.2, max_llm_cache_size=1000, always_cache_system_prompt=True, beam=args.beam)
    eng, weights, sys_n = build_engine(cfg, args.streams, args.gen_tokens, device, args.beam)
    if args.attn_target_wgs:
        from infinisst_amd.engine import load_library
        load_library().isst_op_set_attn_tuning(args.attn_target_wgs)
    if args.gemm_tuning:
        from infinisst_amd.engine import load_library
        load_library().isst_op_set_gemm_tuning(args.gemm_tuning, 0)
    mine = S.assign_streams(args.streams * world, rank, world)  # global stream ids of this rank: stream_id mod n_gpu
    loop = ChunkLoop(eng, cfg, gen, mine, sys_n, host_audio=args.host_audio)
    if not args.cold_start:
        loop.import_steady_state(device)
        log(f"steady state imported: KV {eng.stream_info(loop.sids[0])['llm_cache_len']} entries, encoder window {eng.stream_info(loop.sids[0])['enc_cache_len']}")

    def sync_all():
        torch.cuda.synchronize()
        tg.barrier()
        torch.cuda.synchronize()

    # untimed spin-up before the W warm-up steps (like the steady-state import above it is preparation, not part of the contract's W + K steps): the first
    # seconds of GPU work on a fresh box run with clocks / page tables still settling -- one default run in five showed a 35 ms p95 against 32.0 otherwise
    for _ in range(args.spinup):
        loop.step()
    for i in range(args.warmup):
        loop.step()
        if i == 0:
            log("first chunk done")
    sync_all()
    ev0 = loop.evictions
    kv_min = eng.stream_info(loop.sids[0])["llm_cache_len"]
    log(f"warm-up done ({args.warmup} chunks, KV {kv_min} entries)")
    elapsed_local, lat, _ = timed_steps(loop, args.steps)  # EXACTLY K steps between device synchronisations
    sync_all()
    elapsed = tg.max(elapsed_local)
    audio_s = tg.sum(0.96 * args.steps * len(mine))
    all_lat = tg.gather(lat)
    info = eng.stream_info(loop.sids[0])
    timed_evictions = loop.evictions - ev0
    log(f"timed region done: {args.steps} steps in {elapsed:.3f} s")

    roof = None
    base = None
    s64 = None
    b4 = None
    s64b4 = None
    mult = None
    host_leg = None
    host_ms = 1e3 * loop.batch.host_seconds / max(1, loop.batch.ticks)
    if world > 1 and args.streams == 1 and args.beam == 1 and not args.toy and not args.no_streams64:
        # N > 1: configs[3] -- the 64-stream leg on EVERY rank (the N = 1 headline above stays what BENCH measures, so N = 1 SCALE agrees with it)
        try:
            s64 = run_streams64_all_ranks(cfg, gen, weights, device, args, tg, world, rank)
            if rank == 0 and s64 is not None and "xrt" in s64:
                log(f"64-streams-per-GPU leg done on {world} ranks: {s64['xrt']} xRT aggregate, {s64['ms_per_step']} ms per step")
            elif rank == 0 and s64 is not None:
                log(f"64-streams-per-GPU leg skipped on every rank: {s64.get('failed')}")
        except Exception as e:  # report, never hide (set-up failures are agreed on and skipped by all ranks inside; what lands here failed in the timed part)
            s64 = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0:
        if not args.host_audio and args.host_audio_steps > 0:
            host_leg = host_audio_leg(loop, args.host_audio_steps, len(mine))
            log(f"host-audio leg done: {host_leg['ms_per_step']} ms per step")
        if not args.no_roofline:
            in_situ = args.streams == 1 and args.beam == 1
            roof = gemm_roofline(cfg, device, loop if in_situ else None, eng if in_situ else None)
            roof["whole_step"] = whole_step_roofline(cfg, args.streams, args.gen_tokens, info["llm_cache_len"], 1e3 * elapsed / args.steps)
            log(f"roofline probe done: {roof['achieved']} GB/s, in-situ bracket {roof['in_situ_event_bracket_us']} us")
        legs = world == 1 and args.streams == 1 and args.beam == 1 and not args.toy
        if legs:
            # Every leg below builds its own engine.  The headline engine is released first, and the one-stream legs run BEFORE the 64-stream ones: the same
            # one-stream beam-4 configuration measured 33.13 ms as the first engine built after the headline one, 33.76 ms when its memory was carved out of
            # what a 64-stream engine had just freed, and 34.77 ms next to a second live engine of its own size (profiles/r05/leg_order_probe.txt; rounds 3-4
            # reported that leg ~0.9 ms above the same configuration run in a process of its own for this reason) -- placement of 17 GB of weights in
            # recycled device memory is the driver's business, a leg should not be timed on its leftovers
            eng.close()
        if legs and not args.no_beam4:
            try:
                b4 = run_beam4(cfg, gen, weights, device, args)
                log(f"beam-4 leg done: {b4['xrt']} xRT, {b4['ms_per_step']} ms per step")
            except Exception as e:  # report, never hide
                b4 = {"failed": f"{type(e).__name__}: {e}"}
        if legs and not args.no_multipliers:
            try:
                mult = run_multipliers(cfg, gen, weights, device, args, b4)
                log("latency-multiplier table done: " + ", ".join(f"m={r['multiplier']}: {r['xrt']} xRT" for r in mult["rows"]))
            except Exception as e:  # report, never hide
                mult = {"failed": f"{type(e).__name__}: {e}"}
        if legs and not args.no_streams64:
            try:
                s64 = run_streams64(cfg, gen, weights, device, args)
                log(f"64-stream leg done: {s64['xrt']} xRT, {s64['ms_per_step']} ms per step")
            except Exception as e:  # report, never hide
                s64 = {"failed": f"{type(e).__name__}: {e}"}
        if legs and not args.no_streams64_beam4:
            try:
                s64b4 = run_streams64_beam4(cfg, gen, weights, device, args)
                log(f"64 streams x beam 4 leg done: {s64b4['xrt']} xRT, {s64b4['ms_per_step']} ms per step")
            except Exception as e:  # report, never hide
                s64b4 = {"failed": f"{type(e).__name__}: {e}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                base = cpu_baseline(cfg, gen, weights, sys_n, args.cpu_threads, args.cpu_layers, args.cpu_chunks)
                log(f"cpu baseline done: {base['value']} xRT")
            except Exception as e:  # report, never hide
                base = {"value": None, "unit": "xRT (audio-s/wall-s), 1 stream", "cores": os.cpu_count(), "kind": "port",
                        "sample": f"failed: {type(e).__name__}: {e}"}
    # (every rank knows: run_streams64_all_ranks returns the agreed failure on all of them)
    fatal = not tg.all_ok(not (isinstance(s64, dict) and s64.get("fatal")))
    tg.barrier()
    if rank == 0:
        value = audio_s / elapsed
        line = {
            "metric": "xRT (audio-s/wall-s), InfiniSST en-de 8B, whole job",
            "value": round(value, 3),
            "unit": "audio-seconds per wall-second",
            "n_gpus": world,
            "ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
            "timing_collectives": tg.describe(),
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup_steps_untimed": args.spinup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic 16 kHz audio clip(0.1*N(0,1)), random-init weights N(0,0.02^2), synthetic prompt ids",
            "audio": "host arrays, uploaded inside each step (PCIe-inclusive)" if args.host_audio else "resident in HBM before the timed region",
            "config": {"workload": "toy dims (plumbing only)" if args.toy else
                       ("InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 1 stream per MI355X (BASELINE.json configs[1])"
                        if args.streams == 1 else
                        f"InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, {args.streams} streams per MI355X "
                        "(BASELINE.json configs[2] shape)"),
                       "streams_per_gpu": args.streams, "streams_total": args.streams * world, "chunk_ms": 960, "prompt_tokens": 22,
                       "forward_passes_per_chunk": args.gen_tokens,
                       "latency_definition": ("ms_per_step / value / p50: one isst_generate call per chunk, audio ALREADY RESIDENT in HBM when the step starts "
                                              "(the bench contract: inputs resident before the timed region) -> last token id on the host; SURVEY 8(d)'s form -- H2D of "
                                              "the chunk inside the step -- is the `host_audio` object beside it (never `value`)") if not args.host_audio else
                                             "SURVEY 8(d): H2D of the chunk -> last token id on the host (--host-audio)",
                       "llm_kv_entries": info["llm_cache_len"], "llm_kv_entries_at_timed_start": kv_min, "encoder_window": info["enc_cache_len"],
                       "steady_state": "cold start (--cold-start)" if args.cold_start else
                                       "imported before the first step (KV, checkpoints, encoder window, audio history)",
                       "greedy": args.beam == 1, "num_beams": args.beam,
                       "parallelism": f"stream-parallel replicas x{world}, no collective",
                       "evictions_per_stream": timed_evictions // max(1, len(mine)), "evictions_counted_over": "the timed steps"},
            "p50_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 50)), 3),
            "p95_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 95)), 3),
            "xrt_per_gpu": round(value / world, 3),
            "host_ms_per_step": round(host_ms, 3),
            "host_audio": host_leg,
            "roofline": roof,
            "cpu_baseline": base,
            "streams64": s64,
            "beam4": b4,
            "streams64_beam4": s64b4,
            "multipliers": mult,
            "host_cores_of_rank0": f"{len(cores)} cores ({cores[0]}..{cores[-1]})" if cores else None,
        }
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()
    if fatal:
        sys.exit(3)  # a rank failed inside the timed steps of a multi-rank leg: the line says so, and the job does not end as a success


if __name__ == "__main__":
    main()
