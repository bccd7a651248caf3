#!/usr/bin/env python3
"""Benchmark of the InfiniSST hot path on MI355X: xRT (audio-seconds per wall-second), whole job.

    python bench.py --gpus 1 --steps 64 --warmup 40
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one 960 ms chunk of every stream of this rank through the whole per-chunk path (the body of the
reference's policy(), agents/infinisst.py:287-361): H2D of the chunk, conv extractor, streaming encoder, shrink +
projector, Llama-3.1-8B prefill of the 22-token chunk prompt and greedy decode, then the chunk-wise KV eviction.
Workload = BASELINE.json configs[1]: InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 1 stream
per GPU, synthetic 16 kHz audio, random-init weights; EOS is disabled so that every chunk
runs the worst case max_new_tokens = 10 forward passes (1 prefill + 9 decode steps).  The warm-up fills the LLM KV
cache to its steady state (max_llm_cache_size = 1000 + pinned system prompt, eviction active) and saturates the
encoder window (576 frames).  N ranks = N independent replicas (stream-parallel, no collective); the barrier and the
max-over-ranks reduction are the only cross-rank traffic.

Rank 0 prints ONE JSON line.  Extra objects: `roofline` for the dominant kernel (the packed-weight skinny GEMM
streaming Llama weights, HBM-bound) measured live with HIP events on the launch stream, and `cpu_baseline` (the
CPU oracle timed on this box's host cores on one steady-state chunk, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--streams", type=int, default=1, help="concurrent streams per GPU (configs[1] = 1)")
    ap.add_argument("--gen-tokens", type=int, default=10, help="max_new_tokens per chunk (production: 10 x multiplier)")
    ap.add_argument("--beam", type=int, default=1, help="num_beams (1 = greedy, the north-star mode; 4 = the reference's production setting)")
    ap.add_argument("--toy", action="store_true", help="toy dimensions (plumbing check, not a valid benchmark)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores, capped at 64)")
    ap.add_argument("--attn-target-wgs", type=int, default=0, help="profiling aid: isst_op_set_attn_tuning (0 = library default)")
    ap.add_argument("--cpu-layers", type=int, default=32, help="Llama layers actually run by the CPU baseline (time is scaled to all layers)")
    return ap.parse_args()


def build_engine(cfg, n_streams, gen_tokens, device, beams=1):
    from infinisst_amd.engine import Engine
    sys_n = len(synth.system_prompt_ids(cfg))
    eng = Engine(cfg, max_streams=n_streams, max_multiplier=1, max_prompt_len=sys_n + 32, max_new_tokens=max(gen_tokens, 10),
                 max_llm_cache_size=1000, max_system_prompt=sys_n, max_beams=beams)
    log("engine created")
    w = synth.random_weights_device(cfg, device)
    torch.cuda.synchronize()
    log("random weights drawn on device")
    eng.load_weights(w)
    log("weights packed into the library")
    return eng, w, sys_n


class ChunkLoop:
    """The per-chunk control logic of policy() for the streams of one rank (generate + whole-chunk eviction)."""

    def __init__(self, eng, cfg, gen, n_streams, sys_n, rank):
        self.eng, self.cfg, self.gen, self.sys_n = eng, cfg, gen, sys_n
        self.sids = [eng.open_stream() for _ in range(n_streams)]
        n_chunks = 64
        self.audio = [synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=rank * 1000 + i) for i in range(n_streams)]
        self.n_chunks = n_chunks
        self.ckpts = [[] for _ in range(n_streams)]
        self.targets = [[] for _ in range(n_streams)]
        self.first = True
        self.c = 0
        self.evictions = 0

    def step(self):
        cfg, gen = self.cfg, self.gen
        k = self.c % self.n_chunks
        segs = [a[k * cfg.chunk_samples:(k + 1) * cfg.chunk_samples] for a in self.audio]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=self.first)
        prev = [t[-gen.no_repeat_ngram_lookback:] for t in self.targets]
        outs, _ = self.eng.generate(gen, self.sids, segs, [prompt] * len(self.sids), prev,
                                    system_prompt_size=self.sys_n if self.first else 0)
        for i, sid in enumerate(self.sids):
            self.targets[i].extend(outs[i][:-1])
            self.targets[i] = self.targets[i][-256:]
            cur = self.eng.stream_info(sid)["llm_cache_len"]
            ck = self.ckpts[i]
            ck.append(cur)
            if cur > gen.max_llm_cache_size:  # reference agents/infinisst.py:340-361
                new_size = 0
                for j, c in enumerate(ck):
                    new_size = cur - c
                    if new_size <= gen.max_llm_cache_size:
                        trimmed = c - self.sys_n
                        self.ckpts[i] = [x - trimmed for x in ck[j + 1:]]
                        break
                self.eng.kv_evict(sid, new_size, self.sys_n)
                self.evictions += 1
        self.first = False
        self.c += 1


def gemm_roofline(cfg, device, loop=None, eng=None, iters=40, chunks=4):
    """Dominant kernel: gemm_skinny_kernel streaming the gate/up projection of one Llama layer for one token (M=1, N=2*ffn,
    K=dim, fused RMSNorm prologue, SwiGLU epilogue) -- 235 MB of weights per launch at full size, 32 % of the chunk's kernel
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
    for _ in range(copies):
        w = torch.empty((N, K), device=device, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w))
        del w
    x = torch.randn(1, K, device=device, generator=g).bfloat16()
    nw = (1 + 0.1 * torch.randn(K, device=device, generator=g)).bfloat16()  # post_attention_layernorm weight (fused RMSNorm)
    for p in packs:
        E.op_gemm(x, p, N, "swiglu", norm_w=nw, norm_eps=cfg.rms_eps)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for i in range(iters):
        E.op_gemm(x, packs[i % copies], N, "swiglu", norm_w=nw, norm_eps=cfg.rms_eps)
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
    traffic = None  # HBM bytes per launch from the PMC passes committed under profiles/ (collected with rocprofv3 --pmc)
    tpath = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    if os.path.exists(tpath) and N == 28672 and K == 4096:
        with open(tpath) as f:
            traffic = json.load(f).get("hbm_bytes_per_launch")
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "kernel": "gemm_skinny_kernel<1,2,EPI_SWIGLU,nt,AMODE=2> (gate/up GEMV with fused RMSNorm)",
            "launch_us": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": algo_bytes,
            "in_situ_event_bracket_us": None if in_situ is None else round(in_situ, 2), "in_situ_launches": launches,
            "shape": f"M=1 N={N} K={K} (gate/up of one layer, one token)"}


def cpu_baseline(cfg, gen, weights_dev, sys_n, threads=0, llm_layers_run=4):
    """The CPU oracle (oracle/, PyTorch eager bf16 -- the reference has no CPU path, agents/infinisst.py:153) on a
    bounded sample of ONE steady-state chunk: the full speech encoder, then prefill + decode passes through
    `llm_layers_run` of the Llama layers (+ final norm and lm_head); the layer-stack time is scaled to all layers.
    LLM KV pre-filled to the steady-state length, encoder window saturated."""
    from oracle import generate as ogen
    from oracle import llm as ollm
    from oracle import speech_encoder as oenc
    cores = os.cpu_count() or 1
    nthreads = threads if threads > 0 else min(cores, 64)
    torch.set_num_threads(nthreads)
    run_layers = min(llm_layers_run, cfg.llm_layers)
    keep = lambda k: not k.startswith("model.layers.") or int(k.split(".")[2]) < run_layers
    w = {k: v.cpu() for k, v in weights_dev.items() if keep(k)}
    log(f"cpu baseline: weights of encoder + {run_layers} llama layers on the host, {nthreads} threads")
    sub = cfg.replace(llm_layers=run_layers, eos_ids=())
    g = torch.Generator().manual_seed(3)
    L = sys_n + gen.max_llm_cache_size - 40
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
    audio = torch.from_numpy(synth.synthetic_audio(cfg.chunk_samples, stream_id=99)).unsqueeze(0).bfloat16()
    prompt = synth.chunk_prompt_ids(cfg, 1, first=False)
    with torch.inference_mode():
        t0 = time.perf_counter()
        feats, _ = oenc.encode_speech(w, sub, audio, sc, 1, rope_e)
        t_enc = time.perf_counter() - t0
        log(f"cpu baseline: encoder {t_enc:.2f} s")
        # time the layer stack and the head separately so that only the stack is scaled
        seq = list(prompt)
        t_stack = t_head = 0.0
        n_pass = 0
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
            if step == 0:
                log(f"cpu baseline: prefill over {run_layers} layers {t2 - t1:.2f} s")
    scale = cfg.llm_layers / run_layers
    dt = t_enc + t_stack * scale + t_head
    measured = t_enc + t_stack + t_head
    return {"value": round(0.96 / dt, 4), "unit": "xRT (audio-s/wall-s), 1 stream", "cores": nthreads, "kind": "port",
            "sample": f"1 steady-state chunk (0.96 s audio): full speech encoder + {len(prompt)}-token prefill + {n_pass - 1} decode "
                      f"passes through {run_layers} of {cfg.llm_layers} Llama layers (+ norm, lm_head, processors); measured "
                      f"{measured:.1f} s, layer-stack time scaled x{scale:g} -> {dt:.1f} s per chunk; LLM KV {L} entries, encoder window "
                      f"{cfg.max_cache_size}; torch {torch.__version__} bf16 eager, {nthreads} threads of {cores} cores",
            "chunk_seconds": round(dt, 3)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)

    cfg = (toy_config() if args.toy else full_config()).replace(eos_ids=())  # fixed G: EOS never stops a chunk early
    gen = GenConfig(latency_multiplier=1, max_new_tokens=args.gen_tokens, no_repeat_ngram_size=5, no_repeat_ngram_lookback=100,
                    repetition_penalty=1.2, max_llm_cache_size=1000, always_cache_system_prompt=True, beam=args.beam)
    eng, weights, sys_n = build_engine(cfg, args.streams, args.gen_tokens, device, args.beam)
    if args.attn_target_wgs:
        from infinisst_amd.engine import load_library
        load_library().isst_op_set_attn_tuning(args.attn_target_wgs)
    loop = ChunkLoop(eng, cfg, gen, args.streams, sys_n, rank)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loop.step()
        if i == 0:
            log("first chunk done")
    sync_all()
    log(f"warm-up done ({args.warmup} chunks, KV {eng.stream_info(loop.sids[0])['llm_cache_len']} entries)")
    lat = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s0 = time.perf_counter()
        loop.step()  # returns after the last token id is on the host (generate synchronises the stream)
        lat.append(time.perf_counter() - s0)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    sync_all()
    elapsed = S.max_over_ranks(elapsed_local, device if world > 1 else None)
    audio_s = S.sum_over_ranks(0.96 * args.steps * args.streams, device if world > 1 else None)
    all_lat = S.gather_floats(lat, device if world > 1 else None)
    info = eng.stream_info(loop.sids[0])
    log(f"timed region done: {args.steps} steps in {elapsed:.3f} s")

    roof = None
    base = None
    if rank == 0:
        if not args.no_roofline:
            in_situ = args.streams == 1 and args.beam == 1
            roof = gemm_roofline(cfg, device, loop if in_situ else None, eng if in_situ else None)
            log(f"roofline probe done: {roof['achieved']} GB/s, in-situ bracket {roof['in_situ_event_bracket_us']} us")
        if world == 1 and not args.no_cpu_baseline:
            try:
                base = cpu_baseline(cfg, gen, weights, sys_n, args.cpu_threads, args.cpu_layers)
                log(f"cpu baseline done: {base['value']} xRT")
            except Exception as e:  # report, never hide
                base = {"value": None, "unit": "xRT (audio-s/wall-s), 1 stream", "cores": os.cpu_count(), "kind": "port",
                        "sample": f"failed: {type(e).__name__}: {e}"}
    if world > 1:
        dist.barrier()
    if rank == 0:
        value = audio_s / elapsed
        line = {
            "metric": "xRT (audio-s/wall-s), InfiniSST en-de 8B, whole job",
            "value": round(value, 3),
            "unit": "audio-seconds per wall-second",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic 16 kHz audio clip(0.1*N(0,1)), random-init weights N(0,0.02^2), synthetic prompt ids",
            "config": {"workload": "toy dims (plumbing only)" if args.toy else
                       ("InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, 1 stream per MI355X (BASELINE.json configs[1])"
                        if args.streams == 1 else
                        f"InfiniSST en-de, wav2vec2-large + Llama-3.1-8B bf16, 960 ms chunks, {args.streams} streams per MI355X "
                        "(BASELINE.json configs[2] shape)"),
                       "streams_per_gpu": args.streams, "chunk_ms": 960, "prompt_tokens": 22, "forward_passes_per_chunk": args.gen_tokens,
                       "llm_kv_entries": info["llm_cache_len"], "encoder_window": info["enc_cache_len"], "greedy": args.beam == 1, "num_beams": args.beam,
                       "parallelism": f"stream-parallel replicas x{world}, no collective", "evictions_per_stream": loop.evictions // max(1, args.streams)},
            "p50_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 50)), 3),
            "p95_chunk_latency_ms": round(1e3 * float(np.percentile(all_lat, 95)), 3),
            "xrt_per_gpu": round(value / world, 3),
            "roofline": roof,
            "cpu_baseline": base,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
