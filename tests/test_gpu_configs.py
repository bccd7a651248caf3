"""The other BASELINE.json configurations as parity / property tests on the GPU (toy dimensions, same structure):
latency multiplier 2, many concurrent streams with shared weights, and a long unbounded stream with rolling eviction."""
import numpy as np
import pytest
import torch

from infinisst_amd import synth
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine
from oracle import generate as ogen
from oracle import llm as ollm
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [2, 3, 4])
def test_latency_multiplier_matches_oracle(m):
    """m = 2..4: 48 m-frame blocks, 12 m speech tokens per chunk, max_new_tokens = 10 m (reference agents/infinisst.py:125-128,245;
    model/speech_encoder.py:143-145; the quality-latency curve of infer/infinisst.sh:42-47)."""
    cfg = toy_config()
    gen = GenConfig(latency_multiplier=m, max_new_tokens=12, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=31)
    eng = Engine(cfg, max_streams=1, max_multiplier=m, max_prompt_len=160, max_new_tokens=20, max_llm_cache_size=400,
                 max_system_prompt=64)
    eng.load_weights(w)
    sid = eng.open_stream()
    n = cfg.chunk_samples * m
    audio = synth.synthetic_audio(n * 3, stream_id=7)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), oenc.make_rope(cfg)
    worst = 0.0
    for c in range(3):
        seg = audio[c * n:(c + 1) * n]
        prompt = synth.chunk_prompt_ids(cfg, m, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, [])
        forced = ref.sequences[len(prompt):]
        outs, logits = eng.generate(gen, [sid], [seg], [prompt], [[]], forced_tokens=[forced], return_logits=True)
        assert outs[0] == forced
        for s, rl in enumerate(ref.step_logits):
            worst = max(worst, float(np.abs(logits[0, s] - rl.float().numpy()).max()))
        info = eng.stream_info(sid)
        assert info["enc_n_steps"] == sc.n_steps == 48 * m * (c + 1)
        assert info["llm_cache_len"] == ollm.kv_len(kv)
    print(f"m={m}: worst |logit diff| {worst:.4f}")
    assert worst <= 0.15


def test_64_concurrent_streams_shared_weights():
    """BASELINE configs[2] in miniature: 64 streams stepped together, per-stream KV; every stream must produce the
    logits it produces alone (stream 0, 17 and 63 are checked against single-stream engines)."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=4, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=32)
    n_s = 64
    eng = Engine(cfg, max_streams=n_s + 3, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=150, max_system_prompt=64)
    eng.load_weights(w)
    sids = [eng.open_stream() for _ in range(n_s)]
    solo = {i: eng.open_stream() for i in (0, 17, 63)}
    audio = [synth.synthetic_audio(cfg.chunk_samples * 3, stream_id=100 + i) for i in range(n_s)]
    for c in range(3):
        segs = [a[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for a in audio]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        single = {}
        for i, sid in solo.items():
            o, l = eng.generate(gen, [sid], [segs[i]], [prompt], [[]], return_logits=True)
            single[i] = (o[0], l[0])
        forced = [single[i][0] if i in single else None for i in range(n_s)]
        outs, logits = eng.generate(gen, sids, segs, [prompt] * n_s, [[] for _ in range(n_s)], forced_tokens=forced, return_logits=True)
        for i in single:
            n = len(single[i][0])
            d = float(np.abs(logits[i][:n] - single[i][1][:n]).max())
            assert d <= 0.07, f"chunk {c} stream {i}: {d}"
        lens = {eng.stream_info(s)["llm_cache_len"] for s in sids}
        assert all(l > 0 for l in lens)


def test_long_stream_is_bounded_and_o1():
    """BASELINE configs[4] in miniature: an unbounded stream (150 chunks) with rolling eviction: cache lengths stay
    bounded, the encoder window saturates, outputs stay finite, and a late chunk equals the oracle run on the same
    (evicted) state -- the ring wraps several times on the way."""
    cfg = toy_config().replace(block_size=16, max_cache_size=40)
    gen = GenConfig(max_new_tokens=4, max_llm_cache_size=90)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=33)
    eng = Engine(cfg, max_streams=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=90, max_system_prompt=64)
    eng.load_weights(w)
    sid = eng.open_stream()
    sys_n = len(synth.system_prompt_ids(cfg))
    n_chunks = 150
    audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=9)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), oenc.make_rope(cfg)
    ckpts = []
    from oracle import agent as oag
    worst = 0.0
    for c in range(n_chunks):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, [], keep_logits=(c % 10 == 9))
        forced = ref.sequences[len(prompt):]
        outs, logits = eng.generate(gen, [sid], [seg], [prompt], [[]], system_prompt_size=sys_n if c == 0 else 0,
                                    forced_tokens=[forced], return_logits=(c % 10 == 9))
        if c % 10 == 9:
            for s, rl in enumerate(ref.step_logits):
                assert np.isfinite(logits[0, s]).all()
                worst = max(worst, float(np.abs(logits[0, s] - rl.float().numpy()).max()))
        cur = ollm.kv_len(kv)
        info = eng.stream_info(sid)
        assert info["llm_cache_len"] == cur
        ckpts.append(cur)
        ev = oag.evict(ckpts, cur, gen.max_llm_cache_size, True, sys_n)
        if ev is not None:
            ckpts, new_size = ev
            for layer in kv:
                for j in (0, 1):
                    layer[j] = torch.cat([layer[j][:, :, :sys_n], layer[j][:, :, -new_size:]], dim=2)
            eng.kv_evict(sid, new_size, sys_n)
        assert eng.stream_info(sid)["llm_cache_len"] <= gen.max_llm_cache_size + sys_n
        assert info["enc_cache_len"] <= cfg.max_cache_size + cfg.block_size
    print(f"long stream: worst |logit diff| over sampled chunks {worst:.4f}")
    assert worst <= 0.15


def test_update_multiplier_mid_stream_matches_oracle():
    """`update_multiplier` between chunks (reference agents/infinisst.py:125-128, model/speech_encoder.py:143-145): one stream runs
    m = 1, 2, 2, 1, 3, 1 with its encoder cache and LLM KV alive -- block size, speech-token count and max_new_tokens change from call
    to call while the rings carry over.  Teacher-forced logits against the oracle on the same schedule."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=35)
    eng = Engine(cfg, max_streams=1, max_multiplier=3, max_prompt_len=160, max_new_tokens=30, max_llm_cache_size=600, max_system_prompt=64)
    eng.load_weights(w)
    sid = eng.open_stream()
    schedule = [1, 2, 2, 1, 3, 1]
    audio = synth.synthetic_audio(cfg.chunk_samples * sum(schedule), stream_id=17)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), oenc.make_rope(cfg)
    pos, worst, frames = 0, 0.0, 0
    for c, m in enumerate(schedule):
        gen = GenConfig(latency_multiplier=m, max_new_tokens=min(10 * m, 8), max_llm_cache_size=600)
        n = cfg.chunk_samples * m
        seg = audio[pos:pos + n]
        pos += n
        prompt = synth.chunk_prompt_ids(cfg, m, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, [])
        forced = ref.sequences[len(prompt):]
        outs, logits = eng.generate(gen, [sid], [seg], [prompt], [[]], forced_tokens=[forced], return_logits=True)
        assert outs[0] == forced
        for s, rl in enumerate(ref.step_logits):
            worst = max(worst, float(np.abs(logits[0, s] - rl.float().numpy()).max()))
        frames += 48 * m
        info = eng.stream_info(sid)
        assert info["enc_n_steps"] == sc.n_steps == frames and info["llm_cache_len"] == ollm.kv_len(kv)
        assert info["enc_cache_len"] == sc.layers[0].k.shape[1]
    print(f"multiplier schedule {schedule}: worst |logit diff| {worst:.4f}")
    assert worst <= 0.15


def test_agent_multiplier_2_with_a_short_last_segment_matches_oracle_agent():
    """at m = 2 the last segment of an utterance is padded to whole 960 ms chunks, so a tail <= 960 ms yields 12 speech
    features for a prompt with 24 patch slots.  The reference's slices just splice the shorter run (model/llm.py:101-110): the agent must
    emit the final translation, with the same cache length and token ids as the oracle agent."""
    from infinisst_amd.agent import InfiniSST, WriteAction, default_args
    from oracle import agent as oag
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=36, recipe="peaked")  # decisive greedy steps: synth.apply_recipe
    m = 2
    args = default_args(latency_multiplier=m, max_latency_multiplier=2, max_new_tokens=6, max_llm_cache_size=400)
    eng = Engine(cfg, max_streams=1, max_multiplier=2, max_prompt_len=160, max_new_tokens=16, max_llm_cache_size=400, max_system_prompt=64)
    eng.load_weights(w)
    agent = InfiniSST(args, engine=eng, model_cfg=cfg)
    gen = GenConfig(latency_multiplier=m, max_new_tokens=6, max_llm_cache_size=400)
    oa = oag.OracleAgent(w, cfg, gen, lambda first: synth.chunk_prompt_ids(cfg, m, first), system_prompt_size=agent.system_prompt_size)
    seg = cfg.chunk_samples * m
    wav = synth.synthetic_audio(seg * 2 + 5000, stream_id=23)   # two full 1920 ms segments + a 312 ms tail -> one padded chunk, 12 features
    st, so = agent.states, oa.build_states()  # (the agent's constructor has opened its stream, as SimulEval's base class does)
    st.source_sample_rate = so.source_sample_rate = 16000
    margins, last = [], None
    for pos in range(0, wav.shape[0], seg):
        piece = wav[pos:pos + seg].tolist()
        st.source.extend(piece)
        so.source.extend(piece)
        st.source_finished = so.source_finished = pos + seg >= wav.shape[0]
        last = agent.policy(st)
        oa.policy(so)
        for sc in oa.last_output.step_scores[:-1]:
            top2 = torch.topk(sc, 2).values
            margins.append(float(top2[0] - top2[1]))
    assert isinstance(last, WriteAction) and last.finished
    assert oa.last_output.speech_features.shape[0] == 12, "the oracle saw a 12-feature tail"
    got, ref = list(st.target_ids), list(so.target_ids)
    print("agent ids:", got, "oracle ids:", ref)
    first_tie = next((i for i, mg in enumerate(margins) if mg <= 0.3), len(margins))
    k = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    assert first_tie >= min(8, len(ref)), f"the peaked recipe must give decisive steps (first near-tie at {first_tie} of {len(ref)})"
    assert k >= min(first_tie, len(ref))
    if got == ref:
        assert eng.stream_info(st.stream_id)["llm_cache_len"] == ollm.kv_len(so.past_key_values)


@pytest.mark.parametrize("m,n,mode,chunks", [(1, 176, "bf16", 15), (1, 176, "fp32", 3), (2, 90, "bf16", 8)])
def test_encoder_48_row_blocks_at_many_streams_match_the_streams_alone(m, n, mode, chunks):
    """enc_attn.hip's many-stream form (48-row query blocks, two workgroups per CU, a second key tile in flight, the packed bf16 rotary table) runs from
    1024 (stream, head, block) workgroups on: 176 toy streams (2 heads) at m = 1 -- ONE block per (head, stream): the workgroup appends the chunk's keys and
    V^T groups at its start and phase 3 reads them back from the ring -- and 90 streams at m = 2 -- two blocks per (head, stream): the chunk's keys and values
    come from the qkv rows, fragments of V^T are mixed, block 0 appends at its end.  "fp32" tables are read as handed over (no packed table).
    Streams 0, n // 2 and n - 1 are held to the SAME streams stepped alone (16-row blocks, few-row GEMMs) chunk by chunk: first chunk, growing window, and at
    m = 1 the saturated window (from chunk 13) with the ring wrapping (640 slots).  Batched and alone differ by the GEMM kernels' summation orders only."""
    cfg = toy_config().replace(enc_rope_mode=mode)
    gen = GenConfig(latency_multiplier=m, max_new_tokens=1, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.08, norm_jitter=0.1, seed=35)
    eng = Engine(cfg, max_streams=n + 1, max_multiplier=m, max_prompt_len=160, max_new_tokens=4, max_llm_cache_size=150, max_system_prompt=64, debug_taps=True)
    eng.load_weights(w)
    sids = [eng.open_stream() for _ in range(n)]
    solo = eng.open_stream()
    ns = cfg.chunk_samples * m
    picks = (0, n // 2, n - 1)
    audio = [synth.synthetic_audio(ns * chunks, stream_id=500 + i) for i in range(n)]
    ref = {}
    for i in picks:  # the picked streams alone: the speech features of every chunk
        eng.reset_stream(solo)
        ref[i] = [eng.encode_speech(solo, audio[i][c * ns:(c + 1) * ns], multiplier=m).float().cpu() for c in range(chunks)]
    S = ref[picks[0]][0].shape[0]
    worst = 0.0
    for c in range(chunks):
        p = synth.chunk_prompt_ids(cfg, m, first=(c == 0))
        eng.generate(gen, sids, [a[c * ns:(c + 1) * ns] for a in audio], [p] * n, [[] for _ in range(n)])
        got = eng.debug_tap("speech").view(n, S, -1).float().cpu()
        for i in picks:
            err = (got[i] - ref[i][c]).abs()
            worst = max(worst, float(err.max()))
            bad = err > 0.06 + 0.02 * ref[i][c].abs()
            assert not bad.any(), f"m={m} {mode} chunk {c} stream {i} of {n}: max |d| {float(err.max()):.4f} (max |ref| {float(ref[i][c].abs().max()):.3f}), {int(bad.sum())} out of tolerance"
        for sid in sids:  # keep the LLM cache inside its ring (what it holds does not reach the speech features)
            if eng.stream_info(sid)["llm_cache_len"] > 100:
                eng.kv_evict(sid, 40, 0)
    info = eng.stream_info(sids[0])
    assert info["enc_n_steps"] == 48 * m * chunks and info["enc_cache_len"] == min(576 + 48 * m, 48 * m * chunks)  # (the window is trimmed at the next chunk's start)
    print(f"m={m} {mode}: {n} streams x {chunks} chunks, worst |d| vs the streams alone {worst:.4f}")
