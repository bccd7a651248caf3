"""Parity at BASELINE.json's FULL size (wav2vec2-large + Llama-3.1-8B shapes, random-init weights): the HIP path vs
the CPU oracle on two chunks (first chunk with the system prompt, then a steady 22-token chunk), and the size-independent
property batched == single.  The oracle needs ~10-20 s per chunk on the GPU box's host cores."""
import numpy as np
import pytest
import torch

from infinisst_amd import synth
from infinisst_amd.config import GenConfig, full_config
from infinisst_amd.engine import Engine
from oracle import generate as ogen
from oracle import llm as ollm
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def full():
    torch.set_num_threads(min(64, torch.get_num_threads()))
    cfg = full_config()
    dev = torch.device("cuda")
    w = synth.random_weights_device(cfg, dev)
    sys_n = len(synth.system_prompt_ids(cfg))
    eng = Engine(cfg, max_streams=3, max_prompt_len=sys_n + 32, max_new_tokens=10, max_llm_cache_size=1000, max_system_prompt=sys_n,
                 debug_taps=True)
    eng.load_weights(w)
    return cfg, w, eng, sys_n


def test_full_size_two_chunks_match_oracle(full):
    """At full width the residual stream of the random-init model reaches |x| ~ 60, so bf16 rounding alone moves the
    logits by ~0.07 on average (bf16 oracle vs the same math in fp32).  Criterion: the HIP path's error against the fp32
    oracle must stay within 1.5x the bf16 oracle's own error (mean and max), i.e. it is indistinguishable from the
    reference's bf16 arithmetic; encoder features (O(1) values) keep the absolute tolerance."""
    cfg, w_dev, eng, sys_n = full
    w = {k: v.cpu() for k, v in w_dev.items()}
    w32 = {k: v.float() for k, v in w.items()}  # same bf16-representable values, fp32 arithmetic
    gen = GenConfig(max_new_tokens=2)
    sid = eng.open_stream()
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=42)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    kv32, sc32 = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_e = oenc.make_rope(cfg)
    rope_l, rope_l32 = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), ollm.llm_rope_tables(cfg, 2048, torch.float32)
    stats = []
    for c in range(2):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        with torch.inference_mode():
            ref32 = ogen.generate(w32, cfg, gen, prompt, x.unsqueeze(0).bfloat16().float(), kv32, sc32, rope_l32, rope_e, [])
            forced = ref32.sequences[len(prompt):]
            ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, [], forced_tokens=forced)
        outs, logits = eng.generate(gen, [sid], [seg], [prompt], [[]], system_prompt_size=sys_n if c == 0 else 0,
                                    forced_tokens=[forced], return_logits=True)
        assert outs[0] == forced
        feat = eng.debug_tap("speech").view(-1, cfg.llm_dim).float()
        d = (feat - ref.speech_features.float()).abs()
        assert float(d.max()) <= 0.06 + 0.02 * float(ref.speech_features.float().abs().max()), f"chunk {c}: speech features off by {float(d.max())}"
        for s in range(len(forced)):
            truth = ref32.step_logits[s].float().numpy()
            e_ref = np.abs(ref.step_logits[s].float().numpy() - truth)
            e_hip = np.abs(logits[0, s] - truth)
            print(f"chunk {c} step {s}: bf16-oracle err mean {e_ref.mean():.4f} max {e_ref.max():.4f} | HIP err mean {e_hip.mean():.4f} max {e_hip.max():.4f}"
                  f" | argmax fp32/bf16/HIP {int(np.argmax(truth))}/{int(np.argmax(ref.step_logits[s].float().numpy()))}/{int(np.argmax(logits[0, s]))}", flush=True)
            stats.append((e_ref.mean(), e_ref.max(), e_hip.mean(), e_hip.max()))
            assert e_hip.mean() <= 1.5 * e_ref.mean() + 0.005
            assert e_hip.max() <= 1.5 * e_ref.max() + 0.05
            top2 = np.sort(truth)[-2:]
            if top2[1] - top2[0] > 2.5 * e_ref.max():  # decisive for bf16 arithmetic
                assert int(np.argmax(logits[0, s])) == int(np.argmax(truth))
        assert eng.stream_info(sid)["llm_cache_len"] == ollm.kv_len(kv)
    eng.close_stream(sid)


def test_full_size_batched_equals_single(full):
    cfg, _, eng, sys_n = full
    gen = GenConfig(max_new_tokens=4)
    a, b = eng.open_stream(), eng.open_stream()
    audio = [synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=i) for i in (1, 2)]
    for c in range(2):
        segs = [x[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for x in audio]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        o1, l1 = eng.generate(gen, [a], [segs[0]], [prompt], [[]], return_logits=True)
        eng.reset_stream(b) if c == 0 else None
        # stream b replays stream a's audio inside a 2-stream batch next to another stream? -> compare a (single) with b (batched with itself shifted)
        o2, l2 = eng.generate(gen, [b], [segs[0]], [prompt], [[]], forced_tokens=[o1[0]], return_logits=True)
        n = min(len(o1[0]), len(o2[0]))
        assert np.array_equal(l1[0][:n], l2[0][:n]), "the same stream replayed must be bit-identical (deterministic kernels)"
    eng.close_stream(a)
    eng.close_stream(b)


@pytest.mark.parametrize("n", [4, 8, 14, 16])
def test_full_size_multi_stream_batch_matches_single_stream(full, n):
    """(14 / 16 streams: the decode passes run on gemm_mid + split-K slabs with the reducing RMSNorm instead of the skinny kernel.  8 streams: prefill 176 rows -- q/k/v as K slices + slab reduce too.)  4 streams in one call take different kernels from one stream (prefill 88 rows: split-K slabs on the dense kernel, q/k/v on
    two 64-row blocks; decode 4 rows: fused norm with the LDS overlay).  Same inputs on every stream, teacher-forced with the
    single-stream tokens: the logits of every stream must agree with the single-stream run to bf16 noise (the two paths sum the
    same products in different fp32 orders; at this width that moves logits by ~0.07 on average, see the test above) and must be
    bit-identical across the four streams."""
    cfg, w_dev, eng1, sys_n = full
    gen = GenConfig(max_new_tokens=4)
    eng4 = Engine(cfg, max_streams=n, max_prompt_len=sys_n + 32, max_new_tokens=10, max_llm_cache_size=1000, max_system_prompt=sys_n)
    eng4.load_weights(w_dev)
    s1 = eng1.open_stream()
    sids = [eng4.open_stream() for _ in range(n)]
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=7)
    for c in range(2):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        o1, l1 = eng1.generate(gen, [s1], [seg], [prompt], [[]], system_prompt_size=sys_n if c == 0 else 0, return_logits=True)
        o4, l4 = eng4.generate(gen, sids, [seg] * n, [prompt] * n, [[]] * n, system_prompt_size=sys_n if c == 0 else 0,
                               forced_tokens=[o1[0]] * n, return_logits=True)
        k = len(o1[0])
        for i in range(n):
            assert o4[i] == o1[0]
            assert np.array_equal(l4[i][:k], l4[0][:k]), "identical streams of one batch must be bit-identical"
        d = np.abs(l4[0][:k] - l1[0][:k])
        print(f"chunk {c}: {n}-stream vs 1-stream logits mean |d| {d.mean():.4f} max {d.max():.4f}")
        assert d.mean() <= 0.15 and d.max() <= 1.0
    eng1.close_stream(s1)
    eng4.close()
