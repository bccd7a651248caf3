"""Beam search (the reference's production decoding mode, scripts/infer/infinisst.sh:48 beam=4) on the GPU.

Two kinds of evidence:
  * bookkeeping is exact: after every chunk, ALL beam arenas of the stream hold exactly the KV of [history + prompt +
    fed tokens of the winning hypothesis] -- compared bit for bit with a greedy stream that is teacher-forced along the
    same winner path (this exercises the prompt replication, the per-step tail reorder, hypothesis tail copies and the
    finalize write-back);
  * decisions follow the oracle's restatement of patch_hf.py:43-302,687-967: step by step while both sides are in the
    same state, the chosen (token, parent) lists agree, or the first disagreement is a near-tie of candidate scores.
"""
import numpy as np
import pytest
import torch

from infinisst_amd import synth
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine
from oracle import beam as obeam
from oracle import llm as ollm
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu


def kv_of(eng, sid, n, beam=0):
    ks, vs = zip(*[eng.read_kv(sid, p, layer=1, kv_head=1, beam=beam) for p in range(n)])
    return torch.stack(ks), torch.stack(vs)


@pytest.mark.parametrize("shared", [False, True])
@pytest.mark.parametrize("eos", [True, False])
def test_beam_kv_bookkeeping_is_exact(eos, shared, monkeypatch):
    """shared=False (ISST_BEAM_SHARED=0): every beam attends over its own arena with exactly the greedy kernel's decomposition, so the
    arenas must equal the teacher-forced greedy replay BIT FOR BIT -- this pins the bookkeeping.  shared=True (default): the prefix pass
    is shared between the beams and the per-beam keys sit in extra workgroups; the softmax blocks are cut differently, so the attention
    output (and through it the next layers' K/V) may differ by a bf16 rounding: same positions, same bookkeeping, values within 2 ulps."""
    monkeypatch.setenv("ISST_BEAM_SHARED", "1" if shared else "0")
    cfg = toy_config() if eos else toy_config().replace(eos_ids=())
    if eos:  # make EOS likely enough that hypotheses get closed early: many ids count as EOS
        cfg = cfg.replace(eos_ids=(1001, 1008, 1009, 7, 8, 9))
    B = 4
    gen_b = GenConfig(max_new_tokens=7, beam=B, max_llm_cache_size=400)
    gen_g = GenConfig(max_new_tokens=7, beam=1, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=41)
    eng = Engine(cfg, max_streams=2, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    a, g = eng.open_stream(), eng.open_stream()
    audio = synth.synthetic_audio(cfg.chunk_samples * 5, stream_id=3)
    sys_n = len(synth.system_prompt_ids(cfg))
    prev = []
    for c in range(5):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        pin = sys_n if c == 0 else 0
        outs, _ = eng.generate(gen_b, [a], [seg], [prompt], [prev[-100:]], system_prompt_size=pin)
        win = outs[0]
        assert 1 <= len(win) <= gen_b.max_new_tokens
        # replay the winner path on a greedy stream; its last token is never fed on either side
        outs_g, _ = eng.generate(gen_g, [g], [seg], [prompt], [prev[-100:]], system_prompt_size=pin, forced_tokens=[win])
        assert outs_g[0] == win
        na, ng = eng.stream_info(a)["llm_cache_len"], eng.stream_info(g)["llm_cache_len"]
        assert na == ng, f"chunk {c}: cache {na} vs {ng} (winner {win})"
        kg, vg = kv_of(eng, g, ng)
        for b in range(B):
            kb, vb = kv_of(eng, a, na, beam=b)
            if shared:
                # within two bf16 roundings of the row's largest element (upstream rounding flips act on the scale of the vector, not of each element)
                same = lambda x, y: float((x.float() - y.float()).abs().max()) <= 2.0 ** -6 * float(y.float().abs().max())
            else:
                same = torch.equal
            bad_k = [p for p in range(na) if not same(kb[p], kg[p])]
            bad_v = [p for p in range(na) if not same(vb[p], vg[p])]
            assert not bad_k and not bad_v, (f"chunk {c}: arena of beam {b} differs from the replayed winner path at K positions {bad_k[:12]} "
                                             f"V positions {bad_v[:12]} (cache {na}, prompt {len(prompt)}, winner {win}; "
                                             f"max|dK| {float((kb.float() - kg.float()).abs().max()):.4f})")
        prev.extend(win[:-1])
    print("winner lengths ok; final cache", eng.stream_info(a)["llm_cache_len"])


LP_TOL = 0.15  # processed log-probs: the logit tolerance of tests/test_gpu_engine.py (measured worst at toy width: 0.035)
GAP = 0.1      # an oracle candidate whose neighbours are further away than this is "decisive": the device must rank the same token there


def _oracle_beam(w, cfg, gen, B, prompt, audio, rope_l, rope_e):
    x = torch.cat([torch.zeros(cfg.first_chunk_offset), torch.from_numpy(audio)])
    return obeam.beam_generate(w, cfg, gen, B, prompt, x.unsqueeze(0).bfloat16(), ollm.new_kv(cfg), oenc.new_cache(cfg), rope_l, rope_e, [])


def test_beam_teacher_forced_candidates_match_oracle():
    """Teacher-forced beam search (patch_hf.py:833-913): the engine continues along the ORACLE's (token, parent) choices, so both sides
    are in the same state at every step, and the device's per-beam processed log-probs -- top 2B values and token ids, what the scorer
    consumes -- are compared step by step: values within LP_TOL, identical candidate ids wherever the oracle's neighbouring candidates
    are further apart than GAP, beam scores within LP_TOL per step taken."""
    cfg = toy_config().replace(eos_ids=())  # no EOS: every step has B live beams on both sides
    B = 4
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=47)
    eng = Engine(cfg, max_streams=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 1024, torch.bfloat16), oenc.make_rope(cfg)
    n_keep = 2 * B
    worst_v = worst_s = 0.0
    checked_ids = decisive_ids = 0
    for trial in range(4):
        sid = eng.open_stream()
        audio = synth.synthetic_audio(cfg.chunk_samples, stream_id=90 + trial)
        prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
        ref = _oracle_beam(w, cfg, gen, B, prompt, audio, rope_l, rope_e)
        assert len(ref.steps) == gen.max_new_tokens
        eng.beam_trace_begin(B, [st.next_tokens for st in ref.steps], [st.next_parents for st in ref.steps])
        outs, _ = eng.generate(gen, [sid], [audio], [prompt], [[]])
        trace = eng.beam_trace_end()
        assert len(trace) == len(ref.steps)
        for step, (st, (val, idx, sc)) in enumerate(zip(ref.steps, trace)):
            rows = 1 if step == 0 else B
            assert val.shape == (rows, n_keep)
            for b in range(rows):
                lp = (st.scores[b] - st.beam_scores_in[b]).float()       # the oracle's processed log-probs of beam b
                top = torch.topk(lp, n_keep + 1)
                ov, oi = top.values.numpy(), top.indices.numpy()
                dv = np.abs(val[b] - ov[:n_keep])
                worst_v = max(worst_v, float(dv.max()))
                assert dv.max() <= LP_TOL, f"trial {trial} step {step} beam {b}: top log-probs differ by {dv.max():.3f}"
                for j in range(n_keep):
                    checked_ids += 1
                    lo = ov[j - 1] - ov[j] if j > 0 else np.inf
                    hi = ov[j] - ov[j + 1]
                    if min(lo, hi) > GAP:
                        decisive_ids += 1
                        assert idx[b, j] == oi[j], f"trial {trial} step {step} beam {b} rank {j}: token {idx[b, j]} vs oracle {oi[j]}"
                    else:  # a near-tie may swap ranks, but the token must come from the oracle's neighbourhood of that rank
                        near = {int(t) for t, v_ in zip(top.indices.tolist(), top.values.tolist()) if abs(v_ - ov[j]) <= GAP}
                        near |= {int(t) for t in torch.nonzero((lp - float(ov[j])).abs() <= GAP).flatten().tolist()}
                        assert int(idx[b, j]) in near, f"trial {trial} step {step} beam {b} rank {j}: token {idx[b, j]} is not within GAP of the oracle's rank"
                ds = abs(float(sc[b]) - float(st.beam_scores_in[b]))
                worst_s = max(worst_s, ds)
                assert ds <= LP_TOL * max(1, step), f"trial {trial} step {step} beam {b}: beam score {sc[b]} vs {st.beam_scores_in[b]}"
        # forced along the oracle's path, the open beams ARE the oracle's: the winner may only differ at a near-tie of final scores
        finals = sorted((s_ / (len(ref.steps) ** 1.0) for s_ in ref.steps[-1].next_scores), reverse=True)
        if finals[0] - finals[1] > GAP:
            assert outs[0] == ref.sequences[len(prompt):], f"trial {trial}: {outs[0]} vs {ref.sequences[len(prompt):]}"
        eng.close_stream(sid)
    print(f"teacher-forced beam: worst |d log-prob| {worst_v:.4f}, worst |d beam score| {worst_s:.4f}, {decisive_ids}/{checked_ids} candidate ids decisive")
    assert decisive_ids > 0


def test_beam_decisions_follow_oracle():
    """Free-running beam search against the oracle's restatement of patch_hf.py:43-302,687-967: every trial must produce the
    oracle's sequence, unless its FIRST divergent step (located with the candidate trace: the merged candidate order of the device
    against the oracle's) is a near-tie of the oracle's own candidate scores at that step."""
    cfg = toy_config()
    B = 4
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=42)
    eng = Engine(cfg, max_streams=8, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 1024, torch.bfloat16), oenc.make_rope(cfg)
    V = cfg.vocab
    same = near_tie = 0
    for trial in range(8):
        sid = eng.open_stream()
        audio = synth.synthetic_audio(cfg.chunk_samples, stream_id=50 + trial)
        prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
        ref = _oracle_beam(w, cfg, gen, B, prompt, audio, rope_l, rope_e)
        eng.beam_trace_begin(B)
        outs, _ = eng.generate(gen, [sid], [audio], [prompt], [[]])
        trace = eng.beam_trace_end()
        ref_new = ref.sequences[len(prompt):]
        if outs[0] == ref_new:
            same += 1
        else:
            first = None
            for step, st in enumerate(ref.steps):
                if step >= len(trace):
                    first = step
                    break
                val, idx, sc = trace[step]
                merged = sorted(((float(val[b, j] + sc[b]), b * V + int(idx[b, j])) for b in range(val.shape[0]) for j in range(val.shape[1])),
                                key=lambda t: (-t[0], t[1]))[:len(st.cand_tokens)]
                dev_order = [f for _, f in merged]
                ref_order = [b * V + t for b, t in zip(st.cand_beams, st.cand_tokens)]
                if dev_order != ref_order:
                    first = step
                    break
            assert first is not None, f"trial {trial}: sequences differ ({outs[0]} vs {ref_new}) although every step's candidate order agrees"
            cs = ref.steps[min(first, len(ref.steps) - 1)].cand_scores
            gap = min(abs(cs[j] - cs[j + 1]) for j in range(len(cs) - 1))
            assert gap < 0.08, (f"trial {trial}: sequences differ ({outs[0]} vs {ref_new}); first divergent step {first} has no near-tie "
                                f"(min candidate gap {gap:.3f})")
            near_tie += 1
        eng.close_stream(sid)
    print(f"beam vs oracle: {same} identical, {near_tie} diverged at a near-tie of that step's candidates")
    assert same + near_tie == 8 and same >= 3


def test_beam_many_streams_run_the_mid_row_machinery_and_stay_order_independent():
    """5 streams x 4 beams = 20 rows per decode pass: the 13..64-row machinery (gemm_mid, K slices with the in-launch reduction, norm while staging)
    under beam search.  Properties: the same call twice and the streams in reverse order give every stream the same tokens (bit-transparent
    batching); against each stream searched ALONE (4 rows: the skinny kernels, another summation order) most streams must agree -- a near-tie may
    legitimately flip a beam decision, so 3 of 5 is the bar, as in test_beam_decisions_follow_oracle."""
    cfg = toy_config()
    B, n = 4, 5
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=44)
    eng = Engine(cfg, max_streams=n, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    audio = [synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=70 + i) for i in range(n)]

    def run(order):
        sids = [eng.open_stream() for _ in range(n)]
        toks = [[] for _ in range(n)]
        for c in range(2):
            p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            outs, _ = eng.generate(gen, [sids[i] for i in order], [audio[i][c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for i in order],
                                   [p] * len(order), [[]] * len(order))
            for pos, i in enumerate(order):
                toks[i].append(outs[pos])
        for sid in sids:
            eng.close_stream(sid)
        return toks

    a = run(list(range(n)))
    assert run(list(range(n))) == a, "beam search over 20 rows is not deterministic"
    assert run(list(range(n))[::-1]) == a, "a stream's beam search depends on its position in the batch"
    same = 0
    for i in range(n):
        sid = eng.open_stream()
        alone = []
        for c in range(2):
            p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            outs, _ = eng.generate(gen, [sid], [audio[i][c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]], [p], [[]])
            alone.append(outs[0])
        eng.close_stream(sid)
        same += int(alone == a[i])
    print(f"beam 4, 5 streams batched vs alone: {same} of {n} identical")
    assert same >= 3


def test_beam_finished_stream_is_frozen_while_batch_mates_continue():
    """the scorer must skip a stream whose search is done (patch_hf.py:83-92) while other streams of the call keep decoding --
    otherwise the finished stream goes on closing hypotheses and its winner / KV tail depend on its batch mates.  Stream A (finishes
    early: many ids count as EOS) is run next to a copy of itself and next to the longest-running stream found; the row count of the
    passes is 2B in both calls (finished streams' rows ride along), so A's tokens, cache length and KV must be IDENTICAL."""
    cfg = toy_config().replace(eos_ids=(1001, 1008, 1009, 7, 8, 9))
    B = 4
    gen = GenConfig(max_new_tokens=8, beam=B, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=48)
    w["lm_head.weight"][list(cfg.eos_ids)] *= 3.0  # EOS logits three times as spread: hypotheses close early (the oracle then takes 4..8 steps)
    eng = Engine(cfg, max_streams=2, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
    steps = {}
    for seed in range(200, 208):  # how many scorer steps does each candidate stream take on its own?
        sid = eng.open_stream()
        eng.beam_trace_begin(B)
        eng.generate(gen, [sid], [synth.synthetic_audio(cfg.chunk_samples, stream_id=seed)], [prompt], [[]])
        steps[seed] = len(eng.beam_trace_end())
        eng.close_stream(sid)
    a_seed, b_seed = min(steps, key=steps.get), max(steps, key=steps.get)
    print(f"scorer steps per candidate: {steps}; A = {a_seed} ({steps[a_seed]} steps), B = {b_seed} ({steps[b_seed]} steps)")
    assert steps[a_seed] + 2 <= steps[b_seed], "no early-finishing stream among the candidates"
    audio_a, audio_b = synth.synthetic_audio(cfg.chunk_samples, stream_id=a_seed), synth.synthetic_audio(cfg.chunk_samples, stream_id=b_seed)

    def run(mate_audio):
        s0, s1 = eng.open_stream(), eng.open_stream()
        outs, _ = eng.generate(gen, [s0, s1], [audio_a, mate_audio], [prompt, prompt], [[], []])
        n = eng.stream_info(s0)["llm_cache_len"]
        kv = [kv_of(eng, s0, n, beam=b) for b in range(B)]
        eng.close_stream(s0)
        eng.close_stream(s1)
        return outs, n, kv

    (o_self, n_self, kv_self), (o_long, n_long, kv_long) = run(audio_a), run(audio_b)
    assert o_self[0] == o_self[1], "two copies of one stream in a batch must agree"
    assert o_long[0] == o_self[0] and n_long == n_self, f"stream A changed with its batch mate: {o_long[0]} (cache {n_long}) vs {o_self[0]} (cache {n_self})"
    for b in range(B):
        assert torch.equal(kv_long[b][0], kv_self[b][0]) and torch.equal(kv_long[b][1], kv_self[b][1]), f"arena {b} of stream A differs"


def test_beam_rotated_key_arena_is_bit_identical_to_rotate_on_read(monkeypatch):
    """The beams' arenas are read through the once-per-chunk rotated-key arena like a greedy stream's (pre-pass over all arenas, rotated
    keys travelling with every position copy).  Against ISST_ROT_KEYS=0 (every key rotated on every read): the same outputs and the
    same KV in every beam's arena, across chunks, a pinned system prompt and evictions."""
    from oracle import agent as oag
    cfg = toy_config()
    B = 3
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=44)
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=150, always_cache_system_prompt=True)
    audio = synth.synthetic_audio(cfg.chunk_samples * 7, stream_id=12)
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_ROT_KEYS", flag)
        eng = Engine(cfg, max_streams=2, max_multiplier=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=150, max_system_prompt=64, max_beams=B)
        eng.load_weights(w)
        sid = eng.open_stream()
        outs, kvs, ckpts, prev = [], [], [], []
        for c in range(7):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            ids, _ = eng.generate(gen, [sid], [seg], [prompt], [prev[-100:]], system_prompt_size=sys_n if c == 0 else 0)
            outs.append(ids[0])
            prev.extend(ids[0][:-1])
            cur = eng.stream_info(sid)["llm_cache_len"]
            kvs.append([kv_of(eng, sid, cur, beam=b) for b in range(B)])
            ckpts.append(cur)
            ev = oag.evict(ckpts, cur, gen.max_llm_cache_size, True, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                eng.kv_evict(sid, new_size, sys_n)
        eng.close()
        return outs, kvs

    (oa, ka), (ob, kb) = run("1"), run("0")
    assert oa == ob
    for c, (xa, xb) in enumerate(zip(ka, kb)):
        for b in range(B):
            assert torch.equal(xa[b][0], xb[b][0]) and torch.equal(xa[b][1], xb[b][1]), f"chunk {c} beam {b}: KV differs between the two key schedules"


def test_beam_fused_attention_oproj_is_bit_identical_to_the_three_launches(monkeypatch):
    """The B beams of ONE stream are one shared-prefix attention group of B rows; their decode steps run attention + combine + o_proj (+ residual) as one
    launch (csrc/llm_attn.hip llm_attn_oproj_kernel: prefix splits + one workgroup per beam, B merging waves per head, a B-row GEMV from registers).
    Against ISST_FUSE_ATTN_OPROJ=0 (three launches): the same outputs, the same candidate log-probs bit for bit, the same KV in every beam's arena,
    across chunks, a pinned system prompt and evictions."""
    from oracle import agent as oag
    cfg = toy_config()
    B = 3
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=45)
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=100, always_cache_system_prompt=True)
    audio = synth.synthetic_audio(cfg.chunk_samples * 7, stream_id=13)
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_FUSE_ATTN_OPROJ", flag)
        eng = Engine(cfg, max_streams=1, max_multiplier=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=100, max_system_prompt=64, max_beams=B)
        eng.load_weights(w)
        sid = eng.open_stream()
        outs, kvs, traces, ckpts, prev = [], [], [], [], []
        for c in range(7):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            eng.beam_trace_begin(B)
            ids, _ = eng.generate(gen, [sid], [seg], [prompt], [prev[-100:]], system_prompt_size=sys_n if c == 0 else 0)
            traces.append(eng.beam_trace_end())
            outs.append(ids[0])
            prev.extend(ids[0][:-1])
            cur = eng.stream_info(sid)["llm_cache_len"]
            kvs.append([kv_of(eng, sid, cur, beam=b) for b in range(B)])
            ckpts.append(cur)
            ev = oag.evict(ckpts, cur, gen.max_llm_cache_size, True, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                eng.kv_evict(sid, new_size, sys_n)
        eng.close()
        return outs, kvs, traces

    (oa, ka, ta), (ob, kb, tb) = run("3"), run("0")  # 3: the fused launch for beam groups too (= the default)
    assert oa == ob
    for c, (xa, xb) in enumerate(zip(ta, tb)):
        assert len(xa) == len(xb)
        for step, ((va, ia, sa), (vb, ib, sb)) in enumerate(zip(xa, xb)):
            assert np.array_equal(va, vb) and np.array_equal(ia, ib) and np.array_equal(sa, sb), f"chunk {c} step {step}: candidates differ between the fused launch and the three launches"
    for c, (xa, xb) in enumerate(zip(ka, kb)):
        for b in range(B):
            assert torch.equal(xa[b][0], xb[b][0]) and torch.equal(xa[b][1], xb[b][1]), f"chunk {c} beam {b}: KV differs between the fused launch and the three launches"


@pytest.mark.parametrize("target_wgs", [0, 8, 1])
def test_beam_shared_prefix_agrees_with_per_beam_arenas(monkeypatch, target_wgs):
    """The shared-prefix attention (one group per stream + one workgroup per beam) against one group per beam, three streams at once, with
    the default slot splits, with long multi-tile spans (target 8 workgroups: the running-softmax form of the kernel) and FOLDED (target 1: one span per
    (stream, kv head), whose waves walk the beams' own tiles behind the prefix and write the output themselves -- what 16+ streams select by default).  The two cut the
    softmax differently, so a sequence may part at a near-tie (random toy weights give flat distributions: the engine and the oracle part in 2 of 8
    cases too): 18 independent cases, a clear majority must be identical -- a wrong mask or tile would leave none."""
    from infinisst_amd.engine import load_library
    cfg = toy_config()
    B = 4
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=45)

    def run(flag):
        monkeypatch.setenv("ISST_BEAM_SHARED", flag)
        eng = Engine(cfg, max_streams=3, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
        eng.load_weights(w)
        load_library().isst_op_set_attn_tuning(target_wgs)
        outs = []
        try:
            for trial in range(6):  # fresh streams every time: one flip must not drag later chunks along
                sids = [eng.open_stream() for _ in range(3)]
                segs = [synth.synthetic_audio(cfg.chunk_samples, stream_id=70 + 3 * trial + k) for k in range(3)]
                prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
                ids, _ = eng.generate(gen, sids, segs, [prompt] * 3, [[], [], []])
                outs.extend(ids)
                for sid in sids:
                    eng.close_stream(sid)
        finally:
            load_library().isst_op_set_attn_tuning(0)
            eng.close()
        return outs

    a, b = run("1"), run("0")
    same = sum(x == y for x, y in zip(a, b))
    print(f"shared vs per-beam arenas (target {target_wgs}): {same}/{len(a)} identical sequences")
    assert len(a) == 18 and same >= 11


@pytest.mark.parametrize("folded,new_tokens", [(False, 7), (True, 7), (True, 20)])
def test_beam_shared_prefix_survives_evictions_and_ring_wrap(folded, new_tokens):
    """Shared-prefix beam attention over a small, wrapping KV ring: 10 chunks with an eviction after most of them, so that the per-beam
    tail tiles straddle the ring's end and the pinned system prompt.  `folded`: the form in which ONE workgroup per (stream, kv head) walks the prefix and
    then the beams' own tiles (forced through the span knob; 20 new tokens: own keys over two or three tiles, some of them wrapping).  After every chunk each beam's arena must hold the KV of the winner
    path -- compared (within two bf16 roundings of a row's largest element, see test_beam_kv_bookkeeping_is_exact) with a greedy stream
    that is teacher-forced along the same path and evicted in lockstep."""
    from oracle import agent as oag
    cfg = toy_config().replace(eos_ids=())
    B = 4
    from infinisst_amd.engine import load_library
    gen_b = GenConfig(max_new_tokens=new_tokens, beam=B, max_llm_cache_size=150, always_cache_system_prompt=True)
    gen_g = GenConfig(max_new_tokens=new_tokens, beam=1, max_llm_cache_size=150, always_cache_system_prompt=True)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=46)
    eng = Engine(cfg, max_streams=2, max_multiplier=1, max_prompt_len=96, max_new_tokens=max(8, new_tokens), max_llm_cache_size=150, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    lib = load_library()
    a, g = eng.open_stream(), eng.open_stream()
    audio = synth.synthetic_audio(cfg.chunk_samples * 10, stream_id=21)
    sys_n = len(synth.system_prompt_ids(cfg))
    prev, ckpts, evictions, ring_starts = [], [], 0, set()
    for c in range(10):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        pin = sys_n if c == 0 else 0
        lib.isst_op_set_attn_tuning(1 if folded else 0)
        try:
            outs, _ = eng.generate(gen_b, [a], [seg], [prompt], [prev[-100:]], system_prompt_size=pin)
        finally:
            lib.isst_op_set_attn_tuning(0)
        win = outs[0]
        outs_g, _ = eng.generate(gen_g, [g], [seg], [prompt], [prev[-100:]], system_prompt_size=pin, forced_tokens=[win])
        assert outs_g[0] == win
        na, ng = eng.stream_info(a)["llm_cache_len"], eng.stream_info(g)["llm_cache_len"]
        assert na == ng
        kg, vg = kv_of(eng, g, ng)
        for b in range(B):
            kb, vb = kv_of(eng, a, na, beam=b)
            for name, x, y in (("K", kb, kg), ("V", vb, vg)):
                err = (x.float() - y.float()).abs().amax(dim=-1)
                tol = 2.0 ** -6 * y.float().abs().amax(dim=-1)
                bad = torch.nonzero(err > tol).flatten().tolist()
                assert not bad, f"chunk {c} beam {b}: {name} differs from the replayed winner path at positions {bad[:10]} (cache {na}, worst {float(err.max()):.4f})"
        prev.extend(win[:-1])
        ckpts.append(na)
        ev = oag.evict(ckpts, na, gen_b.max_llm_cache_size, True, sys_n)
        if ev is not None:
            ckpts, new_size = ev
            eng.kv_evict(a, new_size, sys_n)
            eng.kv_evict(g, new_size, sys_n)
            evictions += 1
    assert evictions >= 4, f"only {evictions} evictions: the ring never wrapped"


def test_beam_sample_with_a_cold_temperature_is_beam_search():
    """`--do-sample --beam 4` (patch_hf.py:871-875: 2B draws without replacement from the softmax over all beams' warped scores instead of their top-k).
    At a temperature of 0.02 (colder and fewer than 2B tokens keep a non-zero fp32 probability: torch.multinomial raises, and so does the library) the draws
    take the largest remaining scores with near-certainty, so the head of the drawn set is the head of the top-k set and -- all scores being scaled by the
    same 1 / T -- every decision of the scorer is the beam search's: three chunks per trial (growing cache, the winner's KV carried on),
    the sampled run against the plain beam search of the same engine on the same audio; this exercises the whole sample plumbing (rows to the host,
    warpers, flattening with the beam scores, the draws, the candidate hand-over to the scorer, EOS-closed hypotheses)."""
    cfg = toy_config()
    B = 4
    gen_b = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400)
    gen_s = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400, do_sample=True, temperature=0.02, seed=7)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=48)
    eng = Engine(cfg, max_streams=2, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    same = total = 0
    for trial in range(6):
        a, b = eng.open_stream(), eng.open_stream()
        audio = synth.synthetic_audio(cfg.chunk_samples * 3, stream_id=70 + trial)
        prev = []
        for c in range(3):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            out_b, _ = eng.generate(gen_b, [a], [seg], [prompt], [prev])
            out_s, _ = eng.generate(gen_s, [b], [seg], [prompt], [prev])
            total += 1
            if out_b[0] == out_s[0]:
                same += 1
            else:  # from here on the two streams hold different caches: stop comparing this trial
                print(f"trial {trial} chunk {c}: beam search {out_b[0]} vs cold beam sample {out_s[0]}")
                break
            assert eng.stream_info(a)["llm_cache_len"] == eng.stream_info(b)["llm_cache_len"]
            prev = prev + out_b[0][:-1]
        eng.close_stream(a)
        eng.close_stream(b)
    print(f"cold beam sample == beam search on {same} of {total} chunks")
    assert same >= total - 1 and total >= 16
    eng.close()


def test_beam_sample_is_seeded_reproducible_and_refuses_what_torch_refuses():
    """Temperature 1: the draws are a pure function of (seed, stream, chunk, step, draw) -- the same call twice gives the same tokens, another seed
    gives other tokens, every token is a valid id and the cache advances like a beam search's.  top_k = 1 leaves one live token per beam, fewer than
    the 2B draws the loop asks for: torch.multinomial raises there, the library fails loudly too."""
    from infinisst_amd.engine import IsstError
    cfg = toy_config()
    B = 4
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=49)
    eng = Engine(cfg, max_streams=3, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=400, max_system_prompt=64, max_beams=B)
    eng.load_weights(w)
    audio = synth.synthetic_audio(cfg.chunk_samples, stream_id=81)
    prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
    outs = {}
    for seed in (1, 1, 2, 3):
        sid = eng.open_stream()
        gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=400, do_sample=True, temperature=1.0, top_k=50, top_p=0.98, seed=seed)
        o, _ = eng.generate(gen, [sid], [audio], [prompt], [[]])
        assert all(0 <= t < cfg.vocab for t in o[0]) and 1 <= len(o[0]) <= 6
        assert eng.stream_info(sid)["llm_cache_len"] == len(prompt) + len(o[0]) - 1
        eng.close_stream(sid)
        if seed in outs:
            assert outs[seed] == o[0], "the same seed, stream slot and chunk must draw the same tokens"
        outs[seed] = o[0]
    assert len({tuple(v) for v in outs.values()}) >= 2, f"three seeds, one continuation: {outs}"
    sid = eng.open_stream()
    with pytest.raises(IsstError, match="non-zero probability"):
        eng.generate(GenConfig(max_new_tokens=6, beam=B, do_sample=True, top_k=1, seed=1), [sid], [audio], [prompt], [[]])
    eng.close()


@pytest.mark.parametrize("B,n_streams", [(4, 1), (3, 3), (4, 6), (6, 2)])  # (6 beams: the 8-beam form of the one-launch reorder kernel)
def test_beam_device_scorer_is_bit_identical_to_the_host_scorer(monkeypatch, B, n_streams):
    """The scorer of a beam step runs on the device (csrc/beam.hip beam_select_kernel: merge of the rows' candidates, EOS hypotheses with their tail
    copies, BeamHypotheses.add / is_done, the reorder of tails and token sequences, the next pass's rows) and the stream never waits for the host; the
    host re-derives every step from the logged candidates and fails the call on any disagreement.  Against ISST_BEAM_DEVICE=0 -- the host deciding every
    step between two stream synchronisations, the round-1..4 loop that beam_scorer.npz / beam_loop.npz pin to patch_hf.py:43-157,278-302 -- the same
    tokens, the same candidate lists bit for bit, the same cache lengths and the same KV in EVERY beam's arena: over chunks, many EOS ids (hypotheses
    close early, their tails travel through buffers, streams finish at different steps and ride along), a pinned system prompt and evictions."""
    from oracle import agent as oag
    cfg = toy_config().replace(eos_ids=(1001, 1008, 1009, 7, 8, 9) if B == 4 else (1001, 7))
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=51)
    w["lm_head.weight"][list(cfg.eos_ids)] *= 2.0  # EOS among the leading candidates often enough that hypotheses close
    gen = GenConfig(max_new_tokens=7, beam=B, max_llm_cache_size=110, always_cache_system_prompt=True, no_repeat_ngram_size=3, repetition_penalty=1.2)
    audio = [synth.synthetic_audio(cfg.chunk_samples * 6, stream_id=30 + i) for i in range(n_streams)]
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_BEAM_DEVICE", flag)
        eng = Engine(cfg, max_streams=n_streams, max_multiplier=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=110, max_system_prompt=64, max_beams=B)
        eng.load_weights(w)
        sids = [eng.open_stream() for _ in range(n_streams)]
        outs, kvs, traces, lens = [], [], [], []
        ckpts = [[] for _ in sids]
        prev = [[] for _ in sids]
        for c in range(6):
            segs = [a[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for a in audio]
            prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            eng.beam_trace_begin(B)
            ids, _ = eng.generate(gen, sids, segs, [prompt] * n_streams, [p[-100:] for p in prev], system_prompt_size=sys_n if c == 0 else 0)
            traces.append(eng.beam_trace_end())
            outs.append(ids)
            for i, sid in enumerate(sids):
                prev[i].extend(ids[i][:-1])
                cur = eng.stream_info(sid)["llm_cache_len"]
                lens.append(cur)
                kvs.append([kv_of(eng, sid, cur, beam=b) for b in range(B)])
                ckpts[i].append(cur)
                ev = oag.evict(ckpts[i], cur, gen.max_llm_cache_size, True, sys_n)
                if ev is not None:
                    ckpts[i], new_size = ev
                    eng.kv_evict(sid, new_size, sys_n)
        eng.close()
        return outs, kvs, traces, lens

    (oa, ka, ta, la), (ob, kb, tb, lb) = run("1"), run("0")
    assert oa == ob and la == lb, "the device scorer and the host scorer chose different tokens / cache lengths"
    assert any(len(x) < gen.max_new_tokens for chunk in oa for x in chunk), "no hypothesis was closed by EOS: the tail buffers were not exercised"
    for c, (xa, xb) in enumerate(zip(ta, tb)):
        assert len(xa) == len(xb), f"chunk {c}: {len(xa)} scorer steps on the device against {len(xb)} on the host"
        for step, ((va, ia, sa), (vb, ib, sb)) in enumerate(zip(xa, xb)):
            assert np.array_equal(va, vb) and np.array_equal(ia, ib) and np.array_equal(sa, sb), f"chunk {c} step {step}: candidates / beam scores differ"
    for q, (xa, xb) in enumerate(zip(ka, kb)):
        for b in range(B):
            assert torch.equal(xa[b][0], xb[b][0]) and torch.equal(xa[b][1], xb[b][1]), f"state {q} beam {b}: KV differs between the device scorer and the host scorer"
