"""Property tests (hypothesis) of the host logic that sits between the reference's agent and the library: whole-chunk eviction, the
speech splice's row map and the sampling warpers, each against the oracle's restatement of the reference on RANDOM inputs -- the fixed
fixtures under tests/golden/ pin the oracle, these widen the HIP side's host code around them.  No GPU: the C entry points used here
(isst_op_splice_map, isst_op_warp_sample) are host functions of the library."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from infinisst_amd import engine as E
from infinisst_amd.config import toy_config
from infinisst_amd.streams import effective_new_cache_size, evict_whole_chunks
from oracle import agent as oag
from oracle import generate as ogen
from oracle import llm as ollm

SETTINGS = dict(max_examples=150, deadline=None)


@settings(**SETTINGS)
@given(chunks=st.lists(st.integers(min_value=3, max_value=90), min_size=1, max_size=60), budget=st.integers(min_value=40, max_value=400),
       sys_n=st.integers(min_value=0, max_value=30), keep=st.booleans())
def test_eviction_walk_equals_the_reference_loop(chunks, budget, sys_n, keep):
    """One utterance of random chunk lengths: after every chunk the product's checkpoint walk (streams.evict_whole_chunks, used by the agent and by
    StreamBatch) and the oracle's literal restatement of agents/infinisst.py:340-352 hold the same list and ask for the same tail; the cache never
    exceeds budget + one chunk, the pinned prefix survives, and the tail always ends on a chunk boundary."""
    keep_n = sys_n if keep else 0
    ours, ref, cur = [], [], 0
    first = True
    for n in chunks:
        cur += n + (sys_n if first else 0)
        first = False
        ref.append(cur)
        got = oag.evict(ref, cur, budget, keep, sys_n)
        ours, new_size = evict_whole_chunks(ours, cur, budget, keep_n)
        if got is None:
            assert new_size is None and ours == ref
            continue
        ref, ref_size = got
        assert ours == ref
        # (the oracle returns the raw tail; the product clamps it to what can be evicted: one chunk longer than the whole budget, or a tail
        #  that would reach into the pinned prefix -- DESIGN.md, streams.effective_new_cache_size)
        assert new_size == effective_new_cache_size(ref_size, cur, keep_n)
        if 0 < ref_size <= cur - keep_n:
            assert new_size == ref_size
        assert new_size <= budget or ref_size > budget  # only a single over-long chunk can leave more than the budget behind
        cur = new_size + keep_n
        if ours:
            assert ours[-1] == cur, "the last checkpoint is the new cache length"
        assert all(a < b for a, b in zip(ours, ours[1:]))


@settings(**SETTINGS)
@given(new_size=st.integers(min_value=-500, max_value=500), cur=st.integers(min_value=0, max_value=400), keep=st.integers(min_value=0, max_value=60))
def test_effective_tail_is_always_evictable(new_size, cur, keep):
    got = effective_new_cache_size(new_size, cur, keep)
    assert 0 <= got <= max(0, cur - keep)
    if 0 < new_size <= cur - keep:
        assert got == new_size


@settings(**SETTINGS)
@given(data=st.data())
def test_splice_row_map_equals_the_oracle_on_random_prompts(data):
    """Random chat-shaped prompts -- system text, then 1..3 turns `<|start_header_id|> user h h <region> e <|start_header_id|> assistant text` with regions
    of 0..12 tokens -- and random numbers of speech features (fewer, as many, more than region slots: the reference's slices then shorten the sequence and
    later turns are cut at the ORIGINAL indices).  The library's host row map applied to an embedding table must equal the oracle's torch.cat restatement
    of SpeechLlamaModel.forward (model/llm.py:86-113)."""
    cfg = toy_config()
    user, assist, sh = cfg.user_id, cfg.assistant_id, cfg.start_header_id
    plain = st.integers(min_value=20, max_value=200).filter(lambda t: t not in (user, assist, sh))
    ids = list(data.draw(st.lists(plain, min_size=0, max_size=6)))
    slots = 0
    for _ in range(data.draw(st.integers(min_value=1, max_value=3))):
        n = data.draw(st.integers(min_value=0, max_value=12))
        slots += n
        ids += [sh, user] + data.draw(st.lists(plain, min_size=2, max_size=2)) + data.draw(st.lists(plain, min_size=n, max_size=n))
        ids += data.draw(st.lists(plain, min_size=1, max_size=1)) + [sh, assist] + data.draw(st.lists(plain, min_size=0, max_size=4))
    n_feat = data.draw(st.integers(min_value=0, max_value=slots + 5))
    ids = np.asarray(ids, dtype=np.int32)
    g = torch.Generator().manual_seed(len(ids) * 131 + n_feat)
    table = torch.randn(int(max(user, assist, sh, 200)) + 1, 8, generator=g)
    feats = torch.randn(n_feat, 8, generator=g)
    ref = ollm.splice_speech(cfg, torch.from_numpy(ids), table[ids], feats)
    m = E.op_splice_map(ids, user, assist, sh, n_feat)
    got = torch.stack([table[ids[t]] if t >= 0 else feats[-1 - t] for t in m]) if len(m) else torch.zeros(0, 8)
    assert got.shape == ref.shape and torch.equal(got, ref)


@settings(max_examples=60, deadline=None)
@given(seed=st.integers(min_value=0, max_value=10 ** 6), vocab=st.integers(min_value=8, max_value=700), temp=st.sampled_from([0.5, 0.7, 1.0, 1.3]),
       top_k=st.sampled_from([0, 1, 3, 50]), top_p=st.sampled_from([0.3, 0.8, 0.95, 1.0]), eps=st.sampled_from([0.0, 1e-3, 0.02]),
       u=st.floats(min_value=0.0, max_value=0.999999, allow_nan=False))
def test_host_warpers_equal_the_oracle_on_random_scores(seed, vocab, temp, top_k, top_p, eps, u):
    """csrc/warp.hip (host): Temperature -> TopK -> TopP -> Epsilon and the inverse-CDF draw against oracle/generate.py (itself pinned to transformers' own
    warper classes by sampling_warpers.npz) on random processed scores, -inf entries (suppressed / banned tokens) included."""
    rng = np.random.default_rng(seed)
    sc = (rng.standard_normal(vocab) * 3).astype(np.float32)
    sc[rng.random(vocab) < 0.1] = -np.inf
    if not np.isfinite(sc).any():
        sc[0] = 0.0
    ref_scores = ogen.warp_logits(torch.from_numpy(sc.copy()), temp, top_k, top_p, eps)
    got, tok = E.op_warp_sample(sc.copy(), temp, int(top_k), top_p, eps, u)
    assert np.array_equal(np.isinf(got), torch.isinf(ref_scores).numpy()), "kept set"
    fin = np.isfinite(got)
    assert np.allclose(got[fin], ref_scores.numpy()[fin], rtol=1e-6, atol=1e-6)
    assert tok == ogen.draw(ref_scores, u) or _near_a_cdf_edge(ref_scores, u)


@settings(max_examples=80, deadline=None)
@given(seed=st.integers(min_value=0, max_value=10 ** 6), vocab=st.integers(min_value=8, max_value=500), temp=st.sampled_from([0.6, 1.0, 1.4]),
       top_k=st.sampled_from([0, 2, 40]), top_p=st.sampled_from([0.05, 0.7, 1.0]), eps=st.sampled_from([0.0, 0.01, 0.3]), min_keep=st.sampled_from([1, 2, 4]))
def test_host_warpers_with_min_tokens_to_keep_equal_the_oracle(seed, vocab, temp, top_k, top_p, eps, min_keep):
    """Under beam search HF builds the same warpers with min_tokens_to_keep = eos ids + 1 (pinned for the oracle by sampling_warpers.npz cases 8-11 and
    by the beam-sample cases of beam_loop.npz): csrc/warp.hip against oracle.generate.warp_logits, aggressive cut-offs included."""
    rng = np.random.default_rng(seed)
    sc = (rng.standard_normal(vocab) * 3).astype(np.float32)
    sc[rng.random(vocab) < 0.1] = -np.inf
    sc[:min_keep] = np.sort(rng.standard_normal(min_keep).astype(np.float32))  # (at least min_keep finite entries)
    ref = ogen.warp_logits(torch.from_numpy(sc.copy()), temp, top_k, top_p, eps, min_tokens_to_keep=min_keep).numpy()
    got = E.op_warp(sc, temp, int(top_k), top_p, eps, min_keep)
    assert np.array_equal(np.isinf(got), np.isinf(ref)), "kept set"
    fin = np.isfinite(ref)
    assert np.allclose(got[fin], ref[fin], rtol=1e-6, atol=1e-6) and int(fin.sum()) >= min(min_keep, int(np.isfinite(sc).sum()))


@settings(max_examples=120, deadline=None)
@given(seed=st.integers(min_value=0, max_value=10 ** 6), n=st.integers(min_value=4, max_value=4000), k=st.integers(min_value=1, max_value=16))
def test_draws_without_replacement_equal_the_oracle(seed, n, k):
    """The beam-sample draw (patch_hf.py:871-873: torch.multinomial without replacement over all beams' scores) as csrc/warp.hip does it against
    oracle.generate.multinomial_without_replacement: same picks in the same order, all distinct, never an entry of zero probability; an impossible
    request (fewer live entries than draws) is refused by both."""
    rng = np.random.default_rng(seed)
    sc = (rng.standard_normal(n) * 4).astype(np.float32)
    sc[rng.random(n) < 0.3] = -np.inf
    us = rng.random(k).tolist()
    try:
        ref = ogen.multinomial_without_replacement(torch.from_numpy(sc), k, us)
    except (RuntimeError, IndexError):
        ref = None  # fewer live entries than draws (probabilities that underflow to zero count as dead): torch raises, so must the library
    if ref is None:
        with pytest.raises(E.IsstError):
            E.op_multinomial_wor(sc, k, us)
        return
    got = E.op_multinomial_wor(sc, k, us)
    assert got == ref and len(set(got)) == k and all(np.isfinite(sc[i]) for i in got)


def _near_a_cdf_edge(scores, u, margin=1e-6):
    p = torch.softmax(scores.float(), dim=-1)
    cdf = torch.cumsum(p, dim=-1)
    return bool(((cdf - u).abs() < margin).any())
