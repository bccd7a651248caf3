"""End-to-end parity of the HIP path (through the C ABI) against the CPU oracle on a real MI355X.

Tolerances (bf16 model, fp32 accumulation on both sides; the oracle rounds after every torch op exactly like
the reference, the kernels round at the same points except inside attention):
  * encoder features / hidden states: |d| <= 0.06 + 2% of |ref|   (values are O(1))
  * logits: |d| <= 0.15 (logits are O(1..5) and are themselves bf16-rounded, ulp up to 0.03)
  * greedy ids: identical wherever the oracle's top-2 margin exceeds 2 x the logit tolerance.
"""
import numpy as np
import pytest
import torch

from infinisst_amd import synth
from infinisst_amd.agent import InfiniSST, WriteAction, default_args, feed_segments
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine, IsstError
from oracle import agent as oag
from oracle import generate as ogen
from oracle import llm as ollm
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu

LOGIT_TOL = 0.15


def report(name, got, ref):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    return f"{name}: max|d|={float(err.max()):.4f} mean|d|={float(err.mean()):.5f} max|ref|={float(ref.abs().max()):.3f}"


def assert_close(name, got, ref, atol, rtol):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{name}: {got.shape} vs {ref.shape}"
    err = (got - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), report(name, got, ref) + f" ({int(bad.sum())}/{bad.numel()} out of tolerance)"


def make_engine(cfg, w, **kw):
    args = dict(max_streams=2, max_multiplier=2, max_prompt_len=96, max_new_tokens=16, max_llm_cache_size=150,
                max_system_prompt=64, debug_taps=True)
    args.update(kw)
    eng = Engine(cfg, **args)
    eng.load_weights(w)
    return eng


@pytest.mark.parametrize("block,cache,mode", [(16, 40, "bf16"), (48, 576, "bf16"), (16, 40, "fp32")])
def test_encoder_streaming_matches_oracle(block, cache, mode):
    """conv stack + 399-sample history + encoder ring (first chunk / growing / saturated window) + shrink + proj."""
    cfg = toy_config().replace(block_size=block, max_cache_size=cache, enc_rope_mode=mode)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.08, norm_jitter=0.1, seed=11)
    eng = make_engine(cfg, w)
    sid = eng.open_stream()
    n_chunks = 6 if block == 16 else 3
    audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=1)
    cache_o, rope = oenc.new_cache(cfg), oenc.make_rope(cfg)
    for c in range(n_chunks):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref, cache_o, inter = oenc.encode_speech(w, cfg, x.unsqueeze(0).bfloat16(), cache_o, 1, rope, return_intermediates=True)
        got = eng.encode_speech(sid, seg)
        Q = block
        msgs = []
        for name, r in [("conv_out", inter["conv"][0]), ("post_proj", inter["post_proj"][0])] + [
                (f"enc_layer_{i}", inter["layers"][i][0]) for i in range(cfg.enc_layers)] + [
                ("enc_out", inter["enc_out"][0]), ("shrink", inter["shrink"][0])]:
            t = eng.debug_tap(name).view(r.shape)
            msgs.append(report(name, t, r))
        print(f"chunk {c}: " + " | ".join(msgs))
        assert_close(f"chunk {c} conv_out", eng.debug_tap("conv_out").view(Q, -1), inter["conv"][0], 0.03, 0.02)
        assert_close(f"chunk {c} enc_out", eng.debug_tap("enc_out").view(Q, -1), inter["enc_out"][0], 0.06, 0.02)
        assert_close(f"chunk {c} speech features", got, ref[0], 0.06, 0.02)
        info = eng.stream_info(sid)
        assert info["enc_n_steps"] == cache_o.n_steps
        assert info["enc_cache_len"] == cache_o.layers[0].k.shape[1]


def test_encoder_without_rope_matches_oracle():
    """--rope 0 (patch_speech_encoder.py:488-493, :823): the bf16 sinusoid of the stream position is added to the encoder input and q / k are
    left alone.  20 chunks of 16 frames cross position 256 (where bf16 positions start to repeat); the state is then moved to frame 22491
    (a 7.5-minute stream) on both sides and two more chunks compared.  The position rows themselves must be exact: enc_layer_0's input is
    post_proj + row, so a wrong row shows up as an O(1) error there."""
    cfg = toy_config().replace(block_size=16, max_cache_size=40, enc_rope=False)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.08, norm_jitter=0.1, seed=12)
    eng = make_engine(cfg, w)
    sid = eng.open_stream()
    n_chunks = 22
    audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=2)
    cache_o, rope = oenc.new_cache(cfg), oenc.make_rope(cfg)
    assert rope is oenc.NO_ROPE
    for c in range(n_chunks):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        if c == 20:  # jump: same window, same audio history, positions from 22491 on
            cache_o.n_steps = 22491
            eng.import_speech_cache(sid, [(l.k, l.v) for l in cache_o.layers], n_steps=22491, audio_tail=cache_o.src[0, -cfg.first_chunk_offset:])
        before = cache_o.n_steps
        ref, cache_o, inter = oenc.encode_speech(w, cfg, x.unsqueeze(0).bfloat16(), cache_o, 1, rope, return_intermediates=True)
        got = eng.encode_speech(sid, seg)
        Q = cfg.block_size
        # the row added to the input, recovered from the taps: layer 0 = x + attn(...) + ffn(...), so compare through the oracle's own layer 0
        assert_close(f"chunk {c} post_proj", eng.debug_tap("post_proj").view(Q, -1), inter["post_proj"][0], 0.03, 0.02)
        assert_close(f"chunk {c} enc_layer_0", eng.debug_tap("enc_layer_0").view(Q, -1), inter["layers"][0][0], 0.06, 0.02)
        assert_close(f"chunk {c} enc_out", eng.debug_tap("enc_out").view(Q, -1), inter["enc_out"][0], 0.06, 0.02)
        assert_close(f"chunk {c} speech features", got, ref[0], 0.06, 0.02)
        info = eng.stream_info(sid)
        assert info["enc_n_steps"] == cache_o.n_steps == before + Q
        assert info["enc_cache_len"] == cache_o.layers[0].k.shape[1]
    # and the rotary build of the same weights gives something else (the switch is live)
    cfg_r = cfg.replace(enc_rope=True)
    eng_r = make_engine(cfg_r, w)
    sid_r = eng_r.open_stream()
    a = eng_r.encode_speech(sid_r, audio[:cfg.chunk_samples])
    eng2 = make_engine(cfg, w)
    b = eng2.encode_speech(eng2.open_stream(), audio[:cfg.chunk_samples])
    assert float((a.float() - b.float()).abs().max()) > 0.05


def run_chunks(cfg, gen, w, eng, sid, n_chunks, forced: bool, evict: bool, sys_pin: bool, audio_id=0):
    """Drive oracle and engine side by side; returns per-step (oracle logits, engine logits, oracle tok, engine tok)."""
    audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=audio_id)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), oenc.make_rope(cfg)
    sys_n = len(synth.system_prompt_ids(cfg))
    ckpts, prev_targets, records = [], [], []
    for c in range(n_chunks):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        enc_ids = prev_targets[-gen.no_repeat_ngram_lookback:]
        ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, enc_ids)
        ref_new = ref.sequences[len(prompt):]
        outs, logits = eng.generate(gen, [sid], [seg], [prompt], [enc_ids], system_prompt_size=sys_n if (sys_pin and c == 0) else 0,
                                    forced_tokens=[ref_new] if forced else None, return_logits=True)
        for s in range(min(len(ref_new), len(outs[0]))):
            records.append((c, s, ref.step_logits[s].float().numpy(), logits[0, s], ref.step_scores[s].numpy(), ref_new[s], outs[0][s]))
        if forced:
            assert outs[0] == ref_new
        prev_targets.extend(ref_new[:-1])
        cur = ollm.kv_len(kv)
        assert eng.stream_info(sid)["llm_cache_len"] == cur, f"chunk {c}: cache length {eng.stream_info(sid)['llm_cache_len']} vs {cur}"
        ckpts.append(cur)
        if evict:
            ev = oag.evict(ckpts, cur, gen.max_llm_cache_size, sys_pin, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                for layer in kv:
                    for j in (0, 1):
                        t = layer[j]
                        tail = t[:, :, -new_size:]
                        layer[j] = torch.cat([t[:, :, :sys_n], tail], dim=2) if sys_pin else tail
                eng.kv_evict(sid, new_size, sys_n if sys_pin else 0)
                assert eng.stream_info(sid)["llm_cache_len"] == ollm.kv_len(kv)
    return records


@pytest.mark.parametrize("sys_pin", [True, False])
def test_generate_teacher_forced_logits(sys_pin):
    """Prefill + decode over 8 chunks with rolling KV eviction (positions re-index), teacher-forced with the oracle's
    tokens so every step sees the same context: logits must agree within LOGIT_TOL."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=150, always_cache_system_prompt=sys_pin)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=21)
    eng = make_engine(cfg, w, debug_taps=False)
    sid = eng.open_stream()
    recs = run_chunks(cfg, gen, w, eng, sid, 8, forced=True, evict=True, sys_pin=sys_pin)
    worst = max(float(np.abs(r[2] - r[3]).max()) for r in recs)
    print(f"teacher-forced: {len(recs)} steps, worst |logit diff| = {worst:.4f}")
    assert worst <= LOGIT_TOL
    agree = sum(int(np.argmax(r[3]) == np.argmax(r[2])) for r in recs)
    print(f"raw-logit argmax agreement {agree}/{len(recs)}")


def test_generate_free_running_tokens():
    """Free-running greedy decode over 5 chunks with evictions: ids identical to the oracle's wherever the oracle's top-2 margin exceeds
    2*LOGIT_TOL (near-ties may legitimately flip under bf16).  Weights: the "peaked" recipe (synth.apply_recipe), under which all but the
    steps whose structured continuations are banned by the n-gram processors are decisive -- the test FAILS when fewer than 24 of
    the 40 steps are (round 2 ran it on the plain init: 3 decisive steps)."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=22, recipe="peaked")
    eng = make_engine(cfg, w, debug_taps=False)
    sid = eng.open_stream()
    recs = run_chunks(cfg, gen, w, eng, sid, 5, forced=False, evict=True, sys_pin=True)
    decisive = mismatch = steps = 0
    for c, s, rl, gl, sc, rt, gt in recs:
        steps += 1
        top2 = np.sort(sc[np.isfinite(sc)])[-2:]
        if top2[1] - top2[0] > 2 * LOGIT_TOL:
            decisive += 1
            mismatch += int(rt != gt)
        if rt != gt:
            break  # contexts diverge after the first flip
    print(f"free-running: {steps} steps compared, {decisive} decisive, {mismatch} mismatches")
    assert mismatch == 0
    assert decisive >= 24, f"only {decisive} decisive steps"


def test_two_streams_batched_equals_single():
    """Two streams stepped together (shared weights, per-stream KV) produce the same logits as each stream alone
    (to fp32 accumulation order)."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=5, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=23)
    eng = make_engine(cfg, w, debug_taps=False, max_streams=4)
    a, b, a2, b2 = (eng.open_stream() for _ in range(4))
    audio = [synth.synthetic_audio(cfg.chunk_samples * 3, stream_id=i) for i in (0, 1)]
    for c in range(3):
        segs = [x[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for x in audio]
        pa = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        # stream b starts one chunk later than stream a: ragged prompts (first-chunk system prompt vs later chunk)
        outs_s, logits_s = [], []
        o, l = eng.generate(gen, [a2], [segs[0]], [pa], [[]], return_logits=True)
        outs_s.append(o[0]); logits_s.append(l[0])
        if c >= 1:
            pb = synth.chunk_prompt_ids(cfg, 1, first=(c == 1))
            o, l = eng.generate(gen, [b2], [segs[1]], [pb], [[]], forced_tokens=None, return_logits=True)
            outs_s.append(o[0]); logits_s.append(l[0])
            outs, logits = eng.generate(gen, [a, b], segs, [pa, pb], [[], []], forced_tokens=[outs_s[0], outs_s[1]], return_logits=True)
        else:
            outs, logits = eng.generate(gen, [a], [segs[0]], [pa], [[]], forced_tokens=[outs_s[0]], return_logits=True)
        for i in range(len(outs)):
            n = min(len(outs[i]), len(outs_s[i]))
            d = float(np.abs(logits[i][:n] - logits_s[i][:n]).max())
            print(f"chunk {c} stream {i}: batched vs single max|d| = {d:.4f}")
            assert d <= 0.07


def test_device_resident_audio_is_bit_identical_to_host_audio():
    """isst_gen_params.pcm_on_device: audio the caller already holds in HBM (bench.py's timed region, a capture pipeline) is read in place;
    tokens and logits must equal the host-array hand-over (reference agents/infinisst.py:222 moves host tensors itself) bit for bit --
    3 chunks of 2 streams (history carried between chunks), then a mixed call must be refused."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=4, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=29)
    eng = make_engine(cfg, w, debug_taps=False, max_streams=4)
    h0, h1, d0, d1 = (eng.open_stream() for _ in range(4))
    audio = [synth.synthetic_audio(cfg.chunk_samples * 3, stream_id=i) for i in (5, 6)]
    audio_dev = [torch.from_numpy(x).cuda() for x in audio]
    for c in range(3):
        sl = slice(c * cfg.chunk_samples, (c + 1) * cfg.chunk_samples)
        p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        oh, lh = eng.generate(gen, [h0, h1], [x[sl] for x in audio], [p, p], [[], []], return_logits=True)
        od, ld = eng.generate(gen, [d0, d1], [x[sl] for x in audio_dev], [p, p], [[], []], return_logits=True)
        assert oh == od
        for i in range(2):
            assert np.array_equal(lh[i][:len(oh[i])], ld[i][:len(od[i])])
    with pytest.raises(IsstError):  # non-contiguous device audio
        eng.generate(gen, [d0], [audio_dev[0][::2][:cfg.chunk_samples]], [p], [[]])


@pytest.mark.parametrize("n", [3, 17])
def test_batch_is_deterministic_and_independent_of_stream_order(n):
    """Size-independent properties of the batched path: (1) the same call sequence on fresh streams gives the same bits twice; (2) a stream's
    tokens and logits do not depend on WHERE in the batch it sits -- the same streams stepped in reverse order give every stream the same
    bits (every reduction in the path runs in an order fixed by the stream's own data: K slices, split-KV slabs, k-step waves).  3 streams run the
    skinny kernels (3 rows), 17 the 13..64-row machinery; three chunks with a pinned system prompt and an eviction."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=4, max_llm_cache_size=60)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=31)
    eng = make_engine(cfg, w, debug_taps=False, max_streams=3 * n, max_llm_cache_size=60)
    sys_n = len(synth.system_prompt_ids(cfg))
    audio = [synth.synthetic_audio(cfg.chunk_samples * 3, stream_id=40 + i) for i in range(n)]

    def run(order):
        sids = [eng.open_stream() for _ in range(n)]
        logs = [[] for _ in range(n)]
        toks = [[] for _ in range(n)]
        for c in range(3):
            p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            segs = [audio[i][c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for i in order]
            outs, lg = eng.generate(gen, [sids[i] for i in order], segs, [p] * n, [[]] * n, system_prompt_size=sys_n if c == 0 else 0,
                                    return_logits=True)
            for pos, i in enumerate(order):
                toks[i].append(outs[pos])
                logs[i].append(lg[pos][:len(outs[pos])].copy())
                if c == 1:  # whole-chunk eviction between chunks, as the agent would do it
                    cur = eng.stream_info(sids[i])["llm_cache_len"]
                    eng.kv_evict(sids[i], cur - sys_n - 20, sys_n)
        for sid in sids:
            eng.close_stream(sid)
        return toks, logs

    fwd = list(range(n))
    t1, l1 = run(fwd)
    t2, l2 = run(fwd)
    t3, l3 = run(fwd[::-1])
    for i in range(n):
        assert t1[i] == t2[i] == t3[i], f"stream {i}: tokens differ between runs / batch orders"
        for c in range(3):
            assert np.array_equal(l1[i][c], l2[i][c]), f"stream {i} chunk {c}: not deterministic"
            assert np.array_equal(l1[i][c], l3[i][c]), f"stream {i} chunk {c}: logits depend on the stream's position in the batch"


def test_many_streams_decode_rows_beyond_the_library_threshold():
    """170 streams in one call at toy width: the DECODE passes have 170 rows (> 160: the many-row GEMM path of engine_llm.hip with its SwiGLU and
    residual + RMSNorm passes, last layer's bare residual included), the prefill 170 x prompt rows, the encoder 170 x block rows.  Streams 0, 85 and 169
    are held to the same streams stepped alone (the packed-weight kernels at 1 / few rows) within the batched-vs-single tolerance, two chunks."""
    cfg = toy_config()
    n = 170
    gen = GenConfig(max_new_tokens=4, max_llm_cache_size=150)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=33)
    eng = make_engine(cfg, w, debug_taps=False, max_streams=n + 1, max_multiplier=1)
    sids = [eng.open_stream() for _ in range(n)]
    solo = eng.open_stream()
    audio = [synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=300 + i) for i in range(n)]
    picks = (0, 85, 169)
    ref = {i: [] for i in picks}
    for i in picks:  # the picked streams alone, free-running: their tokens teacher-force the batch
        eng.reset_stream(solo)
        for c in range(2):
            p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            o, l = eng.generate(gen, [solo], [audio[i][c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]], [p], [[]], return_logits=True)
            ref[i].append((o[0], l[0]))
    for c in range(2):
        p = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        forced = [ref[i][c][0] if i in ref else None for i in range(n)]
        outs, logits = eng.generate(gen, sids, [a[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for a in audio], [p] * n, [[]] * n,
                                    forced_tokens=forced, return_logits=True)
        for i in picks:
            o1, l1 = ref[i][c]
            k = min(len(o1), len(outs[i]))
            d = float(np.abs(logits[i][:k] - l1[:k]).max())
            print(f"chunk {c} stream {i} of {n}: batched (library path) vs alone max |d| = {d:.4f}")
            assert d <= 0.07


def test_agent_policy_matches_oracle_agent():
    """InfiniSST.policy over the engine vs OracleAgent.policy: same READ/WRITE actions, cache lengths and
    checkpoints over an utterance with a ragged tail and evictions (ids compared on decisive steps only)."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=24)
    args = default_args(max_llm_cache_size=150, max_new_tokens=6, max_latency_multiplier=1)
    eng = make_engine(cfg, w, debug_taps=False, max_multiplier=1)
    agent = InfiniSST(args, engine=eng, model_cfg=cfg)
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=150)
    sysn = len(synth.system_prompt_ids(cfg))
    oa = oag.OracleAgent(w, cfg, gen, lambda first: synth.chunk_prompt_ids(cfg, 1, first), system_prompt_size=sysn)
    n = cfg.chunk_samples * 5 + 3000
    audio = synth.synthetic_audio(n, stream_id=5)
    st_o = oa.build_states()
    st_o.source_sample_rate = 16000
    st = agent.build_states()
    st.source_sample_rate = 16000
    pos = 0
    while pos < n:
        end = min(pos + cfg.chunk_samples, n)
        seg = audio[pos:end].tolist()
        pos = end
        for s in (st, st_o):
            s.source.extend(seg)
            s.source_finished = pos >= n
        act_o = oa.policy(st_o)
        # keep both sides on the oracle's tokens so that a near-tie flip cannot desynchronise the comparison
        act = agent.policy(st)
        assert type(act).__name__ == type(act_o).__name__
        assert getattr(act, "finished", None) == getattr(act_o, "finished", None)
        info = eng.stream_info(st.stream_id)
        if st.target_ids == st_o.target_ids:
            assert info["llm_cache_len"] == ollm.kv_len(st_o.past_key_values)
            assert agent.cache_checkpoints == oa.cache_checkpoints
        assert info["enc_n_steps"] == st_o.speech_cache.n_steps
    print("agent targets:", st.target_ids, "oracle:", st_o.target_ids)


def test_errors_are_loud():
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, seed=25)
    eng = Engine(cfg, max_streams=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=150, max_system_prompt=64)
    with pytest.raises(IsstError):  # generate before weights are loaded
        eng.generate(GenConfig(max_new_tokens=4), [0], [np.zeros(cfg.chunk_samples, np.float32)], [[1, 2, 3]], [[]])
    bad = dict(w)
    del bad["lm_head.weight"]
    with pytest.raises(IsstError, match="lm_head"):
        eng.load_weights(bad)
    eng.load_weights(w)
    sid = eng.open_stream()
    with pytest.raises(IsstError):  # ragged sample count
        eng.generate(GenConfig(max_new_tokens=4), [sid], [np.zeros(1000, np.float32)], [[1, 2, 3]], [[]])
    with pytest.raises(IsstError):  # no free slot
        eng.open_stream()


def test_decode_step_graph_replay_is_bit_identical(monkeypatch):
    """ISST_GRAPH=1: the decode step (metadata upload, decoder stack, sampling, token download) is captured once and replayed as
    a hipGraph on a non-default stream; tokens and cache lengths must equal the launch-by-launch path exactly."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=41)
    gen = GenConfig(max_new_tokens=7, max_llm_cache_size=150)
    audio = synth.synthetic_audio(cfg.chunk_samples * 4, stream_id=9)

    def run(graph):
        if graph:
            monkeypatch.setenv("ISST_GRAPH", "1")
        else:
            monkeypatch.delenv("ISST_GRAPH", raising=False)
        eng = make_engine(cfg, w, debug_taps=False, max_multiplier=1)
        sid = eng.open_stream()
        out = []
        with torch.cuda.stream(torch.cuda.Stream()):
            for c in range(4):
                seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
                ids, _ = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]])
                out.append((ids[0], eng.stream_info(sid)["llm_cache_len"]))
            torch.cuda.synchronize()
        eng.close()
        return out

    assert run(True) == run(False)


@pytest.mark.parametrize("graph", [False, True])
def test_fused_sampling_tail_equals_the_three_launch_tail(monkeypatch, graph):
    """Up to 16 rows the greedy sampling tail is ONE launch (sample.hip sample_fused_kernel: processors on each block's slice of the row, argmax, last-arriver
    merge, tokens + a sequence number into pinned host memory, the host waits on that number instead of synchronising the stream); ISST_FUSED_SAMPLE=0 keeps the
    three launches + D2H copy + synchronisation.  Same tokens, same cache lengths, chunk after chunk -- launched directly and replayed from a captured graph
    (the sequence number lives on the device, so replays count correctly) -- with repeated tokens in the history (penalty, n-gram bans) and two streams in a call."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=43, recipe="peaked")
    gen = GenConfig(max_new_tokens=7, max_llm_cache_size=150, no_repeat_ngram_size=3)
    audio = [synth.synthetic_audio(cfg.chunk_samples * 5, stream_id=20 + k) for k in range(2)]

    def run(fused):
        monkeypatch.setenv("ISST_FUSED_SAMPLE", "1" if fused else "0")
        if graph:
            monkeypatch.setenv("ISST_GRAPH", "1")
        else:
            monkeypatch.delenv("ISST_GRAPH", raising=False)
        eng = make_engine(cfg, w, debug_taps=False, max_multiplier=1)
        sids = [eng.open_stream(), eng.open_stream()]
        out, prev = [], [[], []]
        with torch.cuda.stream(torch.cuda.Stream()):
            for c in range(5):
                segs = [a[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples] for a in audio]
                ids, _ = eng.generate(gen, sids, segs, [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))] * 2, prev)
                out.append((ids, [eng.stream_info(s_)["llm_cache_len"] for s_ in sids]))
                prev = [prev[k] + ids[k][:-1] for k in range(2)]
                for s_ in sids:
                    if eng.stream_info(s_)["llm_cache_len"] > 120:
                        eng.kv_evict(s_, 60, 0)
            torch.cuda.synchronize()
        eng.close()
        return out

    a, b = run(True), run(False)
    assert a == b
    assert len({tuple(t) for ids, _ in a for t in ids}) > 2


@pytest.mark.parametrize("m,beam,n_streams", [(1, 1, 1), (4, 1, 1), (1, 1, 3), (2, 3, 2)])
def test_rotated_key_arena_filled_by_the_prefill_is_bit_identical_to_the_pre_pass(monkeypatch, m, beam, n_streams):
    """The rotated-key arena of a chunk (the cached keys rotated at their logical position of THIS chunk: patch_llm.py:286-299 rotates every key on every pass)
    is filled by the prefill attention's loader waves on their way to LDS (LlmStreamView::rot_keys == 2) instead of by a pre-pass over every layer's keys.
    ISST_ROPE_FUSE=0 keeps the pre-pass.  Same tokens, same raw logits bit for bit, same cache lengths -- over chunks with evictions (every key re-indexes), a
    pinned system prompt, prompts of one and of two attention units (m = 4: 58 rows = 15 row groups), several streams in a call, and the shared-prefix beam form
    (whose decode passes read arena 0 through the rotated arena)."""
    from oracle import agent as oag
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=61)
    gen = GenConfig(latency_multiplier=m, max_new_tokens=6, beam=beam, max_llm_cache_size=160, always_cache_system_prompt=True)
    audio = [synth.synthetic_audio(cfg.chunk_samples * m * 6, stream_id=40 + i) for i in range(n_streams)]
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_ROPE_FUSE", flag)
        eng = Engine(cfg, max_streams=n_streams, max_multiplier=m, max_prompt_len=sys_n + 24 + 12 * m, max_new_tokens=8, max_llm_cache_size=160, max_system_prompt=64,
                     max_beams=beam)
        eng.load_weights(w)
        sids = [eng.open_stream() for _ in range(n_streams)]
        out = []
        ckpts = [[] for _ in sids]
        prev = [[] for _ in sids]
        for c in range(6):
            segs = [a[c * cfg.chunk_samples * m:(c + 1) * cfg.chunk_samples * m] for a in audio]
            prompt = synth.chunk_prompt_ids(cfg, m, first=(c == 0))
            ids, logits = eng.generate(gen, sids, segs, [prompt] * n_streams, [p[-100:] for p in prev], system_prompt_size=sys_n if c == 0 else 0,
                                       return_logits=(beam == 1))
            lens = [eng.stream_info(sid)["llm_cache_len"] for sid in sids]
            kv = [[eng.read_kv(sid, p_, layer=1, kv_head=1, beam=b) for p_ in (0, sys_n, lens[i] - 1)] for i, sid in enumerate(sids) for b in range(beam)]
            for i, sid in enumerate(sids):
                prev[i].extend(ids[i][:-1])
                cur = lens[i]
                ckpts[i].append(cur)
                ev = oag.evict(ckpts[i], cur, gen.max_llm_cache_size, True, sys_n)
                if ev is not None:
                    ckpts[i], new_size = ev
                    eng.kv_evict(sid, new_size, sys_n)
            out.append((ids, lens, None if logits is None else [logits[i][:len(ids[i])].copy() for i in range(n_streams)], kv))
        eng.close()
        return out

    a, b = run("1"), run("0")
    for c, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[1] == y[1], f"chunk {c}: tokens / cache lengths differ"
        if x[2] is not None:
            for i in range(n_streams):
                assert np.array_equal(x[2][i], y[2][i]), f"chunk {c} stream {i}: logits differ between the prefill-filled arena and the pre-pass"
        for q, (ka, kb) in enumerate(zip(x[3], y[3])):
            for (k1, v1), (k2, v2) in zip(ka, kb):
                assert torch.equal(k1, k2) and torch.equal(v1, v2), f"chunk {c} arena {q}: KV differs"
    assert any(len(set(x[1])) >= 1 for x in a)


def test_one_stream_decode_step_prepared_on_the_device_is_bit_identical(monkeypatch):
    """One stream's greedy loop (BASELINE.json configs[1]): the fused sampling tail appends the sampled id to the id list the processors read and copies its
    embedding row to the decoder's input row ON THE DEVICE, so a decode step is neither preceded by a metadata upload nor opened by the embedding launch
    (engine_llm.hip `advance_on_device`; model/llm.py:114-115 embeds the last token only).  ISST_TAIL_ADVANCE=0 keeps upload + embedding launch.  Same
    tokens, same raw logits bit for bit, same cache lengths over chunks with repeated tokens in the history (penalty / n-gram bans read the id list), a
    previous-target window and evictions."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=43, recipe="peaked")
    gen = GenConfig(max_new_tokens=7, max_llm_cache_size=150, no_repeat_ngram_size=3)
    audio = synth.synthetic_audio(cfg.chunk_samples * 6, stream_id=27)

    def run(flag):
        monkeypatch.setenv("ISST_TAIL_ADVANCE", flag)
        eng = make_engine(cfg, w, debug_taps=False, max_multiplier=1)
        sid = eng.open_stream()
        out, prev = [], []
        with torch.cuda.stream(torch.cuda.Stream()):
            for c in range(6):
                seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
                ids, logits = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [prev[-100:]], return_logits=(c % 2 == 0))
                out.append((ids[0], eng.stream_info(sid)["llm_cache_len"], None if logits is None else logits[0][:len(ids[0])].copy()))
                prev = prev + ids[0][:-1]
                if eng.stream_info(sid)["llm_cache_len"] > 120:
                    eng.kv_evict(sid, 60, 0)
            torch.cuda.synchronize()
        eng.close()
        return out

    a, b = run("1"), run("0")
    assert [(x[0], x[1]) for x in a] == [(x[0], x[1]) for x in b]
    for c, (x, y) in enumerate(zip(a, b)):
        if x[2] is not None:
            assert np.array_equal(x[2], y[2]), f"chunk {c}: logits differ between the device-prepared step and upload + embedding launch"
    assert len({t for ids, _, _ in a for t in ids}) > 3


@pytest.mark.parametrize("target_wgs", [1, 6])
def test_attention_span_forms_match_oracle(target_wgs):
    """The many-stream forms of the decoder attention, forced on one stream through the tuning hook: target 1 = one workgroup per
    (stream, kv head) walking every key tile with the running softmax and writing the output itself (no combine pass); target 6 =
    a few multi-tile spans + combine.  Same teacher-forced run as above, same tolerance."""
    from infinisst_amd.engine import load_library
    lib = load_library()
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=150, always_cache_system_prompt=True)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=21)
    eng = make_engine(cfg, w, debug_taps=False)
    sid = eng.open_stream()
    lib.isst_op_set_attn_tuning(target_wgs)
    try:
        recs = run_chunks(cfg, gen, w, eng, sid, 8, forced=True, evict=True, sys_pin=True)
    finally:
        lib.isst_op_set_attn_tuning(0)
    worst = max(float(np.abs(r[2] - r[3]).max()) for r in recs)
    print(f"attention target {target_wgs}: worst |logit diff| = {worst:.4f}")
    assert worst <= LOGIT_TOL


def test_rotated_key_arena_is_bit_identical_to_rotate_on_read(monkeypatch):
    """ISST_ROT_KEYS=1 (default: cached keys rotated once per chunk into a second arena) against ISST_ROT_KEYS=0 (rotated on every
    read, the reference's schedule): same RoPE function on the same inputs, so logits must be bit-identical -- across chunks, a pinned
    system prompt and evictions that shift every ring key's position."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=43)
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=150, always_cache_system_prompt=True)
    audio = synth.synthetic_audio(cfg.chunk_samples * 7, stream_id=11)
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_ROT_KEYS", flag)
        eng = make_engine(cfg, w, debug_taps=False, max_multiplier=1)
        sids = [eng.open_stream(), eng.open_stream()]
        out, ckpts = [], []
        for c in range(7):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
            ids, logits = eng.generate(gen, sids, [seg, seg], [prompt, prompt], [[], []], system_prompt_size=sys_n if c == 0 else 0, return_logits=True)
            out.append((ids, logits.copy()))
            cur = eng.stream_info(sids[0])["llm_cache_len"]
            ckpts.append(cur)
            ev = oag.evict(ckpts, cur, gen.max_llm_cache_size, True, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                for s in sids:
                    eng.kv_evict(s, new_size, sys_n)
        eng.close()
        return out

    a, b = run("1"), run("0")
    assert len(a) == len(b)
    for (ia, la), (ib, lb) in zip(a, b):
        assert ia == ib
        assert np.array_equal(la, lb)


def test_inline_split_kv_combine_is_bit_identical_to_the_combine_launch(monkeypatch):
    """One-stream decode steps combine their split-KV partials INSIDE the attention launch (last-arriver form: write-through slabs, drained,
    agent-scope arrival counter per kv head, sc1 reads -- csrc/llm_attn.hip).  Against ISST_INLINE_COMBINE=0 (the combine as its own launch):
    identical logits bit for bit over 12 chunks x 8 passes x 2 layers x 2 kv heads of hand-offs, with evictions, at a cache long enough for
    several slot splits.  A stale or torn slab would show up as a differing logit."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=500)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=71)
    audio = synth.synthetic_audio(cfg.chunk_samples * 12, stream_id=5)
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_INLINE_COMBINE", flag)
        eng = make_engine(cfg, w, debug_taps=False, max_llm_cache_size=500, max_streams=1)
        sid = eng.open_stream()
        outs, logs, ckpts = [], [], []
        for c in range(12):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            o, l = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]],
                                system_prompt_size=sys_n if c == 0 else 0, return_logits=True)
            outs.append(o[0])
            logs.append(l[0][:len(o[0])].copy())
            cur = eng.stream_info(sid)["llm_cache_len"]
            ckpts.append(cur)
            ev = oag.evict(ckpts, cur, 200, True, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                eng.kv_evict(sid, new_size, sys_n)
        eng.close()
        return outs, logs

    (oa, la), (ob, lb) = run("1"), run("0")
    assert oa == ob
    for c, (x, y) in enumerate(zip(la, lb)):
        assert np.array_equal(x, y), f"chunk {c}: logits differ between the in-launch combine and the combine launch"


def test_fused_attention_combine_oproj_is_bit_identical_to_the_three_launches(monkeypatch):
    """One stream's decode step runs attention + split-KV combine + o_proj (+ residual) as ONE launch (csrc/llm_attn.hip llm_attn_oproj_kernel: the
    o_proj weights wait in registers, two in-launch hand-offs through write-through stores and grid counters).  Against ISST_FUSE_ATTN_OPROJ=0 (three
    launches): identical logits bit for bit over 12 chunks x 8 passes x 2 layers of hand-offs, with evictions and a wrapping ring.  A stale slab, a
    torn attention row or a counter out of step would show up as a differing logit (or as the launch's time-out error)."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=300)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=72)
    audio = synth.synthetic_audio(cfg.chunk_samples * 12, stream_id=6)
    sys_n = len(synth.system_prompt_ids(cfg))

    def run(flag):
        monkeypatch.setenv("ISST_FUSE_ATTN_OPROJ", flag)
        eng = make_engine(cfg, w, debug_taps=False, max_llm_cache_size=300, max_streams=1)
        sid = eng.open_stream()
        outs, logs, ckpts = [], [], []
        for c in range(12):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            o, l = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]],
                                system_prompt_size=sys_n if c == 0 else 0, return_logits=True)
            outs.append(o[0])
            logs.append(l[0][:len(o[0])].copy())
            cur = eng.stream_info(sid)["llm_cache_len"]
            ckpts.append(cur)
            ev = oag.evict(ckpts, cur, 150, True, sys_n)
            if ev is not None:
                ckpts, new_size = ev
                eng.kv_evict(sid, new_size, sys_n)
        eng.close()
        return outs, logs

    (oa, la), (ob, lb) = run("1"), run("0")
    assert oa == ob
    for c, (x, y) in enumerate(zip(la, lb)):
        assert np.array_equal(x, y), f"chunk {c}: logits differ between the fused launch and the three launches"


def test_fused_launch_that_cannot_complete_falls_back_to_the_three_launches(monkeypatch, capfd):
    """The fused attention + o_proj launch needs all of its workgroups resident and fed; a device shared with another process' kernels (or a CU mask) starves
    it.  ISST_FUSE_AO_TEST_TIMEOUT=1 makes every fused launch wait for an arrival count that never comes: the bounded waits must run out (no hang), and the
    call must NOT fail: nothing of the step is committed when the time-out is noticed, so the handle latches the three-launch path -- saying so
    once on stderr -- re-issues the same pass, and from the first chunk on returns the bits of an engine that never fused."""
    import time
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=300)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=73)
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=8)
    sys_n = len(synth.system_prompt_ids(cfg))

    def chunks(eng, sid):
        outs, logs = [], []
        for c in range(2):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            o, l = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]], system_prompt_size=sys_n if c == 0 else 0, return_logits=True)
            outs.append(o[0])
            logs.append(l[0][:len(o[0])].copy())
        return outs, logs

    monkeypatch.setenv("ISST_FUSE_ATTN_OPROJ", "0")
    ref = make_engine(cfg, w, debug_taps=False, max_llm_cache_size=300, max_streams=1)
    want = chunks(ref, ref.open_stream())
    ref.close()
    monkeypatch.setenv("ISST_FUSE_ATTN_OPROJ", "1")
    monkeypatch.setenv("ISST_FUSE_AO_TEST_TIMEOUT", "1")
    eng = make_engine(cfg, w, debug_taps=False, max_llm_cache_size=300, max_streams=2)
    capfd.readouterr()
    t0 = time.time()
    got = chunks(eng, eng.open_stream())
    assert time.time() - t0 < 30.0, "the bounded waits took too long"
    err = capfd.readouterr().err
    assert err.count("timed out waiting for its own workgroups") == 1, f"the fallback must be announced exactly once:\n{err}"
    more = chunks(eng, eng.open_stream())  # a second stream on the same handle: three launches from the start, no second announcement
    assert "timed out" not in capfd.readouterr().err
    eng.close()
    assert got[0] == want[0] and more[0] == want[0]
    for c, (x, y, z) in enumerate(zip(got[1], want[1], more[1])):
        assert np.array_equal(x, y) and np.array_equal(z, y), f"chunk {c}: the re-issued pass did not return the three-launch path's bits"


@pytest.mark.parametrize("device_scorer", ["1", "0"])
def test_fused_launch_timeout_inside_a_beam_search_falls_back_too(monkeypatch, capfd, device_scorer):
    """The same starvation while the B beams of one stream run the fused launch (a shared-prefix group of B rows): the beam loop -- with the scorer on the
    device (it leaves its state alone when the error word is up) and with the host scorer -- re-issues the pass on the three launches; tokens, candidate
    lists and every arena's KV equal those of a handle that never fused."""
    cfg = toy_config()
    B = 3
    gen = GenConfig(max_new_tokens=6, beam=B, max_llm_cache_size=100)  # (a cache short enough for the fused launch's 64-slot spans at toy width: llm_attn_oproj_supported)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=74)
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=9)
    sys_n = len(synth.system_prompt_ids(cfg))
    monkeypatch.setenv("ISST_BEAM_DEVICE", device_scorer)

    def run(fuse, starve):
        monkeypatch.setenv("ISST_FUSE_ATTN_OPROJ", fuse)
        monkeypatch.setenv("ISST_FUSE_AO_TEST_TIMEOUT", "1" if starve else "0")
        eng = Engine(cfg, max_streams=1, max_multiplier=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=100, max_system_prompt=64, max_beams=B)
        eng.load_weights(w)
        sid = eng.open_stream()
        outs, traces, kvs = [], [], []
        for c in range(2):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            eng.beam_trace_begin(B)
            ids, _ = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]], system_prompt_size=sys_n if c == 0 else 0)
            traces.append(eng.beam_trace_end())
            outs.append(ids[0])
            n = eng.stream_info(sid)["llm_cache_len"]
            kvs.append([[eng.read_kv(sid, p, layer=1, kv_head=1, beam=b) for p in range(n)] for b in range(B)])
        eng.close()
        return outs, traces, kvs

    want = run("0", False)
    capfd.readouterr()
    got = run("3", True)
    assert capfd.readouterr().err.count("timed out waiting for its own workgroups") == 1
    assert got[0] == want[0]
    for c, (xa, xb) in enumerate(zip(got[1], want[1])):
        assert len(xa) == len(xb)
        for step, ((va, ia, sa), (vb, ib, sb)) in enumerate(zip(xa, xb)):
            assert np.array_equal(va, vb) and np.array_equal(ia, ib) and np.array_equal(sa, sb), f"chunk {c} step {step}: candidates differ after the fallback"
    for c, (ka, kb) in enumerate(zip(got[2], want[2])):
        for b in range(B):
            for p_, ((k1, v1), (k2, v2)) in enumerate(zip(ka[b], kb[b])):
                assert torch.equal(k1, k2) and torch.equal(v1, v2), f"chunk {c} beam {b} position {p_}: KV differs after the fallback"


def test_llm_embed_tap_equals_oracle_splice():
    """The `llm_embed` tap (decoder input rows after embedding lookup + speech splice, model/llm.py:86-113) against the oracle's splice of the
    oracle's own speech features: a pure row copy on the device, so token rows are bit-exact and speech rows carry only the encoder's error."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=2)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=72)
    eng = make_engine(cfg, w)
    sid = eng.open_stream()
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=6)
    cache, rope_e = oenc.new_cache(cfg), oenc.make_rope(cfg)
    for c in range(2):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        feats, cache = oenc.encode_speech(w, cfg, x.unsqueeze(0).bfloat16(), cache, 1, rope_e)
        ids = torch.tensor(prompt)
        ref = ollm.splice_speech(cfg, ids, torch.nn.functional.embedding(ids, w["model.embed_tokens.weight"]), feats[0])
        eng.generate(gen, [sid], [seg], [prompt], [[]])
        got = eng.debug_tap("llm_embed").view(len(prompt), cfg.llm_dim)
        is_speech = torch.tensor([t == cfg.sp_patch_id for t in prompt])
        assert torch.equal(got[~is_speech], ref[~is_speech]), f"chunk {c}: token rows of the decoder input differ"
        assert_close(f"chunk {c} spliced speech rows", got[is_speech], ref[is_speech], 0.06, 0.02)


def test_sample_branch_draws_from_the_oracle_distribution():
    """--do-sample (reference agents/infinisst.py:311-315 -> patch_hf.py:606-624 -> HF _sample): processors on the device, the warpers Temperature -> TopK ->
    TopP -> Epsilon and the draw on the host (csrc/warp.hip).  The engine runs FREE over 4 chunks; the oracle follows its tokens (teacher-forced) and
    supplies, per step, the reference's warped distribution and the uniform of (seed, stream, chunk, step): the engine's token must be the inverse-CDF
    draw of that distribution to within the probability mass bf16 logit noise can move (0.04), and must never be a token the oracle's warpers removed
    with a margin.  Same call again, and the same stream next to another one in a batch: identical tokens (counter-based generator)."""
    cfg = toy_config()
    gen = GenConfig(max_new_tokens=8, max_llm_cache_size=150, do_sample=True, temperature=0.8, top_k=50, top_p=0.9, epsilon_cutoff=0.001)
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=91)
    eng = make_engine(cfg, w, debug_taps=False, max_streams=3)
    sid, sid2, sid3 = eng.open_stream(), eng.open_stream(), eng.open_stream()
    audio = synth.synthetic_audio(cfg.chunk_samples * 4, stream_id=9)
    other = synth.synthetic_audio(cfg.chunk_samples * 4, stream_id=10)
    kv, sc = ollm.new_kv(cfg), oenc.new_cache(cfg)
    rope_l, rope_e = ollm.llm_rope_tables(cfg, 2048, torch.bfloat16), oenc.make_rope(cfg)
    sys_n = len(synth.system_prompt_ids(cfg))
    prev, steps, distinct = [], 0, set()
    for c in range(4):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        prompt = synth.chunk_prompt_ids(cfg, 1, first=(c == 0))
        outs, _ = eng.generate(gen, [sid], [seg], [prompt], [prev[-100:]], system_prompt_size=sys_n if c == 0 else 0)
        again, _ = eng.generate(gen, [sid2, sid3], [seg, other[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]], [prompt, prompt], [prev[-100:], []],
                                system_prompt_size=sys_n if c == 0 else 0, forced_tokens=[outs[0], None])
        assert again[0] == outs[0]
        x = torch.from_numpy(seg)
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref = ogen.generate(w, cfg, gen, prompt, x.unsqueeze(0).bfloat16(), kv, sc, rope_l, rope_e, prev[-100:], forced_tokens=outs[0], stream=sid, chunk=c)
        for s, tok in enumerate(outs[0]):
            warped = ogen.warp_logits(ref.step_scores[s], gen.temperature, gen.top_k, gen.top_p, gen.epsilon_cutoff)
            p = warped.softmax(-1).double()
            cum = torch.cumsum(p, 0)
            u = ogen.sample_uniform(gen.seed, sid, c, s) * float(cum[-1])
            lo, hi = (float(cum[tok - 1]) if tok > 0 else 0.0), float(cum[tok])
            assert lo - 0.04 <= u < hi + 0.04, f"chunk {c} step {s}: token {tok} covers [{lo:.4f}, {hi:.4f}) of the oracle's CDF, the uniform is {u:.4f}"
            steps += 1
            distinct.add(tok)
        prev.extend(outs[0][:-1])
        assert eng.stream_info(sid)["llm_cache_len"] == ollm.kv_len(kv)
    # a fresh stream with the same id, audio and seed draws the same tokens; another seed draws others
    eng.reset_stream(sid2)
    gen_b = GenConfig(max_new_tokens=8, max_llm_cache_size=150, do_sample=True, temperature=0.8, top_k=50, top_p=0.9, epsilon_cutoff=0.001, seed=12345)
    prompt = synth.chunk_prompt_ids(cfg, 1, first=True)
    a1, _ = eng.generate(gen_b, [sid2], [audio[:cfg.chunk_samples]], [prompt], [[]], system_prompt_size=sys_n)
    eng.reset_stream(sid2)
    a2, _ = eng.generate(gen_b, [sid2], [audio[:cfg.chunk_samples]], [prompt], [[]], system_prompt_size=sys_n)
    assert a1 == a2
    print(f"sample branch: {steps} steps, {len(distinct)} distinct tokens drawn; another seed: {a1[0]}")
    assert steps >= 16 and len(distinct) >= 8, "the draws must actually vary (top_k 50 at temperature 0.8 over near-flat toy logits)"
    # beams on an engine created for greedy decoding only (max_beams 1) are refused (beam sample itself: tests/test_gpu_beam.py)
    with pytest.raises(IsstError, match="max_beams"):
        eng.generate(GenConfig(max_new_tokens=4, beam=2, do_sample=True), [sid3], [other[:cfg.chunk_samples]], [prompt], [[]])
