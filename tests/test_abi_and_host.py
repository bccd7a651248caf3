"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares, the package
fails loudly without a GPU, and the host-side logic (chunk padding, eviction policy, rope tables, prompts)
agrees with the oracle."""
import os
import re

import numpy as np
import pytest
import torch

from infinisst_amd import engine as E
from infinisst_amd import rope, synth
from infinisst_amd.agent import InfiniSST, S2TAgentStates, default_args
from infinisst_amd.config import GenConfig, full_config, toy_config
from oracle import agent as oag
from oracle import llm as ollm
from oracle import speech_encoder as oenc

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    hdr = open(os.path.join(ROOT, "include", "infinisst_hip.h")).read()
    declared = set(re.findall(r"\b(isst_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(E.EXPORTS)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported"


def test_c_config_struct_matches_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "infinisst_hip.h")).read()
    start = hdr.index("typedef struct isst_config {") + len("typedef struct isst_config {")
    body = hdr[start:hdr.index("} isst_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl or decl.startswith("typedef"):
            continue
        decl = re.sub(r"^(int|float)\s+", "", decl)
        names += [re.sub(r"\[.*\]", "", n).strip() for n in decl.split(",")]
    assert names == [f for f, _ in E._Config._fields_]


@pytest.mark.skipif(torch.cuda.is_available(), reason="needs a box without GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(E.IsstError, match="no GPU"):
        E.Engine(toy_config())


def test_full_config_geometry():
    c = full_config()
    assert (c.chunk_samples, c.first_chunk_offset, c.samples_per_frame, c.receptive_field, c.shrink_factor) == (15360, 399, 320, 400, 4)
    assert oenc.feat_extract_output_lengths(c, 399 + 15360) == 12
    assert len(synth.chunk_prompt_ids(c)) == 22
    n_params = sum(int(np.prod(s)) for s in synth.weight_shapes(c).values())
    assert 8.3e9 < n_params < 8.4e9


def test_rope_tables_match_oracle():
    for mode in ("bf16", "fp32"):
        cfg = toy_config().replace(enc_rope_mode=mode)
        c, s = rope.encoder_tables(cfg, 700)
        oc, os_, _ = oenc.make_rope(cfg, 700)
        assert torch.equal(c, oc) and torch.equal(s, os_)
    cfg = full_config()
    c, s = rope.llm_tables(cfg, 1500)
    oc, os_ = ollm.llm_rope_tables(cfg, 1500, torch.bfloat16)
    assert torch.equal(c, oc[:, :64]) and torch.equal(s, os_[:, :64])
    assert torch.equal(oc[:, :64], oc[:, 64:])  # halves are duplicates, so storing one is lossless


class _FakeEngine:
    """Records what the agent asks of the library; emulates cache growth so the eviction policy can be traced."""

    def __init__(self, rng):
        self.rng, self.len, self.sys, self.calls, self.evictions = rng, 0, 0, [], []

    def open_stream(self):
        return 0

    def reset_stream(self, sid):
        self.len = self.sys = 0

    def generate(self, gen, sids, pcm, prompts, prevs, system_prompt_size=0, **kw):
        n_gen = int(self.rng.integers(2, gen.max_new_tokens + 1))
        self.calls.append(dict(n_samples=len(pcm[0]), pcm=pcm[0].copy(), prompt_len=len(prompts[0]), prev=list(prevs[0]), pin=system_prompt_size, n_gen=n_gen))
        if self.len == 0:
            self.sys = system_prompt_size
        self.len += len(prompts[0]) + n_gen - 1
        return [[int(x) for x in self.rng.integers(10, 900, size=n_gen)]], None

    def stream_info(self, sid):
        return {"llm_cache_len": self.len}

    def kv_evict(self, sid, new_size, keep):
        self.evictions.append((new_size, keep))
        self.len = new_size + keep


@pytest.mark.parametrize("keep_sys,max_cache", [(True, 150), (False, 150), (True, 90)])
def test_agent_host_logic_matches_oracle(keep_sys, max_cache):
    """Chunk padding, encoder_input_ids window, checkpoint list and eviction sizes of the product agent == oracle
    (which is pinned to the reference's policy() by tests/golden/agent.npz)."""
    cfg = toy_config().replace(block_size=48)
    rng = np.random.default_rng(7)
    eng = _FakeEngine(rng)
    args = default_args(max_llm_cache_size=max_cache, always_cache_system_prompt=keep_sys, max_new_tokens=10)
    agent = InfiniSST(args, engine=eng, model_cfg=cfg)
    st = agent.build_states()
    st.source_sample_rate = 16000
    st_o = oag.States(source_sample_rate=16000)
    seg_lens = [15360, 15360, 15361, 15359, 30720, 1, 15360, 7000, 15360, 15360, 15360, 15360]
    audio = (0.1 * rng.standard_normal(sum(seg_lens))).astype(np.float32)
    ckpts, pos, cur, first = [], 0, 0, True
    for c, n in enumerate(seg_lens):
        seg = audio[pos:pos + n].tolist()
        pos += n
        st.source.extend(seg)
        st_o.source.extend(seg)
        st.source_finished = c == len(seg_lens) - 1
        agent.policy(st)
        call = eng.calls[-1]
        ref_speech = oag.prepare_speech(cfg, st_o, torch.float32)[0].numpy()
        if first:
            ref_speech = ref_speech[cfg.first_chunk_offset:]  # the library owns the 399-sample history
        assert np.array_equal(call["pcm"], ref_speech), f"chunk {c}"
        assert call["pin"] == (agent.system_prompt_size if (first and keep_sys) else 0)
        cur += call["prompt_len"] + call["n_gen"] - 1
        ckpts.append(cur)
        ev = oag.evict(ckpts, cur, max_cache, keep_sys, agent.system_prompt_size)
        if ev is not None:
            ckpts, new_size = ev
            assert eng.evictions[-1] == (new_size, agent.system_prompt_size if keep_sys else 0)
            cur = new_size + (agent.system_prompt_size if keep_sys else 0)
        assert agent.cache_checkpoints == ckpts
        assert eng.len == cur
        assert call["prev"] == st.target_ids[: len(st.target_ids) - (call["n_gen"] - 1)][-100:]
        first = False


def test_agent_flags_match_reference_names():
    """Flag names/defaults of agents/options.py + agents/infinisst.py:185-198 (the config contract)."""
    import argparse
    p = argparse.ArgumentParser()
    InfiniSST.add_args(p)
    a = p.parse_args([])
    expect = dict(block_size=12, max_cache_size=125, xpos=1, rope=1, beam=1, no_repeat_ngram_lookback=100, no_repeat_ngram_size=3,
                  repetition_penalty=1.2, max_new_tokens=1000, min_start_sec=0.32, latency_multiplier=4, max_latency_multiplier=4,
                  max_llm_cache_size=10000, always_cache_system_prompt=False, pseudo_batch_size=1, source_lang="English",
                  target_lang="German", max_len_a=5, max_len_b=20, top_p=1.0, top_k=0, temperature=1.0)
    for k, v in expect.items():
        assert getattr(a, k) == v, k


def test_policy_gating_without_compute():
    cfg = toy_config()
    eng = _FakeEngine(np.random.default_rng(1))
    agent = InfiniSST(default_args(min_start_sec=0.32), engine=eng, model_cfg=cfg)
    st = agent.build_states()
    assert type(agent.policy(st)).__name__ == "ReadAction"  # sample rate unknown, nothing read
    st.source_sample_rate = 16000
    st.source = [0.0] * 1000
    assert type(agent.policy(st)).__name__ == "ReadAction"  # < min_start_sec
    st.source_finished = True
    act = agent.policy(st)
    assert type(act).__name__ == "WriteAction" and act.content == "" and act.finished  # < 0.32 s total
    assert eng.calls == []


def test_checkpoint_split_matches_reference_key_layout(tmp_path):
    """A synthetic `pytorch_model.bin` with the reference's key names plus tensors real checkpoints carry but the hot path
    never reads (pos_conv, rotary freqs, pre-training heads): hot-path tensors come back as bf16, the rest is skipped,
    a missing or mis-shaped tensor is an error (strict load, reference agents/infinisst.py:179-180)."""
    from infinisst_amd import checkpoint
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.float32, seed=5)
    state = dict(w)
    enc = "model.speech_encoder.speech_encoder."
    state[enc + "encoder.pos_conv.0.weight_g"] = torch.zeros(1, 1, 8)
    state[enc + "mask_emb"] = torch.zeros(cfg.enc_dim)
    freqs = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)) * 1.01
    for i in range(cfg.enc_layers):
        state[f"{enc}encoder.layers.{i}.self_attn.rotary_emb.freqs"] = freqs
    path = tmp_path / "pytorch_model.bin"
    torch.save(state, path)
    out, inv, skipped = checkpoint.load_checkpoint(cfg, str(path))
    assert set(out) == set(w) and all(t.dtype == torch.bfloat16 for t in out.values())
    assert torch.equal(inv, freqs) and len(skipped) == 2 + cfg.enc_layers
    c, s = rope.encoder_tables(cfg, 16, inv)
    c0, _ = rope.encoder_tables(cfg, 16)
    assert not torch.equal(c, c0)
    bad = dict(state)
    del bad["lm_head.weight"]
    with pytest.raises(KeyError, match="lm_head"):
        checkpoint.split_state_dict(cfg, bad)
    bad = dict(state)
    bad["model.norm.weight"] = torch.zeros(3)
    with pytest.raises(ValueError, match="model.norm.weight"):
        checkpoint.split_state_dict(cfg, bad)


def test_checkpoint_key_layouts_and_geometry_inference(tmp_path):
    """The three layouts a checkpoint can arrive in (pruned `pytorch_model.bin`; the flat dict train/prune_bin.py consumes, every key
    prefixed `model.`; a Lightning file with a `state_dict` wrapper) load to the same tensors, and the model geometry is
    recovered from the shapes alone (the reference reads it from `--model-name` / `--w2v2-path`, agents/infinisst.py:150-171)."""
    from infinisst_amd import checkpoint
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.float32, seed=5)
    pruned, flat, wrapped = tmp_path / "a.bin", tmp_path / "b.bin", tmp_path / "c.bin"
    torch.save(w, pruned)
    torch.save({"model." + k: v for k, v in w.items()}, flat)
    torch.save({"state_dict": {"model." + k: v for k, v in w.items()}, "epoch": 3}, wrapped)
    ref, _, _ = checkpoint.load_checkpoint(cfg, str(pruned))
    for path in (flat, wrapped):
        out, _, skipped = checkpoint.load_checkpoint(cfg, str(path))
        assert set(out) == set(ref) and not skipped
        assert all(torch.equal(out[k], ref[k]) for k in ref)
    got = checkpoint.infer_config(checkpoint.load_state_dict_file(str(flat)), block_size=cfg.block_size, max_cache_size=cfg.max_cache_size)
    for f in ("conv_layers", "conv_bias", "enc_dim", "enc_layers", "enc_heads", "enc_ffn", "shrink_layers", "llm_dim", "llm_layers",
              "llm_heads", "llm_kv_heads", "llm_ffn", "vocab"):
        assert getattr(got, f) == getattr(cfg, f) or list(map(tuple, getattr(got, f))) == list(map(tuple, getattr(cfg, f))), f
    full = full_config()
    shapes = {k: torch.empty(s, device="meta") for k, s in synth.weight_shapes(full).items()}
    inferred = checkpoint.infer_config(shapes)
    assert synth.weight_shapes(inferred) == synth.weight_shapes(full)


def test_effective_new_cache_size_covers_stale_checkpoints():
    """`cache_checkpoints` survives utterances (reference agents/infinisst.py:106): the size handed to the library must stay inside
    the evictable range for every integer the checkpoint loop can produce, following Python's slice semantics where those are sane."""
    from infinisst_amd.agent import effective_new_cache_size as f
    assert f(500, 1010, 66) == 500                 # ordinary
    assert f(944, 1010, 66) == 944 and f(945, 1010, 66) == 944 and f(5000, 1010, 66) == 944   # would overlap the pinned prefix
    assert f(-190, 1010, 66) == 820                # k[:, :, 190:]
    assert f(-2000, 1010, 66) == 0                 # slice past the end: empty tail
    assert f(0, 1010, 66) == 0 and f(3, 2, 5) == 0


class _RecordingEngine(_FakeEngine):
    instances = []

    def __init__(self, cfg, **kw):
        super().__init__(np.random.default_rng(3))
        self.cfg, self.kw, self.loaded = cfg, kw, None
        _RecordingEngine.instances.append(self)

    def load_weights(self, weights, enc_inv_freq=None):
        self.loaded = (dict(weights), enc_inv_freq)


def test_agent_constructs_from_args_alone(tmp_path, monkeypatch):
    """`InfiniSST(args)` -- the only way SimulEval builds an agent -- runs load_model(args): real transformers tokenizer from
    `--model-name` (+ the speech / latency tokens), `--suppress-non-language` scan, geometry + weights from `--state-dict-path`,
    engine sized from the flags, chat-template prompts (reference agents/infinisst.py:69-113,130-183).  The engine class is
    replaced by a recorder here (no GPU); tests/test_gpu_agent_args.py runs the same construction on the real library."""
    import argparse
    import infinisst_amd.agent as A
    from tiny_tokenizer import build_tokenizer_dir
    cfg = toy_config()
    model_dir = build_tokenizer_dir(tmp_path, cfg)
    w = synth.random_weights(cfg, dtype=torch.float32, seed=9)
    freqs = 1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64)) * 1.02
    state = {"model." + k: v for k, v in w.items()}  # un-pruned flat layout
    for i in range(cfg.enc_layers):
        state[f"model.model.speech_encoder.speech_encoder.encoder.layers.{i}.self_attn.rotary_emb.freqs"] = freqs
    state["model.model.speech_encoder.speech_encoder.encoder.pos_conv.0.bias"] = torch.zeros(cfg.enc_dim)
    ckpt = tmp_path / "pytorch_model.bin"
    torch.save(state, ckpt)
    monkeypatch.setattr(A, "Engine", _RecordingEngine)
    _RecordingEngine.instances.clear()
    parser = argparse.ArgumentParser()
    A.InfiniSST.add_args(parser)
    args = parser.parse_args(["--model-name", model_dir, "--state-dict-path", str(ckpt), "--w2v2-type", "w2v2", "--w2v2-path", "unused.pt",
                              "--length-shrink-cfg", "[(128,2,2)] * 2", "--block-size", "48", "--max-cache-size", "576", "--xpos", "0",
                              "--max-llm-cache-size", "150", "--always-cache-system-prompt", "--max-new-tokens", "10", "--beam", "4",
                              "--no-repeat-ngram-size", "5", "--latency-multiplier", "1", "--min-start-sec", "0", "--suppress-non-language"])
    agent = A.InfiniSST(args)
    eng = _RecordingEngine.instances[-1]
    assert agent.engine is eng and eng.loaded is not None
    assert set(eng.loaded[0]) == set(w) and torch.equal(eng.loaded[1], freqs)
    assert all(t.dtype == torch.bfloat16 for t in eng.loaded[0].values())
    got = eng.cfg
    assert (got.llm_dim, got.llm_layers, got.llm_heads, got.llm_kv_heads, got.vocab) == (cfg.llm_dim, cfg.llm_layers, cfg.llm_heads, cfg.llm_kv_heads, cfg.vocab)
    assert (got.sp_patch_id, got.user_id, got.assistant_id, got.start_header_id) == (cfg.sp_patch_id, cfg.user_id, cfg.assistant_id, cfg.start_header_id)
    assert got.eos_ids == cfg.eos_ids and got.block_size == 48 and got.max_cache_size == 576 and got.rope_theta == cfg.rope_theta
    assert eng.kw["max_beams"] == 4 and eng.kw["max_llm_cache_size"] == 150 and eng.kw["max_multiplier"] == 4
    assert agent.bad_words_ids == [7] and agent.llama31
    first, later = agent.prompt_fn(True, 1), agent.prompt_fn(False, 1)
    assert later == synth.chunk_prompt_ids(cfg, 1, first=False)
    assert eng.kw["max_system_prompt"] == agent.system_prompt_size == len(first) - len(later) + 1
    assert eng.kw["max_prompt_len"] >= len(agent.prompt_fn(True, 4))
    # one utterance through policy(): READ/WRITE actions, the pinned prefix, detokenised output through the real tokenizer
    acts, st = A.feed_segments(agent, synth.synthetic_audio(cfg.chunk_samples * 3), cfg.chunk_samples)
    assert eng.calls[0]["pin"] == agent.system_prompt_size and eng.calls[0]["prompt_len"] == len(first)
    assert type(acts[-1]).__name__ == "WriteAction" and acts[-1].finished
    assert all(isinstance(a.content, str) for a in acts if hasattr(a, "content"))
    assert got.enc_rope
    with pytest.raises(NotImplementedError):
        A.InfiniSST(parser.parse_args(["--model-name", model_dir, "--state-dict-path", str(ckpt), "--xpos", "1", "--block-size", "48"]))
    # --rope 0 runs (absolute sinusoid positions), with either --xpos: the rotary module is built and never called (patch_speech_encoder.py:823)
    A.InfiniSST(parser.parse_args(["--model-name", model_dir, "--state-dict-path", str(ckpt), "--rope", "0", "--block-size", "48"]))
    assert not _RecordingEngine.instances[-1].cfg.enc_rope
    with pytest.raises(ValueError, match="length-shrink-cfg"):
        bad = parser.parse_args(["--model-name", model_dir, "--state-dict-path", str(ckpt), "--xpos", "0", "--block-size", "48",
                                 "--length-shrink-cfg", "[(128,2,2)] * 3"])
        A.InfiniSST(bad)


def test_marshalling_helpers_keep_every_sequence():
    """engine._pack_int32 (one buffer per argument, per-stream pointers into it) and streams._Slot's int32 window of target ids."""
    import ctypes as C
    from infinisst_amd.engine import _pack_int32
    from infinisst_amd.streams import _Slot
    rng = np.random.default_rng(3)
    for as_arrays in (False, True):
        seqs = [None, [], [5], list(range(7)), None, [int(x) for x in rng.integers(0, 1 << 20, size=100)]]
        given = [None if q is None else (np.asarray(q, dtype=np.int32) if as_arrays else q) for q in seqs]
        keep, ptrs, lens = _pack_int32(given, len(seqs))
        for q, p_, l in zip(seqs, ptrs, lens):
            assert l == (0 if q is None else len(q))
            if l:
                assert list(np.ctypeslib.as_array(C.cast(p_, C.POINTER(C.c_int)), shape=(l,))) == q
            else:
                assert not p_
    keep, ptrs, lens = _pack_int32([None, []], 2)
    assert keep is None and not ptrs[0] and not ptrs[1] and list(lens) == [0, 0]
    slot, ref, look = _Slot(sid=0), [], 10
    assert slot.window(look) is None
    for step in range(300):
        new = [int(x) for x in rng.integers(0, 1000, size=int(rng.integers(0, 9)))]
        ref.extend(new)
        slot.push_targets(new, look)
        w = slot.window(look)
        assert ([] if w is None else w.tolist()) == ref[-look:]
    slot.push_targets(list(range(500)), look)  # one push longer than the whole buffer
    assert slot.window(look).tolist() == list(range(490, 500))


def test_position_table_rows_are_the_oracle_sinusoid():
    """rope.encoder_position_table (what --rope 0 hands the library): row lookup by the bf16 rounding of the position reproduces
    oracle.sinusoidal_positional_embedding at any offset, and the identity rotary tables leave q / k alone."""
    from infinisst_amd import rope
    from oracle import speech_encoder as oenc
    cfg = toy_config().replace(enc_rope=False)
    table = rope.encoder_position_table(cfg)
    assert table.shape == (rope.ENC_POS_ROWS, cfg.enc_dim) and table.dtype == torch.bfloat16
    vals = rope.encoder_position_values()
    assert vals[255] == 255 and vals[256] == 256 and vals[257] == 258 and vals[-1] == 2 ** 24
    assert torch.equal(vals, vals.bfloat16().float()) and bool((vals[1:] > vals[:-1]).all())

    def row_of(p):  # the kernel's lookup (rowops.hip enc_add_position_kernel)
        if p < 256:
            return p
        bits = torch.tensor([float(p)]).bfloat16().view(torch.int16).item() & 0xFFFF
        return 256 + bits - 0x4380
    for off in (0, 250, 1000, 22491, 65000, 2 ** 24 - 48):
        want = oenc.sinusoidal_positional_embedding(off, 48, cfg.enc_dim)
        got = torch.stack([table[row_of(off + t)] for t in range(48)])
        assert torch.equal(got, want), off
    cos, sin = rope.encoder_tables(cfg, 64)
    assert bool((cos == 1).all()) and bool((sin == 0).all())


def test_policy_closes_the_instance_when_the_last_step_brings_no_audio():
    """The evaluator's final step may carry no new samples (length an exact multiple of the segment size): a ReadAction then never
    finishes the instance; the agent answers WriteAction('', finished=True) without calling the library."""
    cfg = toy_config()
    eng = _FakeEngine(np.random.default_rng(1))
    agent = InfiniSST(default_args(), engine=eng, model_cfg=cfg)
    st = agent.build_states()
    st.source_sample_rate = 16000
    st.source.extend([0.01] * cfg.chunk_samples)
    agent.policy(st)
    n_calls = len(eng.calls)
    assert type(agent.policy(st)).__name__ == "ReadAction" and len(eng.calls) == n_calls   # nothing new, source still open
    st.source_finished = True
    act = agent.policy(st)
    assert type(act).__name__ == "WriteAction" and act.content == "" and act.finished and len(eng.calls) == n_calls


def test_splice_row_map_matches_reference_fixture(golden_dir):
    """The library's host half of the speech splice (engine_llm.hip splice_rows, exported as isst_op_splice_map) against
    tests/golden/splice.npz = the reference's SpeechLlamaModel.forward (model/llm.py:86-113): system + user turn, later-chunk layout,
    surplus features; plus the shortfall case (fewer features than patch slots: the reference's slices shorten the sequence), checked
    against the oracle's literal torch.cat restatement."""
    g = np.load(os.path.join(golden_dir, "splice.npz"))
    user, assist, sh, _ = (int(x) for x in g["ids_cfg"])
    for case in range(3):
        ids, feats, table = g[f"ids_{case}"], torch.from_numpy(g[f"feats_{case}"]), torch.from_numpy(g[f"table_{case}"])
        m = E.op_splice_map(ids, user, assist, sh, feats.shape[0])
        out = torch.stack([table[ids[t]] if t >= 0 else feats[-1 - t] for t in m])
        assert torch.equal(out, torch.from_numpy(g[f"embeds_{case}"])), case
    cfg = toy_config().replace(user_id=user, assistant_id=assist, start_header_id=sh)
    ids, table = g["ids_1"], torch.from_numpy(g["table_1"])  # 24 patch slots
    for n_feat in (0, 5, 12, 23, 24, 30):
        feats = torch.from_numpy(g["feats_1"])[:n_feat] if n_feat <= 24 else torch.cat([torch.from_numpy(g["feats_1"])] * 2)[:n_feat]
        m = E.op_splice_map(ids, user, assist, sh, n_feat)
        ref = ollm.splice_speech(cfg, torch.from_numpy(ids), table[ids], feats)
        assert len(m) == ref.shape[0] == len(ids) - max(0, 24 - n_feat)
        assert torch.equal(torch.stack([table[ids[t]] if t >= 0 else feats[-1 - t] for t in m]), ref), n_feat


def test_two_utterances_with_stale_checkpoints_never_ask_for_an_impossible_eviction():
    """`cache_checkpoints` is agent-level and survives `states.reset()` (reference agents/infinisst.py:106): the second utterance starts with
    the first one's checkpoints still in the list, so the checkpoint loop can return sizes outside [0, cur - pinned].  The library refuses
    those (the reference's slices would silently duplicate or drop entries); the agent must hand over a size inside the evictable range."""
    cfg = toy_config()

    class Strict(_FakeEngine):
        def kv_evict(self, sid, new_size, keep):
            assert 0 <= new_size <= self.len - keep, (new_size, self.len, keep)
            super().kv_evict(sid, new_size, keep)

    eng = Strict(np.random.default_rng(11))
    agent = InfiniSST(default_args(max_llm_cache_size=120, always_cache_system_prompt=True, max_new_tokens=10), engine=eng, model_cfg=cfg)
    st = agent.states
    for utt, n_seg in enumerate((9, 14, 6)):
        st.reset()
        st.source_sample_rate = 16000
        for c in range(n_seg):
            st.source.extend([0.01] * cfg.chunk_samples)
            st.source_finished = c == n_seg - 1
            agent.policy(st)
        assert eng.len <= 120 + agent.system_prompt_size + 40
    assert len(eng.evictions) >= 6


def test_dpo_sampling_writes_the_reference_line_format(tmp_path):
    """`--dpo-sampling` (reference agents/infinisst.py:369-382): every chunk's translation is collected as a quoted string ('' when empty) and the
    utterance's list is appended to --output-file as one bracketed line when the source finishes."""
    cfg = toy_config()
    out = tmp_path / "translations.json"
    eng = _FakeEngine(np.random.default_rng(2))
    agent = InfiniSST(default_args(dpo_sampling=True, output_file=str(out)), engine=eng, model_cfg=cfg,
                      decode_fn=lambda ids: "" if len(ids) % 2 else "w" + str(len(ids)))
    st = agent.states
    expect = []
    for utt in range(2):
        st.reset()
        st.source_sample_rate = 16000
        chunks = []
        for c in range(3):
            st.source.extend([0.01] * cfg.chunk_samples)
            st.source_finished = c == 2
            agent.policy(st)
            n = eng.calls[-1]["n_gen"] - 1
            chunks.append("''" if n % 2 else f"'w{n}'")
        expect.append(f"[{', '.join(chunks)}]")
        assert st.translations_list == []
    assert out.read_text(encoding="utf-8").splitlines() == expect


class _FakeMultiEngine:
    """Many-stream stand-in for the library: cache growth per stream, generated ids a pure function of (stream key, chunk index)."""

    def __init__(self, key_offset=0):
        self.key_offset, self.state, self.calls, self.evicted = key_offset, {}, [], []
        self.last_call_seconds = 0.0

    def open_stream(self):
        sid = len(self.state)
        self.state[sid] = dict(len=0, sys=0, chunk=0)
        return sid

    def reset_stream(self, sid):
        self.state[sid].update(len=0, sys=0)

    def close_stream(self, sid):
        self.state[sid] = None

    def generate(self, gen, sids, pcm, prompts, prevs, system_prompt_size=0, forced_tokens=None, return_logits=False):
        assert len({len(x) for x in pcm}) == 1, "one isst_generate call takes equally long segments"
        outs = []
        for sid, a, pr, pv in zip(sids, pcm, prompts, prevs):
            st = self.state[sid]
            rng = np.random.default_rng(1000 * (sid + self.key_offset) + st["chunk"])
            n_gen = int(rng.integers(2, gen.max_new_tokens + 1))
            if st["len"] == 0:
                st["sys"] = system_prompt_size
            st["len"] += len(pr) + n_gen - 1
            st["chunk"] += 1
            outs.append([int(x) for x in rng.integers(10, 900, size=n_gen)])
            self.calls.append(dict(key=sid + self.key_offset, n=len(sids), prompt_len=len(pr), prev=[] if pv is None else [int(x) for x in pv], samples=len(a)))
        return outs, None

    def stream_info(self, sid):
        return {"llm_cache_len": self.state[sid]["len"]}

    def stream_cache_lens(self, sids):
        return [self.state[s]["len"] for s in sids]

    def kv_evict(self, sid, new_size, keep):
        assert keep == self.state[sid]["sys"] and 0 <= new_size <= self.state[sid]["len"] - keep
        self.evicted.append((sid + self.key_offset, new_size, keep))
        self.state[sid]["len"] = new_size + keep


@pytest.mark.parametrize("keep_sys", [True, False])
def test_stream_batch_equals_one_agent_per_stream(keep_sys):
    """streams.StreamBatch (one library call per tick for every stream that brought a chunk; ragged first / later prompts; streams
    without a new chunk skip the tick; per-stream target ids, checkpoints and eviction) against N independent single-stream agents
    (agent.InfiniSST.policy = the reference's policy(), pinned by agent.npz): same prompts, same encoder_input_ids windows, same
    evictions, same output ids for every stream."""
    from infinisst_amd import synth
    from infinisst_amd.streams import StreamBatch
    cfg = toy_config().replace(block_size=48)
    N, ticks = 5, 30
    args = default_args(max_llm_cache_size=120, always_cache_system_prompt=keep_sys, max_new_tokens=10)
    sys_n = len(synth.system_prompt_ids(cfg, 1))
    eng = _FakeMultiEngine()
    gen = GenConfig(max_new_tokens=10, max_llm_cache_size=120, always_cache_system_prompt=keep_sys)
    batch = StreamBatch(eng, gen, sys_n, lambda first, m: synth.chunk_prompt_ids(cfg, m, first))
    idx = [batch.open() for _ in range(N)]
    agents, states, engines = [], [], []
    for k in range(N):
        e = _FakeMultiEngine(key_offset=k)
        a = InfiniSST(args, engine=e, model_cfg=cfg)
        st = a.states  # the constructor built the agent's states (= opened library stream 0)
        st.source_sample_rate = 16000
        agents.append(a), states.append(st), engines.append(e)
    rng = np.random.default_rng(5)
    seg = cfg.chunk_samples
    for t in range(ticks):
        present = [bool(rng.random() < 0.7) or t == 0 for _ in range(N)]
        present[2] = present[2] and t >= 4  # stream 2 joins late: its FIRST chunk (system prompt) shares a call with later chunks
        audio = [(0.1 * rng.standard_normal(seg)).astype(np.float32) if p else None for p in present]
        outs = batch.step(audio)
        for k in range(N):
            if not present[k]:
                assert outs[k] is None
                continue
            states[k].source.extend(audio[k].tolist())
            before = len(states[k].target_ids)
            agents[k].policy(states[k])
            assert outs[k] == states[k].target_ids[before:], f"tick {t} stream {k}"
            assert batch.slots[idx[k]].ckpts == agents[k].cache_checkpoints, f"tick {t} stream {k}: checkpoints"
            assert eng.state[batch.stream_id(idx[k])]["len"] == engines[k].state[0]["len"]
    assert sorted(eng.evicted) == sorted(ev for e in engines for ev in e.evicted) and len(eng.evicted) > N
    per_key = lambda calls, key: [(c["prompt_len"], c["prev"]) for c in calls if c["key"] == key]
    for k in range(N):
        assert per_key(eng.calls, k) == per_key(engines[k].calls, k)
    assert any(c["n"] > 1 for c in eng.calls) and {c["prompt_len"] for c in eng.calls if c["n"] > 1} >= {len(synth.chunk_prompt_ids(cfg, 1, True)), len(synth.chunk_prompt_ids(cfg, 1, False))}
    # a new utterance on one slot: caches reset, checkpoints stay (reference agents/infinisst.py:60-67,106)
    ck = list(batch.slots[idx[0]].ckpts)
    batch.new_utterance(idx[0])
    assert batch.cache_len(idx[0]) == 0 and batch.slots[idx[0]].ckpts == ck and batch.slots[idx[0]].target_ids == []
    batch.close(idx[1])
    assert batch.open() == idx[1]


def test_host_warpers_and_draw_match_the_oracle(golden_dir):
    """csrc/warp.hip (the sample branch's host half, callable without a GPU): on the fixture's scores and on 200 random rows the same tokens survive as in
    oracle.generate.warp_logits (pinned to transformers' warpers), the surviving scores agree to an fp32 ulp of the temperature division, the draw is the
    oracle's for the same uniform, and the uniforms are the oracle's generator bit for bit."""
    from infinisst_amd import engine as E
    from oracle import generate as ogen
    g = np.load(os.path.join(golden_dir, "sampling_warpers.npz"))
    rng = np.random.default_rng(11)
    rows = [(g[f"c{ci}_scores"], tuple(float(x) for x in g[f"c{ci}_cfg"])) for ci in range(int(g["n_cases"]))]
    for _ in range(200):
        sc = (rng.standard_normal(700) * rng.choice([1.0, 3.0, 6.0])).astype(np.float32)
        sc[rng.integers(0, 700, size=5)] = -np.inf
        rows.append((sc, (float(rng.choice([1.0, 0.7, 1.5])), float(rng.choice([0, 1, 20, 700])), float(rng.choice([1.0, 0.9, 0.5])), float(rng.choice([0.0, 0.003])))))
    n_draws = 0
    for sc, (temp, top_k, top_p, eps) in rows:
        want = ogen.warp_logits(torch.from_numpy(sc), temp, int(top_k), top_p, eps)
        u = float(rng.random())
        got, tok = E.op_warp_sample(sc, temp, int(top_k), top_p, eps, u)
        assert np.array_equal(np.isinf(got), np.isinf(want.numpy())), "kept sets differ"
        fin = ~np.isinf(got)
        np.testing.assert_allclose(got[fin], want.numpy()[fin], rtol=2e-7, atol=0)
        p = want.softmax(-1).double()
        cum = torch.cumsum(p, 0)
        if float((cum - u * float(cum[-1])).abs().min()) > 1e-6:  # (a uniform within 1e-6 of a CDF step may fall either way in fp32)
            assert tok == ogen.draw(want, u)
            n_draws += 1
    assert n_draws > 150
    for key in [(998244353, 0, 0, 0), (998244353, 63, 1874, 39), (7, 5, 3, 1)]:
        assert E.op_sample_uniform(*key) == ogen.sample_uniform(*key)
