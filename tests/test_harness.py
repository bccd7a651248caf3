"""Host-side harness: tokenizer-driven prompts and the instances.log writer.  CPU only."""
import json

import numpy as np
import pytest

from infinisst_amd import harness as H
from infinisst_amd import synth
from infinisst_amd.config import toy_config
from stub_tokenizer import StubTokenizer


def test_chat_prompt_matches_synthetic_layout():
    """The tokenizer path must produce the chunk layout the engine is benchmarked on (synth.chunk_prompt_ids): later
    chunks = [EOT, user header, S patches, EOT, assistant header] after the 25-token strip; first = system + the same
    without the leading EOT (reference agents/infinisst.py:225-268)."""
    cfg = toy_config()
    tok = StubTokenizer(cfg)
    for m in (1, 2, 4):
        p = H.ChatPrompt(tok, "English", "German", cfg.block_size, llama31=True)
        first = p(True, m)
        later = p(False, m)
        assert later == synth.chunk_prompt_ids(cfg, m, first=False)
        assert later.count(cfg.sp_patch_id) == cfg.block_size // 4 * m
        assert first[p.system_prompt_size:] == later[1:]
        assert p.system_prompt_size == len(tok.apply_chat_template([[p.system_message(m)]])[0])
        assert cfg.sp_patch_id + 2 + m in first[:p.system_prompt_size]  # <latency_m>


def test_chat_prompt_llama3_branch_overwrites_first_token_with_eos():
    cfg = toy_config()
    tok = StubTokenizer(cfg)
    p = H.ChatPrompt(tok, "English", "German", cfg.block_size, llama31=False)
    p(True, 1)
    later = p(False, 1)
    assert later[0] == tok.eos_token_id
    assert later.count(cfg.sp_patch_id) == 12


def test_non_language_ids_scans_the_vocabulary():
    cfg = toy_config()
    assert H.non_language_ids(StubTokenizer(cfg)) == [7]


class FakeStates:
    def __init__(self):
        self.reset()

    def reset(self):
        self.source, self.target = [], []
        self.source_finished = False
        self.source_sample_rate = 0


class FakeAgent:
    """Emits 'w<k>' after every second segment and two words at the end (Read / Write actions as the real agent)."""
    source_segment_size = 960

    def __init__(self):
        self.states = FakeStates()
        self.calls = 0

    def policy(self, states):
        from infinisst_amd.agent import ReadAction, WriteAction
        self.calls += 1
        n = len(states.source) // 15360
        if states.source_finished:
            return WriteAction(content="end fin", finished=True)
        if n % 2 == 0:
            return WriteAction(content=f"w{n}", finished=False)
        return ReadAction()


def test_evaluate_writes_instances_log(tmp_path):
    agent = FakeAgent()
    wav = np.zeros(15360 * 5 + 100, dtype=np.float32)  # 5 full segments + a tail
    ticks = iter(np.arange(0, 1000, 0.010))  # every policy call "costs" 10 ms
    inst = H.evaluate(agent, [("a.wav", wav), ("b.wav", wav[:15360])], references=["x y z w v", "x"], output_dir=str(tmp_path),
                      clock=lambda: float(next(ticks)))
    assert [i.prediction for i in inst] == ["w2 w4 end fin", "end fin"]
    a = inst[0]
    assert a.delays == [1920.0, 3840.0, a.source_length, a.source_length]
    assert all(e > d for e, d in zip(a.elapsed, a.delays)) and a.elapsed == sorted(a.elapsed)
    assert abs(a.source_length - 1000.0 * wav.shape[0] / 16000) < 1e-6
    lines = [json.loads(l) for l in open(tmp_path / "instances.log", encoding="utf-8")]
    assert [l["index"] for l in lines] == [0, 1]
    assert set(lines[0]) == {"index", "prediction", "delays", "elapsed", "prediction_length", "reference", "source", "source_length"}
    assert lines[0]["prediction_length"] == 4 and lines[0]["source"] == ["a.wav"]
    scores = json.load(open(tmp_path / "scores.json"))
    assert scores["instances"] == 2 and scores["LAAL_ms"] > 0 and scores["LAAL_CA_ms"] > scores["LAAL_ms"]


def test_char_units_and_laal_known_answer():
    assert H.split_units("你好 世界", "char") == ["你", "好", "世", "界"]
    assert H.split_units("a  b", "word") == ["a", "b"]
    # wait-1-like policy on a 4000 ms source, 4 reference units: unit i at (i+1) * 1000 ms -> lag 1000 ms each
    assert H.laal([1000.0, 2000.0, 3000.0, 4000.0], 4000.0, 4) == pytest.approx(1000.0)
    # a longer hypothesis is not rewarded: gamma uses max(|Y|, |Y*|)
    assert H.laal([1000.0, 2000.0, 3000.0, 4000.0, 4000.0], 4000.0, 4) == pytest.approx((1000 + 1200 + 1400 + 1600) / 4)
    assert H.laal([], 4000.0, 4) is None


def test_chat_prompt_matches_reference_prepare_inputs(golden_dir):
    """§8 row a2 pinned: tests/golden/prompts.npz holds the ids the REFERENCE's own `_prepare_inputs` (agents/infinisst.py:225-268)
    produced over the stub tokenizer and over a real transformers tokenizer with a Llama-3.1-shaped chat template, first and later
    chunks, multipliers 1..4, llama31 (`[:, 25:]`) and llama3 (`[:, 0] = eos`) branches.  `harness.ChatPrompt` must reproduce them
    bit for bit, and `synth.chunk_prompt_ids` (what the engine is benchmarked on) must equal the later-chunk layout."""
    import os
    import transformers
    from tiny_tokenizer import build_tokenizer_dir
    g = np.load(os.path.join(golden_dir, "prompts.npz"))
    cfg = toy_config()
    import tempfile
    hf = transformers.AutoTokenizer.from_pretrained(build_tokenizer_dir(tempfile.mkdtemp(), cfg), padding_side="right", use_fast=False)
    hf.pad_token = H.PAD_TOKEN
    H.preprocess_tokenizer(hf, 4)
    for tname, tok in (("stub", StubTokenizer(cfg)), ("hf", hf)):
        for llama31 in (1, 0):
            for m in (1, 2, 3, 4):
                key = f"{tname}_l31{llama31}_m{m}"
                p = H.ChatPrompt(tok, "English", "German", cfg.block_size, llama31=bool(llama31))
                first, later = p(True, m), p(False, m)
                assert first == g[key + "_first"].tolist(), key
                assert later == g[key + "_later"].tolist(), key
                assert p.system_prompt_size == int(g[key + "_sys"]), key
                if llama31:
                    assert later == synth.chunk_prompt_ids(cfg, m, first=False), key
                    assert len(first) - p.system_prompt_size == len(later) - 1
