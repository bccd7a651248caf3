"""SURVEY section 8(f) row 3 on the GPU: the agent driven through the tokenizer-built prompts and the SimulEval-shaped
evaluation loop; the token stream must equal the oracle agent's on the same prompts."""
import json

import pytest
import torch

from infinisst_amd import harness as H
from infinisst_amd import synth
from infinisst_amd.agent import InfiniSST, default_args
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine
from oracle import agent as oag
from stub_tokenizer import StubTokenizer

pytestmark = pytest.mark.gpu


def test_tokenizer_prompts_and_instances_log(tmp_path):
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=77, recipe="peaked")  # decisive greedy steps: synth.apply_recipe
    tok = StubTokenizer(cfg)
    args = default_args(max_llm_cache_size=150, max_new_tokens=6, max_latency_multiplier=1)
    eng = Engine(cfg, max_streams=1, max_multiplier=1, max_prompt_len=128, max_new_tokens=16, max_llm_cache_size=150, max_system_prompt=80)
    eng.load_weights(w)
    agent = InfiniSST(args, engine=eng, model_cfg=cfg)
    prompt = H.attach_tokenizer(agent, tok, llama31=True, suppress_non_language=True)
    assert agent.bad_words_ids == [7]
    utts = [("u0.wav", synth.synthetic_audio(cfg.chunk_samples * 4 + 2500, stream_id=3)),
            ("u1.wav", synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=4))]
    inst = H.evaluate(agent, utts, references=["a b c", "d"], output_dir=str(tmp_path))
    lines = [json.loads(l) for l in open(tmp_path / "instances.log", encoding="utf-8")]
    assert len(lines) == 2 and lines[1]["source"] == ["u1.wav"]
    for i, l in zip(inst, lines):
        assert len(l["delays"]) == len(l["elapsed"]) == l["prediction_length"] == len(i.units)
        assert l["delays"] == sorted(l["delays"]) and all(d <= l["source_length"] + 1e-6 for d in l["delays"])
    # the pinned system prompt is the tokenizer's, and the second utterance started from a fresh stream
    assert agent.system_prompt_size == prompt.system_prompt_size == len(tok.apply_chat_template([[prompt.system_message(1)]])[0])
    info = eng.stream_info(agent.states.stream_id)
    assert info["enc_n_steps"] == 48 * 2

    # same utterance through the CPU oracle agent with the same prompts: same ids, same READ/WRITE pattern
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=150, suppress_tokens=(7,))
    ref_prompt = H.ChatPrompt(tok, "English", "German", cfg.block_size, True)
    oa = oag.OracleAgent(w, cfg, gen, lambda first: ref_prompt(first, 1), system_prompt_size=prompt.system_prompt_size)
    st = oa.build_states()
    st.source_sample_rate = 16000
    wav = utts[1][1]
    margins = []
    for pos in range(0, wav.shape[0], cfg.chunk_samples):
        st.source.extend(wav[pos:pos + cfg.chunk_samples].tolist())
        st.source_finished = pos + cfg.chunk_samples >= wav.shape[0]
        oa.policy(st)
        for sc in oa.last_output.step_scores[:-1]:  # the steps whose tokens entered target_ids
            top2 = torch.topk(sc, 2).values
            margins.append(float(top2[0] - top2[1]))
    got, ref = list(agent.states.target_ids), list(st.target_ids)
    print("harness ids:", got, "oracle ids:", ref)
    # random toy weights leave near-ties: every id before the oracle's first near-tie (top-2 margin within 2 x the logit tolerance) must agree
    first_tie = next((i for i, m in enumerate(margins) if m <= 0.3), len(margins))
    k = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    assert first_tie >= min(8, len(ref)), f"the peaked recipe must give decisive steps (first near-tie at {first_tie} of {len(ref)})"
    assert k >= min(first_tie, len(ref)), f"ids part at {k}, before the first near-tie at {first_tie}"
    if got == ref:
        from oracle import llm as ollm
        assert eng.stream_info(agent.states.stream_id)["llm_cache_len"] == ollm.kv_len(st.past_key_values)
