"""The drop-in boundary's upper surface on the GPU: `InfiniSST(args)` built from parsed SimulEval flags ONLY (reference
agents/infinisst.py:69-113,130-183) -- real transformers tokenizer directory, `pytorch_model.bin` on disk (un-pruned layout, pos_conv and
rotary `freqs` tensors present), real library -- and `Engine.load_checkpoint` itself."""
import argparse

import numpy as np
import pytest
import torch

from infinisst_amd import harness as H
from infinisst_amd import synth
from infinisst_amd.agent import InfiniSST, feed_segments
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine
from oracle import agent as oag
from oracle import llm as ollm
from oracle import speech_encoder as oenc
from tiny_tokenizer import build_tokenizer_dir

pytestmark = pytest.mark.gpu
ENC = "model.speech_encoder.speech_encoder."


def _checkpoint(tmp_path, cfg, w, freqs=None, prefix=""):
    state = {prefix + k: v.float() for k, v in w.items()}
    state[prefix + ENC + "encoder.pos_conv.0.bias"] = torch.zeros(cfg.enc_dim)
    state[prefix + ENC + "mask_emb"] = torch.zeros(cfg.enc_dim)
    if freqs is not None:
        for i in range(cfg.enc_layers):
            state[f"{prefix}{ENC}encoder.layers.{i}.self_attn.rotary_emb.freqs"] = freqs
    path = tmp_path / f"pytorch_model_{'pruned' if not prefix else 'lightning'}_{'f' if freqs is not None else 'd'}.bin"
    torch.save(state, path)
    return str(path)


def test_load_checkpoint_on_the_gpu(tmp_path):
    """`Engine.load_checkpoint(pytorch_model.bin)` == `load_weights` of the same tensors, bit for bit, for the pruned and the
    Lightning-prefixed layout; a checkpoint whose rotary `freqs` differ from the module default changes the encoder output exactly as
    the oracle says (the table is built from the checkpoint's parameter: strict load_state_dict semantics)."""
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=61)
    gen = GenConfig(max_new_tokens=5, max_llm_cache_size=300)
    audio = synth.synthetic_audio(cfg.chunk_samples * 2, stream_id=8)

    def run(load):
        eng = Engine(cfg, max_streams=1, max_prompt_len=96, max_new_tokens=8, max_llm_cache_size=300, max_system_prompt=64)
        skipped = load(eng)
        sid = eng.open_stream()
        logits, feats = [], []
        for c in range(2):
            seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
            o, l = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]], return_logits=True)
            logits.append(l[0][:len(o[0])].copy())
        sid2 = eng.open_stream() if False else sid
        eng.reset_stream(sid2)
        feats = [eng.encode_speech(sid2, audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]).float() for c in range(2)]
        eng.close()
        return logits, feats, skipped

    base_logits, base_feats, _ = run(lambda e: e.load_weights(w))
    for prefix in ("", "model."):
        logits, feats, skipped = run(lambda e: e.load_checkpoint(_checkpoint(tmp_path, cfg, w, None, prefix)))
        assert len(skipped) == 2
        for a, b in zip(logits, base_logits):
            assert np.array_equal(a, b), "load_checkpoint must give the engine exactly the tensors load_weights gives it"
    freqs = (1.0 / (10000 ** (torch.arange(0, 64, 2).float() / 64))) * 1.3
    logits_f, feats_f, skipped = run(lambda e: e.load_checkpoint(_checkpoint(tmp_path, cfg, w, freqs, "model.")))
    assert len(skipped) == 2 + cfg.enc_layers
    rope_f = oenc.make_rope(cfg, inv_freq=freqs)
    cache = oenc.new_cache(cfg)
    moved = 0.0
    for c in range(2):
        x = torch.from_numpy(audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples])
        if c == 0:
            x = torch.cat([torch.zeros(cfg.first_chunk_offset), x])
        ref, cache = oenc.encode_speech(w, cfg, x.unsqueeze(0).bfloat16(), cache, 1, rope_f)
        d = float((feats_f[c] - ref[0].float()).abs().max())
        moved = max(moved, float((feats_f[c] - base_feats[c]).abs().max()))
        print(f"chunk {c}: checkpoint freqs: HIP vs oracle max |d| {d:.4f}")
        assert d <= 0.06 + 0.02 * float(ref.abs().max())
    assert moved > 0.0, "non-default rotary freqs must change the encoder output"


@pytest.mark.parametrize("rope", [1, 0])
def test_agent_from_simuleval_args_runs_an_utterance_like_the_oracle(tmp_path, rope):
    """`--rope 1 --xpos 0`: the production flags.  `--rope 0 --xpos 1`: the reference's absolute-position encoder (its default --xpos 1 is inert then)."""
    cfg = toy_config().replace(enc_rope=bool(rope))
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=62, recipe="peaked")  # decisive greedy steps: synth.apply_recipe
    model_dir = build_tokenizer_dir(tmp_path, cfg)
    ckpt = _checkpoint(tmp_path, cfg, w, None, "model.")
    parser = argparse.ArgumentParser()
    InfiniSST.add_args(parser)
    args = parser.parse_args(["--model-name", model_dir, "--state-dict-path", ckpt, "--w2v2-type", "w2v2", "--w2v2-path", "unused.pt",
                              "--ctc-finetuned", "True", "--length-shrink-cfg", "[(128,2,2)] * 2", "--block-size", "48", "--max-cache-size", "576",
                              "--xpos", "0" if rope else "1", "--rope", str(rope), "--max-llm-cache-size", "150", "--always-cache-system-prompt", "--max-new-tokens", "6", "--beam", "1",
                              "--no-repeat-ngram-lookback", "100", "--no-repeat-ngram-size", "5", "--repetition-penalty", "1.2",
                              "--latency-multiplier", "1", "--max-latency-multiplier", "4", "--min-start-sec", "0", "--suppress-non-language",
                              "--source-lang", "English", "--target-lang", "German"])
    agent = InfiniSST(args)   # <- everything SimulEval does
    assert agent.bad_words_ids == [7] and agent.cfg.vocab == cfg.vocab and agent.cfg.eos_ids == cfg.eos_ids and agent.cfg.enc_rope == bool(rope)
    wav = synth.synthetic_audio(cfg.chunk_samples * 7 + 3000, stream_id=13)   # 8 segments: evictions at max_llm_cache_size 150
    inst = H.evaluate(agent, [("u0.wav", wav)], references=["a b"], output_dir=str(tmp_path / "out"))
    assert (tmp_path / "out" / "instances.log").exists() and len(inst) == 1
    got = list(agent.states.target_ids)

    # the same utterance through the CPU oracle agent with the reference-pinned prompts
    import transformers
    tok = transformers.AutoTokenizer.from_pretrained(model_dir, padding_side="right", use_fast=False)
    tok.pad_token = H.PAD_TOKEN
    H.preprocess_tokenizer(tok, 4)
    ref_prompt = H.ChatPrompt(tok, "English", "German", cfg.block_size, True)
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=150, suppress_tokens=(7,))
    oa = oag.OracleAgent(w, cfg, gen, lambda first: ref_prompt(first, 1), system_prompt_size=agent.system_prompt_size)
    st = oa.build_states()
    st.source_sample_rate = 16000
    margins = []
    for pos in range(0, wav.shape[0], cfg.chunk_samples):
        st.source.extend(wav[pos:pos + cfg.chunk_samples].tolist())
        st.source_finished = pos + cfg.chunk_samples >= wav.shape[0]
        oa.policy(st)
        for sc in oa.last_output.step_scores[:-1]:  # the steps whose tokens entered target_ids
            top2 = torch.topk(sc, 2).values
            margins.append(float(top2[0] - top2[1]))
    ref = list(st.target_ids)
    print("agent(args) ids:", got, "\noracle ids:     ", ref)
    # identical up to the first step whose oracle margin is within bf16 noise (2 x the logit tolerance); everything before it must agree
    first_tie = next((i for i, m in enumerate(margins) if m <= 0.3), len(margins))
    k = next((i for i, (a, b) in enumerate(zip(got, ref)) if a != b), min(len(got), len(ref)))
    decisive = sum(m > 0.3 for m in margins)
    assert decisive >= 24, f"the peaked recipe must give decisive steps ({decisive} of {len(margins)})"
    # (without rotary positions the fifth token of a chunk is a near-tie of this recipe; the four before it, and 30+ later ones, are not)
    assert first_tie >= min(12 if rope else 4, len(ref)), f"first near-tie at {first_tie} of {len(ref)}"
    assert k >= min(first_tie, len(ref)), f"ids part at {k}, before the first near-tie at {first_tie}"
    if got == ref:
        assert agent.engine.stream_info(agent.states.stream_id)["llm_cache_len"] == ollm.kv_len(st.past_key_values)
        assert agent.cache_checkpoints == oa.cache_checkpoints
