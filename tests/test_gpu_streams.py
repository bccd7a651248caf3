"""streams.StreamBatch on the GPU: the product's multi-stream driver -- what BASELINE.json configs[2] / [3] run --
against 64 single-stream agents (agent.InfiniSST.policy = the reference's policy() for one stream)."""
import numpy as np
import pytest
import torch

from infinisst_amd import synth
from infinisst_amd.agent import InfiniSST, default_args
from infinisst_amd.config import GenConfig, toy_config
from infinisst_amd.engine import Engine
from infinisst_amd.streams import StreamBatch

pytestmark = pytest.mark.gpu


class _Recording:
    """Passes everything through to an Engine, asks for the logits of every generate call and keeps them per stream."""

    def __init__(self, eng):
        self.eng, self.last = eng, {}

    def __getattr__(self, name):
        return getattr(self.eng, name)

    def generate(self, gen, sids, pcm, prompts, prevs, system_prompt_size=0, **kw):
        outs, logits = self.eng.generate(gen, sids, pcm, prompts, prevs, system_prompt_size=system_prompt_size,
                                         forced_tokens=kw.get("forced_tokens"), return_logits=True)
        self.last_call_seconds = self.eng.last_call_seconds
        for i, sid in enumerate(sids):
            self.last[sid] = (list(outs[i]), logits[i].copy(), [int(t) for t in prompts[i]], [] if prevs[i] is None else [int(t) for t in prevs[i]])
        return outs, None


def test_stream_batch_of_64_equals_64_single_stream_agents():
    """64 streams, 9 ticks, toy width.  A stream brings a chunk on ~75 % of the ticks (the others are skipped), a quarter of the
    streams join late -- their first chunk (system prompt, pinned) shares the call with other streams' later chunks -- and the 160-entry
    budget forces whole-chunk evictions at stream-specific times.  Every stream is ALSO run alone through agent.InfiniSST on a second
    engine; the batch is teacher-forced with the agent's tokens (bf16 near-ties of random weights may flip between the 1-row and the
    many-row kernels), and must reproduce, per stream: the prompt handed to the library, the encoder-n-gram window, the logits of every
    step (to fp32 summation order), the cache length, the checkpoint list and the eviction count."""
    cfg = toy_config()
    N, TICKS = 64, 9
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=81)
    sys_n = len(synth.system_prompt_ids(cfg, 1))
    kw = dict(max_multiplier=1, max_prompt_len=96, max_new_tokens=16, max_llm_cache_size=160, max_system_prompt=sys_n)
    eng_b, eng_s = Engine(cfg, max_streams=N, **kw), Engine(cfg, max_streams=N, **kw)
    eng_b.load_weights(w)
    eng_s.load_weights(w)
    rec_b, rec_s = _Recording(eng_b), _Recording(eng_s)
    gen = GenConfig(max_new_tokens=6, max_llm_cache_size=160, always_cache_system_prompt=True)
    args = default_args(max_llm_cache_size=160, always_cache_system_prompt=True, max_new_tokens=6)
    agents = [InfiniSST(args, engine=rec_s, model_cfg=cfg) for _ in range(N)]  # each constructor opens the agent's own stream
    for a in agents:
        a.states.source_sample_rate = 16000
    batch = StreamBatch(rec_b, gen, sys_n, lambda first, m: synth.chunk_prompt_ids(cfg, m, first=first))
    idx = [batch.open() for _ in range(N)]
    rng = np.random.default_rng(3)
    audio = [synth.synthetic_audio(cfg.chunk_samples * TICKS, stream_id=200 + k) for k in range(N)]
    fed = [0] * N
    worst, steps, calls_with_ragged_prompts = 0.0, 0, 0
    for t in range(TICKS):
        present = [(rng.random() < 0.75 or t == 0) and not (k % 4 == 3 and t < 3) for k in range(N)]
        segs, forced = [None] * N, [None] * N
        for k in range(N):
            if not present[k]:
                continue
            seg = audio[k][fed[k] * cfg.chunk_samples:(fed[k] + 1) * cfg.chunk_samples]
            fed[k] += 1
            segs[k] = seg
            agents[k].states.source.extend(seg.tolist())
            agents[k].policy(agents[k].states)
            forced[k] = rec_s.last[agents[k].states.stream_id][0]
        outs = batch.step(segs, forced_tokens=forced)
        lens = {len(rec_b.last[batch.stream_id(idx[k])][2]) for k in range(N) if present[k]}
        calls_with_ragged_prompts += int(len(lens) > 1)
        for k in range(N):
            if not present[k]:
                assert outs[k] is None
                continue
            g_s, l_s, p_s, e_s = rec_s.last[agents[k].states.stream_id]
            g_b, l_b, p_b, e_b = rec_b.last[batch.stream_id(idx[k])]
            assert g_b == g_s and outs[k] == g_s[:-1], f"tick {t} stream {k}"
            assert p_b == p_s and e_b == e_s, f"tick {t} stream {k}: prompt / encoder-n-gram window"
            d = float(np.abs(l_b[:len(g_s)] - l_s[:len(g_s)]).max())
            worst, steps = max(worst, d), steps + len(g_s)
            assert d <= 0.07, f"tick {t} stream {k}: batched vs single logits differ by {d}"
            assert batch.cache_len(idx[k]) == eng_s.stream_info(agents[k].states.stream_id)["llm_cache_len"]
            assert batch.slots[idx[k]].ckpts == agents[k].cache_checkpoints
            assert batch.slots[idx[k]].target_ids[-100:] == agents[k].states.target_ids[-100:]
    n_evict = batch.evictions
    print(f"StreamBatch vs 64 agents: {steps} steps, worst |logit d| {worst:.4f}, {n_evict} evictions, {calls_with_ragged_prompts} calls with ragged prompts, "
          f"host {1e3 * batch.host_seconds / batch.ticks:.2f} ms per tick outside the library")
    assert n_evict >= N // 2 and calls_with_ragged_prompts >= 1
    # free-running: the same batch composition twice gives the same ids (deterministic kernels), whatever the other streams do
    eng_b.close()
    eng_s.close()
