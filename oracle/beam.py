"""ORACLE (test infrastructure, not product code) -- beam search with per-hypothesis KV caches.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates the reference's patched HF beam search (the production decoding mode, beam=4):
  generation_mixin_beam_search   model/patches/patch_hf.py:687-967   (loop: log_softmax -> processors -> + beam scores ->
                                                                     top-k over beams*vocab -> scorer -> reorder KV)
  beam_search_process            :43-157    (EOS candidates inside the top `num_beams` ranks become hypotheses and take
                                             a COPY of that beam's KV cache; the others fill the next beams)
  beam_search_finalize           :159-275   (open beams become hypotheses; best one wins; EOS appended if it fits)
  beam_hypotheses_add            :278-302   (score = sum_logprobs / generated_len ** length_penalty, keep the best n)
  _expand_inputs_for_generation  :305-342   (inputs and KV cache repeated num_beams times)
`scorer_process`, `BeamHypotheses.add` and `finalize` are pinned against the reference's own functions, executed from their
source on a stand-in scorer (tests/golden/gen_golden.py::gen_beam_scorer -> beam_scorer.npz, tests/test_oracle_golden.py).
The LOOP (`beam_search_loop`: expansion, log-softmax, processors on log-probs, + beam scores, top-k over beams x vocab incl.
`n_tokens_to_keep`, reorder, stop test, finalize) is pinned the same way: gen_golden.py::gen_beam_loop compiles
`generation_mixin_beam_search` and `_expand_inputs_for_generation` from patch_hf.py's own text, drives them on a toy model whose
logits depend on the whole per-beam cache, and stores every step's candidates, chosen (token, parent) pairs, the winning
sequence and the cache that travels with it (beam_loop.npz, 6 cases incl. EOS-closed hypotheses and a non-empty past cache).
transformers 4.47.0 cannot be imported here, so `BeamHypotheses.is_done` (early_stopping=False heuristic), `BeamSearchScorer.is_done`
and DynamicCache (update / reorder_cache) stay restated from that release's published code: parity unpinned for those three.

The model is the batch-1 oracle; every beam carries its own KV list (what `_temporary_reorder_cache` + index_select
achieve in the reference).
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from . import generate as ogen
from . import llm as ollm
from . import speech_encoder as oenc


def clone_kv(kv):
    return [[None if t is None else t.clone() for t in layer] for layer in kv]


@dataclass
class BeamHypotheses:
    num_beams: int
    length_penalty: float = 1.0
    early_stopping: bool = False
    beams: list = field(default_factory=list)  # (score, tokens, kv)
    worst_score: float = 1e9

    def __len__(self):
        return len(self.beams)

    def add(self, hyp: List[int], sum_logprobs: float, generated_len: int, kv):
        """patch_hf.py:278-302."""
        score = sum_logprobs / (generated_len ** self.length_penalty)
        if len(self) < self.num_beams or score > self.worst_score:
            self.beams.append((score, list(hyp), kv))
            if len(self) > self.num_beams:
                ranked = sorted((s, idx) for idx, (s, _, _) in enumerate(self.beams))
                del self.beams[ranked[0][1]]
                self.worst_score = ranked[1][0]
            else:
                self.worst_score = min(score, self.worst_score)

    def is_done(self, best_sum_logprobs: float, cur_len: int, decoder_prompt_len: int) -> bool:
        """[3P transformers 4.47 BeamHypotheses.is_done], early_stopping=False branch."""
        if len(self) < self.num_beams:
            return False
        if self.early_stopping is True:
            return True
        highest_attainable = best_sum_logprobs / (cur_len - decoder_prompt_len) ** self.length_penalty
        return self.worst_score >= highest_attainable


@dataclass
class BeamStepRecord:
    scores: List[torch.Tensor]  # per beam: processed log-probs + beam score (fp32, V)
    cand_scores: List[float]
    cand_tokens: List[int]
    cand_beams: List[int]
    next_tokens: List[int]
    next_parents: List[int]
    next_scores: List[float]
    beam_scores_in: List[float] = field(default_factory=list)  # the beams' scores BEFORE this step (scores[b] = processed log-probs + beam_scores_in[b])


@dataclass
class BeamOutput:
    sequences: List[int]  # prompt + best hypothesis (+ EOS if it fits under max_length)
    kv: list  # KV cache of the best hypothesis
    steps: List[BeamStepRecord]
    speech_features: Optional[torch.Tensor] = None


def scorer_process(hyps: BeamHypotheses, done: bool, input_ids: List[List[int]], cand_scores, cand_tokens, cand_beams, kvs,
                   eos_ids: Sequence[int], num_beams: int, decoder_prompt_len: int, clone=clone_kv):
    """beam_search_process for batch size 1 (patch_hf.py:43-157).  Returns (next_scores, next_tokens, next_parents, done)."""
    cur_len = len(input_ids[0]) + 1
    next_scores, next_tokens, next_parents = [], [], []
    for rank, (tok, sc, b) in enumerate(zip(cand_tokens, cand_scores, cand_beams)):
        if tok in eos_ids:
            if rank >= num_beams:
                continue
            hyps.add(input_ids[b], sc, cur_len - decoder_prompt_len, clone(kvs[b]))
        else:
            next_scores.append(sc)
            next_tokens.append(tok)
            next_parents.append(b)
        if len(next_tokens) == num_beams:
            break
    if len(next_tokens) < num_beams:
        raise ValueError("not enough non-EOS candidates")
    done = done or hyps.is_done(max(cand_scores), cur_len, decoder_prompt_len)
    return next_scores, next_tokens, next_parents, done


def finalize(hyps: BeamHypotheses, done: bool, seqs: List[List[int]], beam_scores: Sequence[float], kvs, decoder_prompt_len: int,
             max_length: int, first_eos: int):
    """beam_search_finalize for batch size 1, one hypothesis kept (patch_hf.py:159-275): open beams become hypotheses unless
    the scorer is done; the best-scoring hypothesis wins and takes its KV cache along; EOS is appended if it fits."""
    if not done:
        for b in range(len(seqs)):
            hyps.add(seqs[b], beam_scores[b], len(seqs[b]) - decoder_prompt_len, kvs[b])
    best = sorted(hyps.beams, key=lambda x: x[0])[-1]
    hyp_tokens, best_kv = best[1], best[2]
    sent_max_len = min(len(hyp_tokens) + 1, max_length)
    out = list(hyp_tokens)
    if len(hyp_tokens) < sent_max_len:
        out.append(first_eos)  # "inserting only the first eos_token_id"
    return out, best_kv


def beam_search_loop(forward, process, num_beams: int, input_ids: List[int], kv, eos_ids: Sequence[int], max_new_tokens: int,
                     length_penalty: float = 1.0, clone=clone_kv, draw=None):
    """generation_mixin_beam_search for batch size 1 (patch_hf.py:687-967) over a pluggable model.

    `forward(tokens, kv, first) -> logits (V,)` runs one beam's forward pass and appends to that beam's `kv` in place (step 0:
    the whole prompt, later steps: the last token -- model/llm.py:114-115); `process(log_probs, seq) -> scores` are the logits
    processors (:839, applied to log-probs; with do_sample the warpers are part of them).  `draw(flat_scores, n, step) -> indices`: the beam-SAMPLE branch
    (:871-875: softmax over all beams' scores, n draws without replacement, then sorted by score) instead of the top-k; None = beam search.
    Returns (sequence, best_kv, steps).  Pinned against the reference's own loop, compiled from patch_hf.py's text and driven on a toy model
    (tests/golden/gen_golden.py::gen_beam_loop -> beam_loop.npz; its do_sample cases run with torch.multinomial replaced by the same `draw`)."""
    prompt_len = len(input_ids)
    max_length = prompt_len + max_new_tokens
    n_keep = max(2, 1 + len(eos_ids)) * num_beams  # :869-870
    seqs = [list(input_ids) for _ in range(num_beams)]
    kvs = [clone(kv) for _ in range(num_beams)]  # _expand_inputs_for_generation :305-342
    beam_scores = [0.0] + [-1e9] * (num_beams - 1)  # :770-771
    hyps = BeamHypotheses(num_beams, length_penalty)
    done = False
    steps: List[BeamStepRecord] = []
    step = 0
    while True:
        rows = []
        for b in range(num_beams):
            logits = forward(seqs[b] if step == 0 else seqs[b][-1:], kvs[b], step == 0)
            lp = torch.log_softmax(logits.float(), dim=-1)  # :833-837
            sc = process(lp, seqs[b])  # :839, on log-probs
            rows.append(sc + torch.tensor(beam_scores[b], dtype=torch.float32))  # :840 (fp32 add, as beam_scores[:, None])
        V = rows[0].numel()
        flat = torch.cat(rows)
        if draw is None:
            top = torch.topk(flat, n_keep, largest=True, sorted=True)  # :878
            top_values, top_indices = top.values, top.indices
        else:  # :871-875
            picked = torch.tensor(draw(flat, n_keep, step), dtype=torch.long)
            vals, order = torch.sort(flat[picked], descending=True)
            top_values, top_indices = vals, picked[order]
        cand_scores = [float(x) for x in top_values]
        cand_beams = [int(i) // V for i in top_indices]
        cand_tokens = [int(i) % V for i in top_indices]
        ns, nt, npar, done = scorer_process(hyps, done, seqs, cand_scores, cand_tokens, cand_beams, kvs, eos_ids, num_beams, prompt_len, clone)
        steps.append(BeamStepRecord(rows, cand_scores, cand_tokens, cand_beams, nt, npar, ns, list(beam_scores)))
        seqs = [seqs[p] + [t] for p, t in zip(npar, nt)]  # :902 input_ids[beam_idx] + token
        kvs = [clone(kvs[p]) for p in npar]  # :910-913 _temporary_reorder_cache
        beam_scores = ns
        step += 1
        if done or len(seqs[0]) >= max_length:  # :921
            break
    out, best_kv = finalize(hyps, done, seqs, beam_scores, kvs, prompt_len, max_length, eos_ids[0] if eos_ids else 0)
    return out, best_kv, steps


def beam_generate(w, cfg, gen, num_beams: int, input_ids: List[int], speech_batch: torch.Tensor, kv, speech_cache, rope_llm,
                  rope_enc, encoder_input_ids: Sequence[int], length_penalty: float = 1.0, sample_stream: int = 0, sample_chunk: int = 0) -> BeamOutput:
    """One chunk with beam search.  `kv` (the stream's cache before this chunk) is not modified; the winning
    hypothesis' cache is returned (reference agents/infinisst.py:334-336: past_key_values[0])."""
    m = gen.latency_multiplier
    feats, _ = oenc.encode_speech(w, cfg, speech_batch, speech_cache, m, rope_enc)
    feats = feats[0]

    def forward(tokens, beam_kv, first):
        return ollm.model_forward(w, cfg, torch.tensor(tokens), beam_kv, rope_llm, speech=feats if first else None)

    def process(lp, seq):
        sc = ogen.process_logits(lp, seq, encoder_input_ids, gen.repetition_penalty, gen.no_repeat_ngram_size,
                                 gen.no_repeat_ngram_size, gen.suppress_tokens)
        if gen.do_sample:  # the warpers `_get_logits_processor` appends for do_sample are part of the processor list (beam_loop.npz records the order)
            sc = ogen.warp_logits(sc, gen.temperature, gen.top_k, gen.top_p, gen.epsilon_cutoff, min_tokens_to_keep=len(cfg.eos_ids) + 1)
        return sc

    draw = None
    if gen.do_sample:  # uniforms keyed like the greedy-sample branch; draw j of step s uses counter 64 s + j (csrc/warp.hip, engine_llm.hip beam_decode)
        def draw(flat, n, step):
            return ogen.multinomial_without_replacement(flat, n, [ogen.sample_uniform(gen.seed, sample_stream, sample_chunk, 64 * step + j) for j in range(n)])

    out, best_kv, steps = beam_search_loop(forward, process, num_beams, input_ids, kv, cfg.eos_ids, gen.max_new_tokens, length_penalty, draw=draw)
    return BeamOutput(sequences=out, kv=best_kv, steps=steps, speech_features=feats)
