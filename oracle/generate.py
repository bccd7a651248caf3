"""ORACLE (test infrastructure, not product code) -- greedy generation loop and logits processors.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates what the reference's patched `generate` does on the greedy/sample branch
(reference model/patches/patch_hf.py:586-624) with the arguments the agent passes
(reference agents/infinisst.py:307-332).  The processors and `_sample` themselves live in the un-vendored
transformers==4.47.0: order RepetitionPenalty -> NoRepeatNGram -> EncoderNoRepeatNGram -> SuppressTokens, applied to the
fp32 copy of the last position's logits; argmax; stop on EOS or max length.  `process_logits` is pinned bit-exactly against
HF's own processor classes as shipped in this image (transformers 5.15.0, tests/golden/logits_processors.npz); the 4.47.0
release itself is absent, so the pin is to the upstream implementation of a later version ("parity pinned to 5.15, not 4.47").

Greedy deviation: the reference asserts beam > 1 (agents/infinisst.py:86); the north star asks for greedy,
which is this path with that assert waived.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import llm as ollm
from . import speech_encoder as oenc


def repetition_penalty_(scores: torch.Tensor, input_ids: Sequence[int], penalty: float) -> None:
    """[3P] RepetitionPenaltyLogitsProcessor: gather, (<0 ? *p : /p), scatter.  scores (V,) fp32, in place."""
    if penalty == 1.0 or len(input_ids) == 0:
        return
    idx = torch.tensor(sorted(set(int(t) for t in input_ids)), dtype=torch.long)
    s = scores[idx]
    scores[idx] = torch.where(s < 0, s * penalty, s / penalty)


def banned_ngram_tokens(source_ids: Sequence[int], context_ids: Sequence[int], n: int) -> List[int]:
    """Tokens t such that (last n-1 tokens of `context_ids`) + (t,) occurs as an n-gram in `source_ids`.
    [3P] _get_ngrams + _get_generated_ngrams.  NoRepeatNGram: source == context == input_ids;
    EncoderNoRepeatNGram: source == encoder_input_ids, context == input_ids."""
    if n <= 0:
        return []
    cur_len = len(context_ids)
    start = cur_len + 1 - n
    if start < 0:  # fewer than n-1 context tokens: HF's negative slice yields a shorter key -> no match
        return []
    key = tuple(int(t) for t in context_ids[start:cur_len])
    src = list(source_ids)
    out = []
    for j in range(len(src) - n + 1):
        if tuple(src[j: j + n - 1]) == key:
            out.append(int(src[j + n - 1]))
    return out


def process_logits(scores: torch.Tensor, input_ids: Sequence[int], encoder_input_ids: Sequence[int],
                   repetition_penalty: float, no_repeat_ngram_size: int, encoder_no_repeat_ngram_size: int,
                   suppress_tokens: Sequence[int]) -> torch.Tensor:
    """scores (V,) fp32 raw logits of the last position -> processed copy."""
    s = scores.clone()
    repetition_penalty_(s, input_ids, repetition_penalty)
    if no_repeat_ngram_size > 0 and len(input_ids) + 1 >= no_repeat_ngram_size:
        for t in banned_ngram_tokens(input_ids, input_ids, no_repeat_ngram_size):
            s[t] = float("-inf")
    if encoder_no_repeat_ngram_size > 0:
        for t in banned_ngram_tokens(encoder_input_ids, input_ids, encoder_no_repeat_ngram_size):
            s[t] = float("-inf")
    for t in suppress_tokens:
        s[int(t)] = float("-inf")
    return s


def warp_logits(scores: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, epsilon_cutoff: float = 0.0,
                min_tokens_to_keep: int = 1) -> torch.Tensor:
    """[3P transformers] the warpers `_get_logits_processor` appends for do_sample, in its order: TemperatureLogitsWarper -> TopKLogitsWarper ->
    TopPLogitsWarper -> EpsilonLogitsWarper (filter value -inf), restated from that library's published code and pinned to the image's transformers 5.15
    classes (tests/golden/sampling_warpers.npz).  `min_tokens_to_keep`: 1 for the sample branch, len(eos ids) + 1 under beam search ("keep at least one
    non-eos token", the same function).  scores (V,) fp32, processed -> warped copy."""
    s = scores.clone()
    ninf = float("-inf")
    keep = max(1, int(min_tokens_to_keep))
    if temperature > 0 and temperature != 1.0:
        s = s / temperature
    if top_k > 0:
        k = min(max(top_k, keep), s.numel())
        s = s.masked_fill(s < torch.topk(s, k)[0][-1], ninf)
    if top_p < 1.0:
        sorted_logits, sorted_idx = torch.sort(s, descending=False)
        cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
        remove = cum <= (1 - top_p)
        remove[-keep:] = False
        s = s.masked_fill(remove.scatter(0, sorted_idx, remove), ninf)
    if 0 < epsilon_cutoff < 1:
        probs = s.softmax(dim=-1)
        s = s.masked_fill((probs < epsilon_cutoff) & (s < torch.topk(s, min(keep, s.numel()))[0][-1]), ninf)
    return s


_M64 = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + 0x9E3779B97F4A7C15) & _M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _M64
    return x ^ (x >> 31)


def sample_uniform(seed: int, stream: int, chunk: int, step: int) -> float:
    """The uniform in [0, 1) of one draw: counter-based, keyed by (seed, stream, chunk, step) -- csrc/warp.hip's generator, restated."""
    x = _splitmix64(seed & _M64)
    x = _splitmix64(x ^ (stream & 0xFFFFFFFF))
    x = _splitmix64(x ^ ((chunk & 0xFFFFFFFF) << 20))
    x = _splitmix64(x ^ (step & 0xFFFFFFFF))
    return (x >> 11) / 9007199254740992.0


def draw(warped: torch.Tensor, u: float) -> int:
    """softmax + ONE multinomial draw (HF _sample: probs = softmax(scores); torch.multinomial(probs, 1)) as the inverse CDF in vocabulary order at u.
    (torch.multinomial's own random stream is not reproducible across libraries; the DISTRIBUTION is the reference's.)"""
    p = warped.softmax(dim=-1).double()
    cum = torch.cumsum(p, dim=0)
    idx = int(torch.searchsorted(cum, torch.tensor(u * float(cum[-1]), dtype=torch.float64), right=True))  # first index whose running sum exceeds the target
    return idx if idx < p.numel() else int(torch.nonzero(p > 0).flatten()[-1])


def multinomial_without_replacement(scores: torch.Tensor, n: int, uniforms: Sequence[float]) -> List[int]:
    """`torch.multinomial(softmax(scores), num_samples=n)` (replacement=False: patch_hf.py:871-873, the beam-sample branch) as n sequential inverse-CDF draws
    in index order at the given uniforms, each over what is left (the definition of sampling without replacement; torch's own stream of random numbers cannot be
    reproduced).  softmax in fp32 as the reference computes it, the running sums in fp64, left to right.  Raises like torch when fewer than n entries have
    non-zero probability."""
    return draw_without_replacement(torch.softmax(scores.float(), dim=-1), n, uniforms)


def draw_without_replacement(probs: torch.Tensor, n: int, uniforms: Sequence[float]) -> List[int]:
    """The draws of `multinomial_without_replacement` from probabilities (what torch.multinomial is handed at patch_hf.py:873)."""
    p = probs.double().numpy().copy()
    picked: List[int] = []
    for j in range(n):
        cum = np.cumsum(p)  # sequential fp64 adds, index order
        total = float(cum[-1]) if cum.size else 0.0
        if not total > 0.0:
            raise RuntimeError("invalid multinomial distribution (with replacement=False, not enough non-negative category to sample)")
        idx = int(np.searchsorted(cum, uniforms[j] * total, side="right"))  # first index whose running sum exceeds the target
        if idx >= p.size:  # the target fell on the very end: the last live entry
            idx = int(np.flatnonzero(p > 0)[-1])
        picked.append(idx)
        p[idx] = 0.0
    return picked


@dataclass
class GenerateOutput:
    sequences: List[int]  # prompt + generated (the last generated token is never fed to the model)
    step_logits: List[torch.Tensor]  # raw last-position logits per step, model dtype
    step_scores: List[torch.Tensor]  # processed fp32 scores per step
    speech_features: Optional[torch.Tensor] = None


def generate(w: Dict[str, torch.Tensor], cfg, gen, input_ids: List[int], speech_batch: torch.Tensor, kv,
             speech_cache, rope_llm, rope_enc, encoder_input_ids: Sequence[int],
             forced_tokens: Optional[Sequence[int]] = None,
             keep_logits: bool = True, stream: int = 0, chunk: int = 0) -> GenerateOutput:
    """One chunk: encoder (step 0) + prefill + greedy decode (or, gen.do_sample, the sample branch: warpers + one draw per step at
    sample_uniform(gen.seed, stream, chunk, step)).  Mutates `kv` and `speech_cache`.

    `forced_tokens` (teacher forcing, test aid): token j of the list is appended instead of the argmax at step j;
    the loop then runs exactly len(forced_tokens) steps unless EOS/max length stops it first."""
    m = gen.latency_multiplier
    feats, _ = oenc.encode_speech(w, cfg, speech_batch, speech_cache, m, rope_enc)  # model/llm.py:69-81
    feats = feats[0]
    seq = list(input_ids)
    max_length = len(seq) + gen.max_new_tokens
    out = GenerateOutput(sequences=seq, step_logits=[], step_scores=[], speech_features=feats)
    step = 0
    while True:
        if step == 0:
            logits = ollm.model_forward(w, cfg, torch.tensor(seq), kv, rope_llm, speech=feats)
        else:
            logits = ollm.model_forward(w, cfg, torch.tensor(seq[-1:]), kv, rope_llm)
        raw = logits.float()  # outputs.logits[:, -1, :].float()
        scores = process_logits(raw, seq, encoder_input_ids, gen.repetition_penalty, gen.no_repeat_ngram_size,
                                gen.no_repeat_ngram_size, gen.suppress_tokens)
        if keep_logits:
            out.step_logits.append(logits)
            out.step_scores.append(scores)
        if forced_tokens is not None and step < len(forced_tokens):
            tok = int(forced_tokens[step])
        elif getattr(gen, "do_sample", False):  # patch_hf.py:606-624 -> HF _sample with do_sample
            warped = warp_logits(scores, gen.temperature, gen.top_k, gen.top_p, gen.epsilon_cutoff)
            if keep_logits:
                out.step_scores[-1] = warped
            tok = draw(warped, sample_uniform(gen.seed, stream, chunk, step))
        else:
            tok = int(torch.argmax(scores))
        seq.append(tok)
        step += 1
        if tok in cfg.eos_ids or len(seq) >= max_length:
            break
        if forced_tokens is not None and step >= len(forced_tokens):
            break
    return out
