"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference agent's per-chunk logic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows reference agents/infinisst.py:50-67 (states), :200-223 (_prepare_speech), :270-395 (policy:
gating, generate, LLM-KV eviction by whole chunks, output slicing).  The tokenizer / chat template are
third-party and absent, so prompts come from a `prompt_fn(first: bool) -> list[int]` and detokenisation from
`decode_fn(list[int]) -> str` supplied by the caller.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional

import torch

from . import generate as ogen
from . import llm as ollm
from . import speech_encoder as oenc


@dataclass
class ReadAction:
    pass


@dataclass
class WriteAction:
    content: str
    finished: bool


@dataclass
class States:
    """simuleval AgentStates fields the policy reads + S2TAgentStates (agents/infinisst.py:50-67)."""
    source: List[float] = field(default_factory=list)
    source_sample_rate: int = 0
    source_finished: bool = False
    target: List[str] = field(default_factory=list)
    src_len: int = 0
    speech_cache: Optional[object] = None
    past_key_values: Optional[object] = None
    target_ids: List[int] = field(default_factory=list)
    segment_idx: int = 0
    MAX_SRC_LEN = 1600000


def prepare_speech(cfg, states: States, dtype) -> torch.Tensor:
    """agents/infinisst.py:200-223: new samples since src_len, zero-pad to a multiple of 15360,
    prepend 399 zeros on the first chunk, cast to the model dtype."""
    seg = cfg.chunk_samples
    if len(states.source) > states.MAX_SRC_LEN:  # :204-206
        states.src_len -= len(states.source) - states.MAX_SRC_LEN
        states.source = states.source[-states.MAX_SRC_LEN:]
    source = torch.tensor(states.source[states.src_len:], dtype=torch.float32)
    if source.size(0) % seg != 0:  # :211-213
        source = torch.cat([source, torch.zeros(seg - source.size(0) % seg)], dim=0)
    if states.src_len == 0:  # :216-218
        source = torch.cat([torch.zeros(cfg.first_chunk_offset), source], dim=0)
    states.src_len = len(states.source)
    return source.unsqueeze(0).to(dtype)


def evict(cache_checkpoints: List[int], cur: int, max_size: int, keep_system: bool, system_prompt_size: int):
    """agents/infinisst.py:340-352.  Returns (new_checkpoints, new_llm_cache_size) or None when cur <= max.
    `cache_checkpoints` already contains `cur` as its last element."""
    if cur <= max_size:
        return None
    new_size = 0
    ckpts = list(cache_checkpoints)
    for i, ckpt in enumerate(cache_checkpoints):
        new_size = cur - ckpt
        if new_size <= max_size:
            ckpts = cache_checkpoints[i + 1:]
            trimmed = ckpt
            if keep_system:
                trimmed -= system_prompt_size
            ckpts = [c - trimmed for c in ckpts]
            break
    return ckpts, new_size


class OracleAgent:
    """Per-chunk policy of InfiniSST (agents/infinisst.py:270-395) over the oracle model."""

    def __init__(self, w, cfg, gen, prompt_fn: Callable[[bool], List[int]],
                 decode_fn: Callable[[List[int]], str] = lambda ids: " ".join(str(i) for i in ids),
                 min_start_sec: float = 0.0, system_prompt_size: Optional[int] = None, target_lang: str = "German"):
        self.w, self.cfg, self.gen = w, cfg, gen
        self.dtype = w["lm_head.weight"].dtype
        self.prompt_fn, self.decode_fn = prompt_fn, decode_fn
        self.min_start_sec = min_start_sec
        self.cache_checkpoints: List[int] = []  # agent-level, NOT reset per utterance (agents/infinisst.py:106)
        self.system_prompt_size = system_prompt_size
        self.target_lang = target_lang
        self.rope_llm = ollm.llm_rope_tables(cfg, 4096 + gen.max_llm_cache_size, self.dtype)
        self.rope_enc = oenc.make_rope(cfg)
        self.last_output = None

    def build_states(self) -> States:
        return States()

    def policy(self, states: States):
        cfg, gen = self.cfg, self.gen
        length = 0.0 if states.source_sample_rate == 0 else len(states.source) / states.source_sample_rate
        if not states.source_finished and length < self.min_start_sec:  # :281-282
            return ReadAction()
        if states.source_finished and length < 0.32:  # :284-285
            return WriteAction("", True)
        first = states.speech_cache is None
        speech = prepare_speech(cfg, states, self.dtype)
        prompt = self.prompt_fn(first)
        if first and self.system_prompt_size is None:
            raise ValueError("system_prompt_size must be given (the chat template is not available)")
        if first:
            states.speech_cache = oenc.new_cache(cfg)
            states.past_key_values = ollm.new_kv(cfg)
        enc_ids = states.target_ids[-gen.no_repeat_ngram_lookback:]  # :298-300
        out = ogen.generate(self.w, cfg, gen, prompt, speech, states.past_key_values, states.speech_cache,
                            self.rope_llm, self.rope_enc, enc_ids)
        self.last_output = out
        kv = states.past_key_values
        cur = ollm.kv_len(kv)  # :337
        self.cache_checkpoints.append(cur)
        ev = evict(self.cache_checkpoints, cur, gen.max_llm_cache_size, gen.always_cache_system_prompt,
                   self.system_prompt_size)
        if ev is not None:  # :354-361
            self.cache_checkpoints, new_size = ev
            for layer in kv:
                for j in (0, 1):
                    t = layer[j]
                    tail = t[:, :, -new_size:] if new_size > 0 else t[:, :, :0]
                    if gen.always_cache_system_prompt:
                        tail = torch.cat([t[:, :, : self.system_prompt_size], tail], dim=2)
                    layer[j] = tail
        output_ids = out.sequences[len(prompt):-1]  # :363
        states.target_ids.extend(output_ids)
        translation = self.decode_fn(output_ids).strip().replace("�", "")
        states.segment_idx += 1
        if translation != "" or states.source_finished:  # :389-395
            if translation:
                states.target.append(translation)
            return WriteAction(translation, states.source_finished)
        return ReadAction()
