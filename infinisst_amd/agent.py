"""SimulEval-shaped agent over the MI355X engine: the drop-in for reference agents/infinisst.py.

Same surface as the reference's `InfiniSST(SpeechToTextAgent)`: `add_args(parser)`, `__init__(args)`,
`build_states()`, `policy(states) -> ReadAction | WriteAction(content, finished)`, `update_multiplier(m)`,
flag names and defaults of agents/options.py and agents/infinisst.py:185-198.  SimulEval itself is a third-party
package that is not installed here, so minimal `ReadAction` / `WriteAction` / `AgentStates` stand-ins are defined
below; when simuleval is importable its classes are used instead, which makes this module loadable by
`simuleval --agent infinisst_amd/agent.py`.

What stays on the host (exactly as in the reference): gating (:275-285), chunk padding (:200-223), prompt
building (:225-268), the whole-chunk LLM-KV eviction policy with its `cache_checkpoints` list (:337-352) and the
output slicing / detokenisation (:363-395).  What moved to the GPU library: everything behind
`self.model.generate(...)` (:307-332) and the KV re-binding of the eviction (:354-361, now a ring-start advance).
"""
from __future__ import annotations

import argparse
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import synth
from .config import GenConfig, ModelConfig, full_config
from .engine import Engine

try:  # pragma: no cover - simuleval is absent in the build image
    from simuleval.agents import SpeechToTextAgent as _AgentBase
    from simuleval.agents.actions import ReadAction, WriteAction
    from simuleval.agents.states import AgentStates as _StatesBase
    HAVE_SIMULEVAL = True
except Exception:  # stand-ins with the fields the policy touches
    HAVE_SIMULEVAL = False

    @dataclass
    class ReadAction:
        pass

    @dataclass
    class WriteAction:
        content: str
        finished: bool

    class _StatesBase:
        def __init__(self):
            self.reset()

        def reset(self):
            self.source: List[float] = []
            self.target: List[str] = []
            self.source_finished = False
            self.target_finished = False
            self.source_sample_rate = 0

    class _AgentBase:
        def __init__(self, args=None):
            self.args = args
            self.states = self.build_states()


class S2TAgentStates(_StatesBase):
    """reference agents/infinisst.py:50-67; `speech_cache` / `past_key_values` live in the library, the state only
    carries the library's stream id."""
    MAX_SRC_LEN = 1600000

    def __init__(self, stream_id: Optional[int] = None):
        self.stream_id = stream_id
        super().__init__()

    def reset(self):
        super().reset()
        self.src_len = 0
        self.started = False  # reference: speech_cache is not None
        self.target_ids: List[int] = []
        self.segment_idx = 0
        self.translations_list: List[str] = []
        self.needs_reset = True


def add_speech_encoder_args(parser):  # reference agents/options.py:1-41
    parser.add_argument("--w2v2-path", type=str, default=None)
    parser.add_argument("--w2v2-type", type=str, default=None)
    parser.add_argument("--ctc-finetuned", type=lambda x: (str(x).lower() == "true"), default=False)
    parser.add_argument("--length-shrink-cfg", type=str, default=None)
    parser.add_argument("--block-size", type=int, default=12)
    parser.add_argument("--max-cache-size", type=int, default=125)
    parser.add_argument("--xpos", type=int, default=1)
    parser.add_argument("--rope", type=int, default=1)


def add_gen_args(parser):  # reference agents/options.py:43-108
    parser.add_argument("--max-len-a", type=int, default=5)
    parser.add_argument("--max-len-b", type=int, default=20)
    parser.add_argument("--beam", type=int, default=1)
    parser.add_argument("--no-repeat-ngram-lookback", type=int, default=100)
    parser.add_argument("--no-repeat-ngram-size", type=int, default=3)
    parser.add_argument("--repetition-penalty", type=float, default=1.2)
    parser.add_argument("--suppress-non-language", action="store_true")
    parser.add_argument("--max-new-tokens", type=int, default=1000)
    parser.add_argument("--do-sample", action="store_true")
    parser.add_argument("--top-p", type=float, default=1.0)
    parser.add_argument("--top-k", type=int, default=0)
    parser.add_argument("--epsilon-cutoff", type=float, default=0.0)
    parser.add_argument("--temperature", type=float, default=1.0)


def add_simuleval_args(parser):  # reference agents/options.py:110-125
    parser.add_argument("--source-lang", type=str, default="English")
    parser.add_argument("--target-lang", type=str, default="German")
    parser.add_argument("--min-start-sec", default=0.32, type=float)


class InfiniSST(_AgentBase):
    """MI355X-native InfiniSST agent (greedy decoding; see DESIGN.md for the beam>1 assert of the reference)."""

    def __init__(self, args, engine: Optional[Engine] = None, model_cfg: Optional[ModelConfig] = None,
                 prompt_fn: Optional[Callable[[bool, int], List[int]]] = None,
                 decode_fn: Optional[Callable[[Sequence[int]], str]] = None, system_prompt_size: Optional[int] = None,
                 weights=None):
        self.min_start_sec = args.min_start_sec
        self.latency_multiplier = args.latency_multiplier
        self.source_segment_size = getattr(args, "source_segment_size", 960 * args.latency_multiplier)
        self.max_latency_multiplier = args.max_latency_multiplier
        self.source_lang, self.target_lang = args.source_lang, args.target_lang
        self.beam = args.beam  # 1: greedy (the reference asserts beam > 1, agents/infinisst.py:86); >1: beam search
        if self.beam < 1:
            raise ValueError("--beam must be >= 1")
        self.no_repeat_ngram_lookback = args.no_repeat_ngram_lookback
        self.no_repeat_ngram_size = args.no_repeat_ngram_size
        self.repetition_penalty = args.repetition_penalty
        self.max_new_tokens = args.max_new_tokens
        if getattr(args, "do_sample", False):
            raise NotImplementedError("sampling is not part of the hot path (the reference scripts never enable it)")
        self.max_llm_cache_size = args.max_llm_cache_size
        self.always_cache_system_prompt = args.always_cache_system_prompt
        self.cache_checkpoints: List[int] = []  # agent-level, not reset per utterance (reference :106)
        self.bad_words_ids: List[int] = list(getattr(args, "bad_words_ids", []) or [])
        self.cfg = model_cfg or full_config().replace(block_size=args.block_size, max_cache_size=args.max_cache_size)
        self.prompt_fn = prompt_fn or (lambda first, m: synth.chunk_prompt_ids(self.cfg, m, first))
        self.decode_fn = decode_fn or (lambda ids: " ".join(str(i) for i in ids))
        self.system_prompt_size = system_prompt_size if system_prompt_size is not None else len(
            synth.system_prompt_ids(self.cfg, self.latency_multiplier))
        if engine is None:
            engine = Engine(self.cfg, max_streams=1, max_multiplier=self.max_latency_multiplier,
                            max_prompt_len=self.system_prompt_size + 16 + 12 * self.max_latency_multiplier * 1,
                            max_new_tokens=max(self.max_new_tokens, 10 * self.max_latency_multiplier),
                            max_llm_cache_size=self.max_llm_cache_size, max_system_prompt=self.system_prompt_size,
                            max_beams=self.beam)
            if weights is None:
                raise ValueError("either an engine with loaded weights or a weight dict is required")
            engine.load_weights(weights)
        self.engine = engine
        super().__init__(args)

    # ------------------------------------------------------------------ reference surface
    @staticmethod
    def add_args(parser):  # reference agents/infinisst.py:185-198
        add_simuleval_args(parser)
        add_speech_encoder_args(parser)
        add_gen_args(parser)
        parser.add_argument("--model-name", type=str, default="facebook/opt-350m")
        parser.add_argument("--state-dict-path", type=str, default=None)
        parser.add_argument("--latency-multiplier", type=int, default=4)
        parser.add_argument("--max-latency-multiplier", type=int, default=4)
        parser.add_argument("--max-llm-cache-size", type=int, default=10000)
        parser.add_argument("--always-cache-system-prompt", action="store_true")
        parser.add_argument("--dpo-sampling", action="store_true")
        parser.add_argument("--output-file", type=str, default="translations.json")
        parser.add_argument("--pseudo-batch-size", type=int, default=1)

    def build_states(self) -> S2TAgentStates:
        return S2TAgentStates(stream_id=self.engine.open_stream())

    def update_multiplier(self, multiplier: int):  # reference :125-128
        self.latency_multiplier = multiplier
        self.max_new_tokens = 10 * multiplier

    # ------------------------------------------------------------------ per-chunk host logic
    def _prepare_speech(self, states: S2TAgentStates) -> np.ndarray:
        """reference :200-223 without the 399-sample first-chunk offset (the library owns that history) and
        without the bf16 cast (done on the device)."""
        seg = self.cfg.chunk_samples
        if len(states.source) > states.MAX_SRC_LEN:
            states.src_len -= len(states.source) - states.MAX_SRC_LEN
            states.source = states.source[-states.MAX_SRC_LEN:]
        source = np.asarray(states.source[states.src_len:], dtype=np.float32)
        if source.shape[0] % seg != 0:
            source = np.concatenate([source, np.zeros(seg - source.shape[0] % seg, dtype=np.float32)])
        states.src_len = len(states.source)
        return source

    def _gen_config(self) -> GenConfig:
        return GenConfig(latency_multiplier=self.latency_multiplier, max_new_tokens=self.max_new_tokens, beam=self.beam,
                         no_repeat_ngram_size=self.no_repeat_ngram_size,
                         no_repeat_ngram_lookback=self.no_repeat_ngram_lookback,
                         repetition_penalty=self.repetition_penalty, max_llm_cache_size=self.max_llm_cache_size,
                         always_cache_system_prompt=self.always_cache_system_prompt,
                         suppress_tokens=tuple(self.bad_words_ids))

    def policy(self, states: Optional[S2TAgentStates] = None):
        if states is None:
            states = self.states
        if states.source_sample_rate == 0:
            length_in_seconds = 0.0
        else:
            length_in_seconds = float(len(states.source)) / states.source_sample_rate
        if not states.source_finished and length_in_seconds < self.min_start_sec:  # :281-282
            return ReadAction()
        if states.source_finished and length_in_seconds < 0.32:  # :284-285
            return WriteAction(content="", finished=True)

        if states.needs_reset:  # new utterance: fresh speech cache / KV cache (:60-67)
            self.engine.reset_stream(states.stream_id)
            states.needs_reset = False
        first = not states.started
        speech = self._prepare_speech(states)
        if speech.shape[0] == 0:
            return ReadAction()
        input_ids = self.prompt_fn(first, self.latency_multiplier)
        encoder_input_ids = states.target_ids[-self.no_repeat_ngram_lookback:]  # :298-300
        if states.source_finished:
            states.segment_idx = -1
        pin = self.system_prompt_size if (first and self.always_cache_system_prompt) else 0
        outs, _ = self.engine.generate(self._gen_config(), [states.stream_id], [speech], [input_ids],
                                       [encoder_input_ids], system_prompt_size=pin)
        states.started = True
        generated = outs[0]

        # LLM-KV eviction by whole chunks (:337-361)
        cur = self.engine.stream_info(states.stream_id)["llm_cache_len"]
        self.cache_checkpoints.append(cur)
        if cur > self.max_llm_cache_size:
            new_size = 0
            for i, ckpt in enumerate(self.cache_checkpoints):
                new_size = cur - ckpt
                if new_size <= self.max_llm_cache_size:
                    self.cache_checkpoints = self.cache_checkpoints[i + 1:]
                    n_trimmed = ckpt
                    if self.always_cache_system_prompt:
                        n_trimmed -= self.system_prompt_size
                    self.cache_checkpoints = [c - n_trimmed for c in self.cache_checkpoints]
                    break
            self.engine.kv_evict(states.stream_id, new_size,
                                 self.system_prompt_size if self.always_cache_system_prompt else 0)

        output_ids = generated[:-1]  # outputs.sequences[0, len(prompt):-1]  (:363)
        states.target_ids.extend(output_ids)
        translation = self.decode_fn(output_ids).strip().replace("�", "")
        states.segment_idx += 1
        if translation != "" or states.source_finished:  # :389-395
            return WriteAction(content=translation, finished=states.source_finished)
        return ReadAction()


def default_args(**overrides) -> argparse.Namespace:
    """Parsed defaults of `InfiniSST.add_args` with the production values of scripts/infer/infinisst.sh:42-87."""
    parser = argparse.ArgumentParser()
    InfiniSST.add_args(parser)
    args = parser.parse_args([])
    prod = dict(block_size=48, max_cache_size=576, xpos=0, max_llm_cache_size=1000, always_cache_system_prompt=True,
                max_new_tokens=10, beam=1, no_repeat_ngram_lookback=100, no_repeat_ngram_size=5, repetition_penalty=1.2,
                latency_multiplier=1, max_latency_multiplier=4, min_start_sec=0.0, length_shrink_cfg="[(1024,2,2)] * 2")
    prod.update(overrides)
    for k, v in prod.items():
        setattr(args, k, v)
    return args


def feed_segments(agent: InfiniSST, audio: np.ndarray, segment_samples: int, sample_rate: int = 16000):
    """Minimal stand-in for SimulEval's evaluator loop: push `segment_samples` at a time, call policy,
    set source_finished on the last segment.  Returns the list of actions."""
    states = agent.build_states()
    states.source_sample_rate = sample_rate
    actions = []
    pos = 0
    n = len(audio)
    while pos < n:
        end = min(pos + segment_samples, n)
        states.source.extend(np.asarray(audio[pos:end], dtype=np.float32).tolist())
        pos = end
        states.source_finished = pos >= n
        act = agent.policy(states)
        if isinstance(act, WriteAction) and act.content:
            states.target.append(act.content)
        actions.append(act)
    return actions, states
