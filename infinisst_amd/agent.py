"""SimulEval-shaped agent over the MI355X engine: the drop-in for reference agents/infinisst.py.

Same surface as the reference's `InfiniSST(SpeechToTextAgent)`: `add_args(parser)`, `__init__(args)`,
`build_states()`, `policy(states) -> ReadAction | WriteAction(content, finished)`, `update_multiplier(m)`,
flag names and defaults of agents/options.py and agents/infinisst.py:185-198.  SimulEval itself is a third-party
package that is not installed here, so minimal `ReadAction` / `WriteAction` / `AgentStates` stand-ins are defined
below; when simuleval is importable its classes (and `@entrypoint`) are used instead, so `simuleval --agent infinisst_amd/agent.py
--model-name <llama dir> --state-dict-path <pytorch_model.bin> ...` constructs `InfiniSST(args)` exactly as it constructs the
reference agent: `__init__(args)` -> `load_model(args)` (tokenizer, bad-word scan, geometry, engine, checkpoint).

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

import ast
import json
import logging
import os

from . import synth
from .config import GenConfig, ModelConfig, full_config
from .engine import Engine
from .streams import effective_new_cache_size, evict_whole_chunks  # noqa: F401  (effective_new_cache_size: re-exported for tests)

logger = logging.getLogger(__name__)

try:  # pragma: no cover - simuleval is absent in the build image
    from simuleval.agents import SpeechToTextAgent as _AgentBase
    from simuleval.agents.actions import ReadAction, WriteAction
    from simuleval.agents.states import AgentStates as _StatesBase
    from simuleval.utils import entrypoint
    HAVE_SIMULEVAL = True
except Exception:  # stand-ins with the fields the policy touches
    HAVE_SIMULEVAL = False

    def entrypoint(cls):  # simuleval.utils.entrypoint marks the class `simuleval --agent <file>` instantiates
        return cls

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


def parse_length_shrink_cfg(text: Optional[str]):
    """`--length-shrink-cfg "[(1024,2,2)] * 2"` (reference model/speech_encoder.py:117, fairseq `eval_str_dict`) -> list of
    (dim, kernel, stride) without `eval`: literals, list repetition and concatenation only."""
    if text is None:
        return None

    def ev(node):
        if isinstance(node, ast.Expression):
            return ev(node.body)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Mult):
            return ev(node.left) * ev(node.right)
        if isinstance(node, ast.BinOp) and isinstance(node.op, ast.Add):
            return ev(node.left) + ev(node.right)
        return ast.literal_eval(node)
    layers = ev(ast.parse(text.strip(), mode="eval"))
    return [tuple(int(x) for x in layer) for layer in layers]


@entrypoint
class InfiniSST(_AgentBase):
    """MI355X-native InfiniSST agent: `InfiniSST(args)` is all SimulEval needs (reference agents/infinisst.py:69-113).
    `--beam 1` = greedy (the reference asserts beam > 1, :86), `--beam N` = the reference's beam search."""

    def __init__(self, args, engine: Optional[Engine] = None, model_cfg: Optional[ModelConfig] = None,
                 prompt_fn: Optional[Callable[[bool, int], List[int]]] = None,
                 decode_fn: Optional[Callable[[Sequence[int]], str]] = None, system_prompt_size: Optional[int] = None,
                 weights=None, tokenizer=None):
        """Only `args` is part of the reference surface.  The keyword arguments are injection points for tests and benchmarks
        (a ready engine, synthetic prompts, a weight dict instead of `--state-dict-path`, a tokenizer object instead of
        `--model-name`)."""
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
        self.suppress_non_language = getattr(args, "suppress_non_language", False)
        self.max_new_tokens = args.max_new_tokens
        # the sample branch (reference :311-315): warpers + draw in the library (csrc/warp.hip); the draws come from a counter-based generator, so they
        # are reproducible but not torch.multinomial's.  With --beam > 1 this is HF's beam sample (patch_hf.py:871-875): 2 x beams draws without replacement
        # from the softmax over all beams' warped scores instead of their top-k
        self.do_sample = bool(getattr(args, "do_sample", False))
        self.top_p, self.top_k = float(getattr(args, "top_p", 1.0)), int(getattr(args, "top_k", 0))
        self.epsilon_cutoff, self.temperature = float(getattr(args, "epsilon_cutoff", 0.0)), float(getattr(args, "temperature", 1.0))
        self.pseudo_batch_size = getattr(args, "pseudo_batch_size", 1)  # accepted, not used: see add_args
        self.dpo_sampling = getattr(args, "dpo_sampling", False)         # reference :108-110
        self.output_file = getattr(args, "output_file", "translations.json")
        self.max_llm_cache_size = args.max_llm_cache_size
        self.always_cache_system_prompt = args.always_cache_system_prompt
        self.cache_checkpoints: List[int] = []  # agent-level, not reset per utterance (reference :106)
        self.bad_words_ids: List[int] = list(getattr(args, "bad_words_ids", []) or [])
        self.tokenizer = None
        injected = engine is not None or weights is not None or model_cfg is not None
        if injected:
            self.cfg = model_cfg or full_config().replace(block_size=args.block_size, max_cache_size=args.max_cache_size)
            self.prompt_fn = prompt_fn or (lambda first, m: synth.chunk_prompt_ids(self.cfg, m, first))
            self.decode_fn = decode_fn or (lambda ids: " ".join(str(i) for i in ids))
            self.system_prompt_size = system_prompt_size if system_prompt_size is not None else len(
                synth.system_prompt_ids(self.cfg, self.latency_multiplier))
            if engine is None:
                if weights is None:
                    raise ValueError("model_cfg without an engine needs a weight dict")
                engine = self._new_engine(self.cfg, self.system_prompt_size, self.system_prompt_size + 16 + 12 * self.max_latency_multiplier)
                engine.load_weights(weights)
            self.engine = engine
            if tokenizer is not None:
                self._attach(tokenizer, llama31=True)
        else:
            self.load_model(args, tokenizer=tokenizer)
        super().__init__(args)

    # ------------------------------------------------------------------ load_model (reference :130-183)
    def _new_engine(self, cfg: ModelConfig, max_system_prompt: int, max_prompt_len: int) -> Engine:
        return Engine(cfg, max_streams=1, max_multiplier=self.max_latency_multiplier, max_prompt_len=max_prompt_len,
                      max_new_tokens=max(self.max_new_tokens, 10 * self.max_latency_multiplier),
                      max_llm_cache_size=self.max_llm_cache_size, max_system_prompt=max_system_prompt, max_beams=self.beam)

    def _attach(self, tokenizer, llama31: bool):
        from . import harness
        self.tokenizer = tokenizer
        return harness.attach_tokenizer(self, tokenizer, llama31=llama31, suppress_non_language=self.suppress_non_language)

    @staticmethod
    def _llama_side_config(model_name: str) -> dict:
        """EOS ids / rotary parameters from `<model_name>/generation_config.json` and `config.json` when `--model-name` is a local
        directory (the reference reads them through from_pretrained, :150-154); otherwise `load_model` keeps the Llama-3.1 values for a
        Llama-3.1 model name and falls back to the tokenizer's EOS for anything else."""
        out = {}
        if model_name and os.path.isdir(model_name):
            gpath, cpath = os.path.join(model_name, "generation_config.json"), os.path.join(model_name, "config.json")
            if os.path.exists(gpath):
                with open(gpath) as f:
                    eos = json.load(f).get("eos_token_id")
                if eos is not None:
                    out["eos_ids"] = tuple(int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos]))
            if os.path.exists(cpath):
                with open(cpath) as f:
                    c = json.load(f)
                if "rms_norm_eps" in c:
                    out["rms_eps"] = float(c["rms_norm_eps"])
                rp = c.get("rope_parameters") or c.get("rope_scaling") or {}
                theta = c.get("rope_theta", rp.get("rope_theta"))
                if theta is not None:
                    out["rope_theta"] = float(theta)
                if rp.get("rope_type", rp.get("type")) == "llama3":
                    out.update(rope_factor=float(rp["factor"]), rope_low_freq_factor=float(rp["low_freq_factor"]),
                               rope_high_freq_factor=float(rp["high_freq_factor"]),
                               rope_original_max_pos=int(rp["original_max_position_embeddings"]))
                if "eos_ids" not in out and c.get("eos_token_id") is not None:
                    eos = c["eos_token_id"]
                    out["eos_ids"] = tuple(int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos]))
        return out

    def load_model(self, args, tokenizer=None):
        """reference agents/infinisst.py:130-183 on the MI355X engine: tokenizer (+ the speech / latency tokens `preprocess` adds),
        the `--suppress-non-language` scan, model geometry, engine, `--state-dict-path`, chat-template prompts.

        * `--rope 1 --xpos 0` is the production setting (scripts/infer/infinisst.sh:74); `--rope 0` (absolute sinusoid positions, the reference's own
          arithmetic: patch_speech_encoder.py:448-461, :488-493) runs too, whatever `--xpos` says (the rotary module is built and never called, :823);
          `--rope 1 --xpos 1` is refused: the xpos scaling lives in the un-vendored rotary_embedding_torch and cannot be pinned here.
          `--w2v2-type` must be w2v2 (:168-171).
        * `--w2v2-path` / `--ctc-finetuned` only decide the fairseq architecture the reference instantiates before
          `load_state_dict` overwrites every weight (:155-180); here the geometry is read from the state dict itself
          (`checkpoint.infer_config`), so they are accepted and unused."""
        from . import checkpoint, harness
        rotary = bool(int(getattr(args, "rope", 1)))
        if rotary and int(getattr(args, "xpos", 0)) != 0:
            raise NotImplementedError("--rope 1 --xpos 1 is not implemented (xpos scaling of the un-vendored rotary_embedding_torch); "
                                      "--xpos 0 (the production setting) and --rope 0 are")
        if getattr(args, "w2v2_type", None) not in (None, "w2v2"):
            raise ValueError(f"Unsupported type: {args.w2v2_type}")  # reference :171
        if not getattr(args, "state_dict_path", None):
            raise ValueError("--state-dict-path is required (there is no model without the checkpoint)")
        if tokenizer is None:
            import transformers  # the reference's own dependency; fails loudly when absent
            tokenizer = transformers.AutoTokenizer.from_pretrained(args.model_name, padding_side="right", use_fast=False)
        tokenizer.pad_token = harness.PAD_TOKEN  # :140
        ids = harness.preprocess_tokenizer(tokenizer, self.max_latency_multiplier)  # model.preprocess(...), :174

        state = checkpoint.load_state_dict_file(args.state_dict_path)  # :179
        side = self._llama_side_config(args.model_name)
        if "eos_ids" not in side and "3.1" not in str(args.model_name) and getattr(tokenizer, "eos_token_id", None) is not None:
            # no generation_config.json / config.json at hand (a hub id): Llama-3.1 keeps ModelConfig's three stop ids (what from_pretrained reads
            # from its generation_config.json, agents/infinisst.py:150-154); any other model falls back to the tokenizer's EOS
            side["eos_ids"] = (int(tokenizer.eos_token_id),)
        shrink = parse_length_shrink_cfg(getattr(args, "length_shrink_cfg", None))
        cfg = checkpoint.infer_config(state, block_size=args.block_size, max_cache_size=args.max_cache_size,
                                      sp_patch_id=ids["sp_patch_id"], user_id=ids["user_id"], assistant_id=ids["assistant_id"],
                                      start_header_id=ids["start_header_id"], **side)
        if shrink is not None and [tuple(x) for x in cfg.shrink_layers] != shrink:
            raise ValueError(f"--length-shrink-cfg {shrink} does not match the checkpoint's length_shrink convs {cfg.shrink_layers}")
        if cfg.vocab != ids["vocab"]:  # load_state_dict would fail on the embedding shape (:180)
            raise ValueError(f"checkpoint vocabulary {cfg.vocab} != tokenizer + speech/latency tokens {ids['vocab']}")
        # the one restated third-party switch that changes every encoder output (DESIGN.md section 4 "parity unpinned": rotary_embedding_torch after
        # `.to(bf16)`): say which reading is in use, and let a maintainer with the real checkpoint flip it without touching code
        mode = os.environ.get("INFINISST_ENC_ROPE_MODE")
        if mode:
            cfg = cfg.replace(enc_rope_mode=mode)
        cfg = cfg.replace(enc_rope=rotary)
        if rotary:
            logger.warning("speech-encoder rotary arithmetic: enc_rope_mode=%s (INFINISST_ENC_ROPE_MODE=bf16|fp32 switches it; how to decide: "
                           "DESIGN.md section 4)", cfg.enc_rope_mode)
        else:
            logger.warning("speech encoder without rotary positions (--rope 0): bf16 sinusoid of the stream position added to the encoder input")
        self.cfg = cfg
        self.llama31 = "3.1" in str(args.model_name)  # :183
        self.decode_fn = self.prompt_fn = None
        self.system_prompt_size = 0
        prompt = self._attach(tokenizer, self.llama31)
        longest = max(len(prompt(True, m)) for m in range(1, self.max_latency_multiplier + 1))
        weights, inv_freq, skipped = checkpoint.split_state_dict(cfg, state)
        del state
        self.engine = self._new_engine(cfg, max_system_prompt=prompt.system_prompt_size, max_prompt_len=longest + 8)
        self.engine.load_weights(weights, enc_inv_freq=inv_freq)
        logger.info("loaded %d tensors from %s (%d keys outside the hot path skipped)", len(weights), args.state_dict_path, len(skipped))

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
        parser.add_argument("--pseudo-batch-size", type=int, default=1,
                            help="accepted for CLI compatibility and ignored: the reference replicates ONE stream N times to imitate "
                                 "batching (agents/infinisst.py:291-301); this engine batches real streams (isst_generate with n > 1)")

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
                         suppress_tokens=tuple(self.bad_words_ids), do_sample=self.do_sample, temperature=self.temperature, top_k=self.top_k,
                         top_p=self.top_p, epsilon_cutoff=self.epsilon_cutoff)

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
            # no new samples since the last call (the reference would hand an empty tensor to the conv stack and fail): after the
            # end of the source that is the evaluator's final step -- e.g. an utterance whose length is an exact multiple of the
            # segment size -- and the instance has to be closed; before it there is simply nothing to do yet
            if states.source_finished:
                return WriteAction(content="", finished=True)
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

        # LLM-KV eviction by whole chunks (:337-361); the walk itself is shared with the multi-stream driver (streams.StreamBatch)
        cur = self.engine.stream_info(states.stream_id)["llm_cache_len"]
        keep = self.system_prompt_size if self.always_cache_system_prompt else 0
        self.cache_checkpoints, new_size = evict_whole_chunks(self.cache_checkpoints, cur, self.max_llm_cache_size, keep)
        if new_size is not None:
            self.engine.kv_evict(states.stream_id, new_size, keep)

        output_ids = generated[:-1]  # outputs.sequences[0, len(prompt):-1]  (:363)
        states.target_ids.extend(output_ids)
        translation = self.decode_fn(output_ids).strip().replace("�", "")
        if self.dpo_sampling:  # reference :369-382: per-chunk translations of an utterance, one bracketed line per utterance
            states.translations_list.append(f"'{translation}'" if translation else "''")
            if states.source_finished:
                try:
                    with open(self.output_file, "a", encoding="utf-8") as f:
                        f.write(f"[{', '.join(states.translations_list)}]" + "\n")
                    states.translations_list = []
                except Exception as e:  # the reference prints and carries on
                    print(f"Error writing translations to file: {e}")
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
