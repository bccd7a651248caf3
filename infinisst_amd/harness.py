"""Tokenizer-driven prompts and a SimulEval-shaped evaluation loop that writes `instances.log`.

SURVEY section 8(f) row 3: closes the loop from the agent (agent.py) to the files the reference's evaluation consumes
(`simuleval ... --output $dir` -> `$dir/instances.log`, scored by StreamLAAL: reference README.md:101-125,
scripts/infer/infinisst.sh:49-87).  Nothing here touches the GPU; it is host logic around `InfiniSST.policy`.

* `ChatPrompt`      = reference agents/infinisst.py:225-268 (`_prepare_inputs`) against any HF-style tokenizer
                      (`apply_chat_template`, `eos_token_id`): the system prompt with the latency token on the first chunk,
                      `block_size // 4 * m` speech patch tokens in the user turn, an empty assistant turn, last token cut;
                      later chunks drop the 25 tokens of Llama-3.1's default system header (or overwrite token 0 with EOS for
                      Llama-3).
* `non_language_ids` = reference :142-148 (`--suppress-non-language`: every token whose decoded text has '(' or the
                      full-width '（').
* `attach_tokenizer` wires both plus `tokenizer.decode(ids, skip_special_tokens=True)` (:366) into an agent.
* `evaluate`        = what SimulEval's `SentenceLevelEvaluator` does for a speech-to-text agent [3P simuleval 1.1.x, not
                      installed here: restated from its documented instance format, parity unpinned]: feed
                      `source_segment_size` ms of audio per step, apply Read/Write actions, record for every emitted unit
                      (word or char) the source time consumed so far (`delays`) and that time plus the wall-clock compute
                      spent so far (`elapsed`, computation-aware), one JSON line per utterance in `instances.log`.
  The transformers / simuleval packages and the Llama tokenizer files are not in the build image; tests drive this module
  with a stub tokenizer, and the same code runs unchanged with `transformers.AutoTokenizer`.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass, field
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

DEFAULT_SPEECH_PATCH_TOKEN = "<sp_patch>"   # reference train/dataset.py:53
DEFAULT_LATENCY_TOKEN = "<latency_{}>"      # reference train/dataset.py:57
LLAMA31_SYSTEM_HEADER_TOKENS = 25           # reference agents/infinisst.py:264: `input_ids[:, 25:]`


def _ids(x) -> List[int]:
    """apply_chat_template may return a tensor / nested list (batch of 1) / flat list / a BatchEncoding (transformers >= 5)."""
    if hasattr(x, "keys") and "input_ids" in x:
        x = x["input_ids"]
    if hasattr(x, "tolist"):
        x = x.tolist()
    if len(x) > 0 and isinstance(x[0], (list, tuple)):
        x = x[0]
    return [int(t) for t in x]


class ChatPrompt:
    """`_prepare_inputs` of the reference (agents/infinisst.py:225-268)."""

    def __init__(self, tokenizer, source_lang: str, target_lang: str, block_size: int = 48, llama31: bool = True):
        self.tok = tokenizer
        self.source_lang, self.target_lang = source_lang, target_lang
        self.block_size = block_size
        self.llama31 = llama31
        self.system_prompt_size: Optional[int] = None

    def _template(self, messages) -> List[int]:
        return _ids(self.tok.apply_chat_template([messages], return_tensors=None, padding=True, truncation=False,
                                                 add_special_tokens=False))

    def system_message(self, multiplier: int) -> dict:
        latency_token = DEFAULT_LATENCY_TOKEN.format(multiplier)
        return {"role": "system",
                "content": f"Translate the following speech from {self.source_lang} to {self.target_lang} with latency {latency_token}."}

    def __call__(self, first: bool, multiplier: int) -> List[int]:
        messages = []
        if first:
            messages.append(self.system_message(multiplier))
            self.system_prompt_size = len(self._template(messages))  # :236-242
        messages.append({"role": "user", "content": self.block_size // 4 * multiplier * DEFAULT_SPEECH_PATCH_TOKEN})
        messages.append({"role": "assistant", "content": ""})
        ids = self._template(messages)[:-1]  # :256-262
        if not first:  # remove the system prompt, keep the last EOT (:263-267)
            if self.llama31:
                ids = ids[LLAMA31_SYSTEM_HEADER_TOKENS:]
            else:
                ids[0] = int(self.tok.eos_token_id)
        return ids


DEFAULT_SPEECH_START_TOKEN = "<sp_start>"   # reference train/dataset.py:54
DEFAULT_SPEECH_END_TOKEN = "<sp_end>"       # reference train/dataset.py:55
PAD_TOKEN = "<|finetune_right_pad_id|>"     # reference agents/infinisst.py:140


def preprocess_tokenizer(tokenizer, max_multiplier: int = 4) -> dict:
    """What `SpeechLlamaForCausalLM.preprocess(tokenizer, max_multiplier, resize=False)` does to the tokenizer and reads from it
    (reference model/llm.py:148-190): add `<sp_patch>`, `<sp_start>`, `<sp_end>`, `<latency_1..max>` as special tokens (and the pad
    token if it has no id) and look up the ids the speech splice keys on.  Returns the ModelConfig fields they determine."""
    tokenizer.add_tokens([DEFAULT_SPEECH_PATCH_TOKEN, DEFAULT_SPEECH_START_TOKEN, DEFAULT_SPEECH_END_TOKEN] +
                         [DEFAULT_LATENCY_TOKEN.format(i) for i in range(1, max_multiplier + 1)], special_tokens=True)
    if getattr(tokenizer, "pad_token_id", None) is None and getattr(tokenizer, "pad_token", None):
        tokenizer.add_tokens([tokenizer.pad_token], special_tokens=True)
    ids = tokenizer.convert_tokens_to_ids
    return {"sp_patch_id": int(ids(DEFAULT_SPEECH_PATCH_TOKEN)), "user_id": int(ids("user")), "assistant_id": int(ids("assistant")),
            "start_header_id": int(ids("<|start_header_id|>")), "vocab": len(tokenizer)}


def non_language_ids(tokenizer, bad_words: Sequence[str] = ("(", "（")) -> List[int]:
    """reference agents/infinisst.py:142-148."""
    out = []
    for idx in range(len(tokenizer)):
        text = tokenizer.decode(idx, skip_special_tokens=True)
        if any(b in text for b in bad_words):
            out.append(idx)
    return out


def attach_tokenizer(agent, tokenizer, llama31: bool = True, suppress_non_language: bool = False) -> ChatPrompt:
    """Replace the agent's synthetic prompt / decode hooks with the tokenizer's (what reference load_model +
    _prepare_inputs + :366 do).  The system prompt size becomes known at the first chunk, as in the reference."""
    prompt = ChatPrompt(tokenizer, agent.source_lang, agent.target_lang, agent.cfg.block_size, llama31)
    sys_ids = prompt._template([prompt.system_message(agent.latency_multiplier)])
    prompt.system_prompt_size = len(sys_ids)
    agent.system_prompt_size = len(sys_ids)

    def prompt_fn(first: bool, multiplier: int) -> List[int]:
        ids = prompt(first, multiplier)
        if first:
            agent.system_prompt_size = prompt.system_prompt_size
        return ids

    agent.prompt_fn = prompt_fn
    agent.decode_fn = lambda ids: tokenizer.decode(list(ids), skip_special_tokens=True)
    if suppress_non_language:
        agent.bad_words_ids = non_language_ids(tokenizer)
    return prompt


# ------------------------------------------------------------------------------------------------
# evaluation loop + instances.log
# ------------------------------------------------------------------------------------------------
@dataclass
class Instance:
    """One utterance, in the shape of a SimulEval instances.log line [3P]."""
    index: int
    source: str
    source_length: float  # ms
    reference: str = ""
    units: List[str] = field(default_factory=list)
    delays: List[float] = field(default_factory=list)    # ms of source read when the unit was emitted
    elapsed: List[float] = field(default_factory=list)   # delays + wall-clock compute so far (computation-aware)
    latency_unit: str = "word"

    @property
    def prediction(self) -> str:
        return ("" if self.latency_unit == "char" else " ").join(self.units)

    def to_json(self) -> str:
        return json.dumps({"index": self.index, "prediction": self.prediction, "delays": self.delays, "elapsed": self.elapsed,
                           "prediction_length": len(self.units), "reference": self.reference, "source": [self.source],
                           "source_length": self.source_length}, ensure_ascii=False)


def split_units(text: str, latency_unit: str) -> List[str]:
    """word: whitespace tokens; char: every non-space character (SimulEval --eval-latency-unit, infer/infinisst.sh:86)."""
    if latency_unit == "char":
        return [c for c in text if not c.isspace()]
    if latency_unit != "word":
        raise ValueError(f"latency unit {latency_unit!r}")
    return text.split()


def laal(delays: Sequence[float], source_length: float, reference_units: int) -> Optional[float]:
    """DIAGNOSTIC ONLY -- scoring stays with the reference's own StreamLAAL / SimulEval tooling on `instances.log`; this number is
    printed next to the log as a sanity check and is not a replacement for it.  Length-adaptive average lagging (Papi et al. 2022; the metric StreamLAAL segments and averages) in ms:
    (1/tau) sum_{i<=tau} d_i - (i-1) * |X| / max(|Y|, |Y*|),  tau = first unit emitted once the whole source was read."""
    n = len(delays)
    if n == 0:
        return None
    gamma = max(n, reference_units) / source_length if source_length > 0 else 0.0
    total, tau = 0.0, 0
    for i, d in enumerate(delays):
        total += d - (i / gamma if gamma > 0 else 0.0)
        tau = i + 1
        if d >= source_length:
            break
    return total / tau


def evaluate(agent, sources: Iterable[Tuple[str, np.ndarray]], references: Optional[Sequence[str]] = None, output_dir: Optional[str] = None,
             sample_rate: int = 16000, source_segment_size: Optional[float] = None, latency_unit: str = "word",
             clock: Callable[[], float] = time.perf_counter) -> List[Instance]:
    """Run the agent over `sources` = (name, mono float waveform) pairs the way SimulEval drives a speech-to-text agent
    and write `output_dir/instances.log` (+ `scores.json` with mean LAAL / computation-aware LAAL / real-time factor)."""
    seg_ms = float(source_segment_size if source_segment_size is not None else agent.source_segment_size)
    seg = int(round(seg_ms * sample_rate / 1000.0))
    instances: List[Instance] = []
    total_audio_s = total_wall_s = 0.0
    for index, (name, wav) in enumerate(sources):
        wav = np.asarray(wav, dtype=np.float32).reshape(-1)
        inst = Instance(index=index, source=name, source_length=1000.0 * wav.shape[0] / sample_rate,
                        reference=(references[index] if references is not None else ""), latency_unit=latency_unit)
        states = agent.states
        states.reset()
        states.source_sample_rate = sample_rate
        compute_ms, pos = 0.0, 0
        while True:
            nxt = min(pos + seg, wav.shape[0])
            states.source.extend(wav[pos:nxt].tolist())
            pos = nxt
            states.source_finished = pos >= wav.shape[0]
            t0 = clock()
            action = agent.policy(states)
            compute_ms += 1000.0 * (clock() - t0)
            content = getattr(action, "content", None)
            if content is not None:  # WriteAction
                read_ms = 1000.0 * pos / sample_rate
                new_units = split_units(content, latency_unit)
                states.target.extend(new_units)
                for u in new_units:
                    inst.units.append(u)
                    inst.delays.append(read_ms)
                    inst.elapsed.append(read_ms + compute_ms)
                if getattr(action, "finished", False):
                    break
            if states.source_finished and content is None:
                break  # a ReadAction after the end of the source cannot make progress
        total_audio_s += wav.shape[0] / sample_rate
        total_wall_s += compute_ms / 1000.0
        instances.append(inst)
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "instances.log"), "w", encoding="utf-8") as f:
            for inst in instances:
                f.write(inst.to_json() + "\n")
        ref_units = [len(split_units(i.reference, latency_unit)) for i in instances]
        l1 = [laal(i.delays, i.source_length, r) for i, r in zip(instances, ref_units)]
        l2 = [laal(i.elapsed, i.source_length, r) for i, r in zip(instances, ref_units)]
        mean = lambda xs: (sum(x for x in xs if x is not None) / max(1, sum(x is not None for x in xs)))
        with open(os.path.join(output_dir, "scores.json"), "w", encoding="utf-8") as f:
            json.dump({"LAAL_ms": mean(l1), "LAAL_CA_ms": mean(l2), "instances": len(instances),
                       "RTF": (total_wall_s / total_audio_s if total_audio_s > 0 else None), "latency_unit": latency_unit}, f)
    return instances
