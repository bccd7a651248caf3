"""Helper of tests/test_simuleval_surface.py (run as a subprocess): installs a stand-in `simuleval` package -- the class surface the reference's agent
imports (agents/infinisst.py:1-25: simuleval.utils.entrypoint, simuleval.agents.SpeechToTextAgent, simuleval.agents.actions.ReadAction / WriteAction,
simuleval.agents.states.AgentStates) -- BEFORE infinisst_amd.agent is imported, so that the module's `HAVE_SIMULEVAL = True` branch is the one that
runs; then builds the agent from parsed flags (recording engine: no GPU) and steps it the way SimulEval's evaluator does.  Prints one JSON line."""
import json
import os
import sys
import types

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(__file__))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class AgentStates:  # simuleval.agents.states.AgentStates (restated from its documented surface; the reference subclasses it, :50-67)
    def __init__(self):
        self.reset()

    def reset(self):
        self.source, self.target = [], []
        self.source_finished, self.target_finished = False, False
        self.source_sample_rate = 0

    def update_source(self, segment):  # SpeechSegment: content = samples, sample_rate, finished
        self.source_finished = segment.finished
        self.source_sample_rate = segment.sample_rate
        self.source += list(segment.content)

    def update_target(self, segment):
        self.target_finished = segment.finished
        if segment.content:
            self.target.append(segment.content)


class SpeechToTextAgent:  # simuleval.agents.SpeechToTextAgent: GenericAgent.__init__(args) keeps the flags and builds the states
    source_type, target_type = "speech", "text"

    def __init__(self, args=None):
        self.args = args
        self.states = self.build_states()

    def build_states(self):
        return AgentStates()

    def reset(self):
        self.states.reset()


class Action:
    pass


class ReadAction(Action):
    pass


class WriteAction(Action):
    def __init__(self, content, finished):
        self.content, self.finished = content, finished


class Segment:
    def __init__(self, content, finished, sample_rate=16000):
        self.content, self.finished, self.sample_rate = content, finished, sample_rate


ENTRY = []


def entrypoint(cls):
    ENTRY.append(cls.__name__)
    return cls


_mod("simuleval").__path__ = []
_mod("simuleval.agents", SpeechToTextAgent=SpeechToTextAgent).__path__ = []
_mod("simuleval.agents.states", AgentStates=AgentStates)
_mod("simuleval.agents.actions", ReadAction=ReadAction, WriteAction=WriteAction)
_mod("simuleval.utils", entrypoint=entrypoint)

import argparse  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import infinisst_amd.agent as A  # noqa: E402
from infinisst_amd import synth  # noqa: E402
from infinisst_amd.config import toy_config  # noqa: E402
from test_abi_and_host import _RecordingEngine  # noqa: E402
from tiny_tokenizer import build_tokenizer_dir  # noqa: E402


def main(tmp):
    out = {"have_simuleval": A.HAVE_SIMULEVAL, "entrypoint_saw": ENTRY, "agent_base": A.InfiniSST.__mro__[1].__module__ + "." + A.InfiniSST.__mro__[1].__name__,
           "states_is_simuleval_states": issubclass(A.S2TAgentStates, AgentStates), "write_action_is_simuleval": A.WriteAction is WriteAction,
           "read_action_is_simuleval": A.ReadAction is ReadAction}
    cfg = toy_config()
    model_dir = build_tokenizer_dir(tmp, cfg)
    w = synth.random_weights(cfg, dtype=torch.float32, seed=9)
    ckpt = os.path.join(tmp, "pytorch_model.bin")
    torch.save({"model." + k: v for k, v in w.items()}, ckpt)
    A.Engine = _RecordingEngine
    parser = argparse.ArgumentParser()
    A.InfiniSST.add_args(parser)
    args = parser.parse_args(["--model-name", model_dir, "--state-dict-path", ckpt, "--w2v2-type", "w2v2", "--length-shrink-cfg", "[(128,2,2)] * 2",
                              "--block-size", "48", "--max-cache-size", "576", "--xpos", "0", "--max-llm-cache-size", "150", "--always-cache-system-prompt",
                              "--max-new-tokens", "10", "--beam", "4", "--latency-multiplier", "1", "--min-start-sec", "0"])
    agent = A.InfiniSST(args)
    out["agent_args_kept_by_base"] = agent.args is args
    states = agent.build_states()  # what SimulEval's GenericAgent does per instance
    out["built_states_class"] = type(states).__name__
    audio = synth.synthetic_audio(cfg.chunk_samples * 3)
    acts = []
    for k in range(3):  # the evaluator loop: push a segment, ask the policy, feed a WRITE back
        seg = Segment(np.asarray(audio[k * cfg.chunk_samples:(k + 1) * cfg.chunk_samples], dtype=np.float32).tolist(), finished=(k == 2))
        states.update_source(seg)
        act = agent.policy(states)
        acts.append(type(act).__name__ + ("/finished" if getattr(act, "finished", False) else ""))
        if isinstance(act, WriteAction):
            states.update_target(Segment(act.content, act.finished))
    out["actions"] = acts
    out["engine_calls"] = len(_RecordingEngine.instances[-1].calls)
    states.reset()
    out["reset_clears_source"] = len(states.source) == 0
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
