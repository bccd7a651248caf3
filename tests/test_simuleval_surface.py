"""The agent through the SimulEval CLASS surface (no GPU): with a `simuleval` package importable, infinisst_amd.agent must subclass ITS
SpeechToTextAgent / AgentStates, return ITS ReadAction / WriteAction and be marked by ITS entrypoint -- the `HAVE_SIMULEVAL = True` branch of
agent.py:36-41, which no other test reaches because simuleval is absent from the image.  Runs in a subprocess: the stand-in package has to be in
sys.modules before the first import of the agent module."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_agent_runs_on_the_simuleval_class_surface(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(HERE, "simuleval_stub_driver.py"), str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["have_simuleval"] is True
    assert j["entrypoint_saw"] == ["InfiniSST"]                      # @entrypoint is simuleval's, applied once, to the agent class
    assert j["agent_base"].endswith("SpeechToTextAgent") and not j["agent_base"].startswith("infinisst_amd")
    assert j["states_is_simuleval_states"] and j["write_action_is_simuleval"] and j["read_action_is_simuleval"]
    assert j["agent_args_kept_by_base"] and j["built_states_class"] == "S2TAgentStates"
    assert len(j["actions"]) == 3 and j["actions"][-1] == "WriteAction/finished"
    assert all(a.split("/")[0] in ("ReadAction", "WriteAction") for a in j["actions"])
    assert j["engine_calls"] >= 1 and j["reset_clears_source"]
