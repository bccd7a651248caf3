"""Bisecting aid for the fused attention + combine + o_proj launch: the same 12 toy-size chunks (free-running, evictions) through engines created with
ISST_FUSE_ATTN_OPROJ = 1 / 0 / 1 / 2 -- fused against three launches, fused against itself (a race would differ from run to run), and fused attention +
combine with a separate o_proj (which half differs?).  This is how the 1-ulp differences from compiler-chosen fma contraction were found."""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from infinisst_amd import synth
from infinisst_amd.config import GenConfig
from test_gpu_engine import toy_config, make_engine
from oracle import agent as oag
cfg = toy_config(); gen = GenConfig(max_new_tokens=8, max_llm_cache_size=300)
w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, norm_jitter=0.05, seed=72)
audio = synth.synthetic_audio(cfg.chunk_samples * 12, stream_id=6); sys_n = len(synth.system_prompt_ids(cfg))
def run(flag):
    os.environ["ISST_FUSE_ATTN_OPROJ"] = flag
    eng = make_engine(cfg, w, debug_taps=False, max_llm_cache_size=300, max_streams=1); sid = eng.open_stream(); logs = []; toks = []; ckpts = []
    for c in range(12):
        seg = audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples]
        o, l = eng.generate(gen, [sid], [seg], [synth.chunk_prompt_ids(cfg, 1, first=(c == 0))], [[]], system_prompt_size=sys_n if c == 0 else 0, return_logits=True, forced_tokens=FORCED)
        logs.append(l[0][:len(o[0])].copy()); toks.append(o[0])
        cur = eng.stream_info(sid)["llm_cache_len"]; ckpts.append(cur)
        ev = oag.evict(ckpts, cur, 150, True, sys_n)
        if ev is not None:
            ckpts, new_size = ev; eng.kv_evict(sid, new_size, sys_n)
    eng.close(); print(flag, [len(t) for t in toks], flush=True); return logs
FORCED = None
a1 = run("1"); a0 = run("0"); a1b = run("1"); a2 = run("2")
def cmp(xs, ys, n):
    for c, (x, y) in enumerate(zip(xs, ys)):
        if x.shape != y.shape: print(n, "chunk", c, "shapes", x.shape, y.shape); return
        if not np.array_equal(x, y): print(n, "chunk", c, "DIFF rows", sorted(set(np.argwhere(x != y)[:, 0].tolist())), "max", np.abs(x - y).max()); return
    print(n, "equal")
cmp(a0, a1, "three-launch vs fused"); cmp(a1, a1b, "fused vs fused again"); cmp(a0, a2, "three-launch vs fused attention+combine, separate o_proj")
