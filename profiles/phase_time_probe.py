"""Wall time of the parts of a one-stream steady-state chunk WITHOUT a profiler attached (rocprofv3 inflates every launch issued outside a graph):
encoder alone (isst_encode_speech), a whole chunk with --gen-tokens 1 (encoder + prefill + one sampling tail), and whole chunks with 2 / 10
tokens (each further decode pass is one graph replay)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from infinisst_amd import synth
from infinisst_amd.config import GenConfig, full_config
dev = torch.device("cuda:0")
cfg = full_config().replace(eos_ids=())
weights = None
def run(gen_tokens, steps=24):
    global weights
    gen = GenConfig(max_new_tokens=gen_tokens, max_llm_cache_size=1000)
    eng, weights, sys_n = bench.build_engine(cfg, 1, gen_tokens, dev, 1, weights)
    loop = bench.ChunkLoop(eng, cfg, gen, [0], sys_n)
    loop.import_steady_state(dev)
    for _ in range(4): loop.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): loop.step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
    # encoder alone on the same engine
    sid = loop.sids[0]
    seg = loop.audio[0][:cfg.chunk_samples]
    for _ in range(3): eng.encode_speech(sid, seg)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.encode_speech(sid, seg)
    torch.cuda.synchronize(); de = (time.perf_counter() - t0) / steps * 1e3
    del eng
    return dt, de
for g in (1, 2, 10):
    dt, de = run(g)
    print(f"gen tokens {g:2d}: chunk {dt:7.3f} ms   (encoder alone {de:6.3f} ms)", flush=True)
