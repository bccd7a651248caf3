"""Where the host time of a 64-stream step goes (Python + ctypes around isst_generate): cProfile over steady-state steps on the real library."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from infinisst_amd.config import GenConfig, full_config
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = full_config().replace(eos_ids=())
dev = torch.device("cuda:0")
gen = GenConfig(max_new_tokens=10, max_llm_cache_size=1000)
eng, weights, sys_n = bench.build_engine(cfg, NS, 10, dev, 1, None)
loop = bench.ChunkLoop(eng, cfg, gen, list(range(NS)), sys_n); loop.import_steady_state(dev)
for _ in range(6): loop.step()
dt, lat, host = bench.timed_steps(loop, 12)
print(f"{NS} streams: {1e3 * dt / 12:.3f} ms per step, host {1e3 * host / 12:.3f} ms per step")
pr = cProfile.Profile(); pr.enable()
for _ in range(12): loop.step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
