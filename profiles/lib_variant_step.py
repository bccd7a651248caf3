"""A/B aid: the steady-state step of `streams` concurrent streams through ANOTHER build of the library (path given), e.g. one compiled with a -D switch
of a kernel under study.  usage: python3 profiles/lib_variant_step.py <lib.so> [streams] [steps]   (under rocprofv3 --kernel-trace --stats for per-kernel times)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infinisst_amd import engine as E
lib = E.load_library(os.path.abspath(sys.argv[1])); E._lib = lib
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 12
if os.environ.get("AB_ATTN_WGS"):  # decode attention: target number of workgroups (splits per (stream, kv head) follow from it)
    lib.isst_op_set_attn_tuning(int(os.environ["AB_ATTN_WGS"]))
if os.environ.get("AB_GEMM_TUNE"):  # "a,b" -> isst_op_set_gemm_tuning(a, b) (gemm.hip gemm_set_tuning: e.g. "0,128" = gemm_mid's four-chunk ring)
    lib.isst_op_set_gemm_tuning(*[int(v) for v in os.environ["AB_GEMM_TUNE"].split(",")])
import bench
from infinisst_amd.config import GenConfig, full_config
cfg = full_config().replace(eos_ids=())
dev = torch.device("cuda:0")
B = int(os.environ.get("AB_BEAM", "1"))  # num_beams (1 = greedy)
gen = GenConfig(max_new_tokens=10, max_llm_cache_size=1000, beam=B)
eng, weights, sys_n = bench.build_engine(cfg, NS, 10, dev, B, None)
loop = bench.ChunkLoop(eng, cfg, gen, list(range(NS)), sys_n); loop.import_steady_state(dev)
for _ in range(6): loop.step()
dt, lat, _ = bench.timed_steps(loop, STEPS)
print(f"{os.path.basename(sys.argv[1])}: {NS} streams x {B} beams {1e3 * dt / STEPS:.3f} ms per step, p50 {1e3 * float(np.percentile(lat, 50)):.3f}")
