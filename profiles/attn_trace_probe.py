"""Where do the decode attention's microseconds go at one stream (llm_attn_partial_kernel<4,1,false>: 136 workgroups, one 16-key tile per wave)?
Steady-state chunks through the -DISST_ATTN_TRACE build; stamps of the LAST launch (last layer of the last decode pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from infinisst_amd import engine as E
here = os.path.dirname(os.path.abspath(__file__))
lib = E.load_library(os.path.join(here, "..", "infinisst_amd", "libinfinisst_hip_trace.so")); E._lib = lib
import bench
from infinisst_amd.config import GenConfig, full_config
cfg = full_config().replace(eos_ids=())
dev = torch.device("cuda:0")
gen = GenConfig(max_new_tokens=4, max_llm_cache_size=1000)
eng, weights, sys_n = bench.build_engine(cfg, 1, 4, dev, 1, None)
loop = bench.ChunkLoop(eng, cfg, gen, [0], sys_n); loop.import_steady_state(dev)
for _ in range(4): loop.step()
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
assert lib.isst_debug_attn_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(8192, 8).astype(np.int64)
live = t[:, 0] > 0
t = t[live]
# keep the workgroups of the last launch: entries within 50 us of the latest entry
t = t[t[:, 0] > t[:, 0].max() - 5000]
t0 = t[:, 0].min()
print(f"{len(t)} workgroups in the last launch")
for i, n in enumerate(["entry", "queries rotated", "tiles done (keys, values, QK, softmax, PV)", "slab stored"]):
    a = (t[:, i] - t0) / 100.0
    print(f"{n:44s} min {a.min():6.2f}  p50 {np.median(a):6.2f}  max {a.max():6.2f} us")
