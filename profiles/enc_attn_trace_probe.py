"""Where do the encoder attention's 24 us go at one stream?  Steady-state chunks through the -DISST_ENC_TRACE build (make -C infinisst_amd/csrc trace); the stamps of
the LAST layer's launch of the last chunk are read: entry / queries rotated / scores written / softmax done / P.V done / output stored / V^T appended."""
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
gen = GenConfig(max_new_tokens=1, max_llm_cache_size=1000)
eng, weights, sys_n = bench.build_engine(cfg, 1, 1, dev, 1, None)
loop = bench.ChunkLoop(eng, cfg, gen, [0], sys_n); loop.import_steady_state(dev)
for _ in range(6): loop.step()
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
assert lib.isst_debug_enc_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(4096, 8)[:48].astype(np.int64)
t0 = t[:, 0].min()
names = ["entry", "queries rotated", "scores written (phase 1)", "softmax done (phase 2)", "P.V done (phase 3)", "output stored", "V^T appended / end"]
for i, n in enumerate(names):
    a = (t[:, i] - t0) / 100.0
    print(f"{n:28s} min {a.min():6.2f}  p50 {np.median(a):6.2f}  max {a.max():6.2f} us")
