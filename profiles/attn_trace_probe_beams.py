"""Where do the microseconds of the batched decode attention go at 64 streams x 4 beams (llm_attn_partial_kernel<4,1,true>, folded shared-prefix
form: 512 workgroups, wave w walks ~16 prefix tiles and the own tiles of beam w)?  -DISST_ATTN_TRACE build; stamps of the LAST launch.
argv: streams beams"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from infinisst_amd import engine as E
here = os.path.dirname(os.path.abspath(__file__))
lib = E.load_library(os.path.join(here, "..", "infinisst_amd", "libinfinisst_hip_trace.so")); E._lib = lib
import bench
from infinisst_amd.config import GenConfig, full_config
n, B = int(sys.argv[1]), int(sys.argv[2])
cfg = full_config().replace(eos_ids=())
dev = torch.device("cuda:0")
gen = GenConfig(max_new_tokens=10, max_llm_cache_size=1000, beam=B)
eng, weights, sys_n = bench.build_engine(cfg, n, 10, dev, B, None)
loop = bench.ChunkLoop(eng, cfg, gen, list(range(n)), sys_n); loop.import_steady_state(dev)
for _ in range(2): loop.step()
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
assert lib.isst_debug_attn_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(8192, 8).astype(np.int64)
t = t[t[:, 0] > 0]
t = t[t[:, 0] > t[:, 0].max() - 20000]  # the last launch: entries within 200 us of the latest entry
t0 = t[:, 0].min()
print(f"{n} streams x {B} beams: {len(t)} workgroups in the last launch")
for i, nm in enumerate(["entry", "queries rotated", "tiles done", "output / slab stored"]):
    a = (t[:, i] - t0) / 100.0
    print(f"{nm:24s} min {a.min():7.2f}  p10 {np.percentile(a, 10):7.2f}  p50 {np.median(a):7.2f}  p90 {np.percentile(a, 90):7.2f}  max {a.max():7.2f} us")
d = (t[:, 2] - t[:, 1]) / 100.0
print(f"tile loop per workgroup   min {d.min():7.2f}  p50 {np.median(d):7.2f}  max {d.max():7.2f} us")
