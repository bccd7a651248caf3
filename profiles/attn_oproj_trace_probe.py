"""Timeline of the fused attention + combine + o_proj launch at one stream (llm_attn_oproj_kernel<4,16>: 256 workgroups, one per CU), through the
-DISST_ATTN_TRACE build (make trace): thread 0 of every workgroup stamps the 100 MHz wall clock.  Stamps of the LAST launch (last layer, last pass)."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # argv: num_beams (1 = greedy)
gen = GenConfig(max_new_tokens=4, max_llm_cache_size=1000, beam=B)
eng, weights, sys_n = bench.build_engine(cfg, 1, 4, dev, B, None)
loop = bench.ChunkLoop(eng, cfg, gen, [0], sys_n); loop.import_steady_state(dev)
for _ in range(4): loop.step()
torch.cuda.synchronize()
buf = np.zeros(8192 * 8, dtype=np.uint64)
assert lib.isst_debug_attn_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(8192, 8).astype(np.int64)[:256]
t0 = t[:, 0].min()
names = ["entry", "queries rotated (attention wgs)", "tiles done (attention wgs)", "slab stored (attention wgs)", "arriving at barrier 1",
         "head merged + published (wgs < heads)", "all heads seen, workgroup released", "o_proj columns stored"]
n_attn = int((t[:, 3] > t0).sum())
print(f"num_beams {B}: 256 workgroups, {n_attn} of them with attention stamps")
for i, n in enumerate(names):
    sel = t[:, i] >= t0
    if not sel.any(): continue
    a = (t[sel, i] - t0) / 100.0
    print(f"{n:44s} n {int(sel.sum()):3d}  min {a.min():6.2f}  p50 {np.median(a):6.2f}  max {a.max():6.2f} us")
# per attention workgroup, in workgroup order (the launch maps workgroup w to (kv head, slot split)): when its tiles were done / its slab stored
sel = np.where(t[:, 3] > t0)[0]
print("workgroup: tiles done / slab stored (us after the first entry)")
for k in range(0, len(sel), 8):
    print("  " + "  ".join(f"{int(w):3d}: {(t[w, 2] - t0) / 100.0:4.1f}/{(t[w, 3] - t0) / 100.0:4.1f}" for w in sel[k:k + 8]))
