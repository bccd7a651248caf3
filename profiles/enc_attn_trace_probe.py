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
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 1  # streams
eng, weights, sys_n = bench.build_engine(cfg, NS, 1, dev, 1, None)
loop = bench.ChunkLoop(eng, cfg, gen, list(range(NS)), sys_n); loop.import_steady_state(dev)
for _ in range(6): loop.step()
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
assert lib.isst_debug_enc_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(4096, 8)[:min(4096, 16 * NS * (3 if NS == 1 else 1))].astype(np.int64)
t0 = t[:, 0].min()
names = ["entry", "queries rotated", "scores written (phase 1)", "softmax done (phase 2)", "P.V done (phase 3)", "output stored", "V^T appended / end"]
for i, n in enumerate(names):
    a = (t[:, i] - t0) / 100.0
    print(f"{n:28s} min {a.min():6.2f}  p50 {np.median(a):6.2f}  max {a.max():6.2f} us")

print(f"{len(t)} workgroups; per-phase durations (us):")
for i in range(1, len(names)):
    d = (t[:, i] - t[:, i - 1]) / 100.0
    print(f"  {names[i]:28s} min {d.min():6.2f}  p50 {np.median(d):6.2f}  max {d.max():6.2f}")
life = (t[:, len(names) - 1] - t[:, 0]) / 100.0
print(f"workgroup lifetime min {life.min():.2f} p50 {np.median(life):.2f} max {life.max():.2f} us; launch span {(t[:, len(names) - 1].max() - t0) / 100.0:.2f} us")
ent = np.sort((t[:, 0] - t0) / 100.0)
print("entry times (us) at quantiles 0/25/50/75/100 %:", [round(float(np.quantile(ent, q)), 2) for q in (0, .25, .5, .75, 1)])
print("entry time of every 32nd workgroup in start order (us):", [round(float(x), 1) for x in ent[::32]])
end = np.sort((t[:, len(names) - 1] - t0) / 100.0)
alive = [(int((ent <= x).sum() - (end <= x).sum())) for x in np.arange(0, ent.max() + 40, 5.0)]
print("workgroups alive at 5 us steps:", alive)
