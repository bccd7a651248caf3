"""In-kernel stamps of gemm_wide (make -C infinisst_amd/csrc EXTRA=-DISST_WIDE_PROBE trace): cycles per K-step, the shader clock, the phases of step 8.
    python profiles/wide_trace_probe.py [M] [variant] [dbg]     (dbg 3: both descriptors emptied = the skeleton alone)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from infinisst_amd import engine as E
here = os.path.dirname(os.path.abspath(__file__))
M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
var = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib = E.load_library(os.path.join(here, "..", "infinisst_amd", "libinfinisst_hip_trace.so"))
E._lib = lib
P = E._ptr; dev = "cuda"
N, K = 28672, 4096
packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(4)]
A = torch.randn(M, K, device=dev).bfloat16()
out = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
lib.isst_op_set_gemm_tuning(900000 + 100 * dbg + 20 + var, 0)
for i in range(12):
    rc = lib.isst_op_gemm(P(A), K, P(packs[i % 4]), None, None, 0, P(out), out.stride(0), M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr()); assert rc == 0
torch.cuda.synchronize()
WG = 224
buf = np.zeros((WG, 256), dtype=np.uint64)
assert lib.isst_debug_wide_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
b = buf.astype(np.int64)
loop_cyc = b[:, 2] - b[:, 1]; loop_rt = (b[:, 3] - b[:, 0]) * 10.0  # ns
print(f"M={M} variant {var} dbg {dbg}: k-loop {np.median(loop_cyc):.0f} cycles (min {loop_cyc.min()}, max {loop_cyc.max()}), {np.median(loop_rt) / 1e3:.2f} us -> clock {np.median(loop_cyc / loop_rt):.3f} GHz")
steps = b[:, 4:4 + 64]
d = np.diff(steps, axis=1)
pro = b[:, 1] - b[:, 241]; epi = b[:, 242] - b[:, 2]; tot_rt = (b[:, 243] - b[:, 240]) * 10.0
print(f"entry -> loop entry {np.median(pro):.0f} cycles; loop exit -> stored {np.median(epi):.0f} cycles; wave 0 lifetime {np.median(tot_rt) / 1e3:.2f} us; launch envelope (first entry -> last exit) {(b[:, 243].max() - b[:, 240].min()) / 100:.2f} us; entries spread {(b[:, 240].max() - b[:, 240].min()) / 100:.2f} us")
print("cycles per step (median over workgroups), steps 1..63:", " ".join(f"{int(x)}" for x in np.median(d, axis=0)))
print(f"first step end - loop entry: {np.median(steps[:, 0] - b[:, 1]):.0f}")
NG = 2 if M <= 128 else 4
ph = b[:, [205] + [210 + g for g in range(NG)] + [206, 4 + 8]]
names = [f"step-8 entry -> after MFMA group 0 (+ fillers)"] + [f"-> after MFMA group {g}" for g in range(1, NG)] + ["-> before the barrier", "-> after the barrier"]
pd = np.diff(ph, axis=1)
for n, col in zip(names, pd.T): print(f"  {n}: median {np.median(col):.0f}  (p10 {np.percentile(col, 10):.0f}, p90 {np.percentile(col, 90):.0f})")
