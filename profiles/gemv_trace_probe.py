"""Where do the decode projections' microseconds go?  Runs the four Llama decode projections (q/k/v with fused RMSNorm, o_proj + residual,
gate/up with fused RMSNorm + SwiGLU, down + residual; one row) as a captured graph over L layers of distinct weights and reads the
per-workgroup wall-clock stamps the -DISST_GEMV_TRACE build leaves (make -C infinisst_amd/csrc trace): entry, rows staged, first weight
batch consumed, k-loop done, output stored.  Prints, per projection of the LAST layer, the launch envelope and the distribution of each
phase, plus the boundary to the launch before it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, ctypes as C
from infinisst_amd import engine as E
here = os.path.dirname(os.path.abspath(__file__))
L = int(sys.argv[1]) if len(sys.argv) > 1 else 8
FLAGS = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # gemm_set_tuning(700000 + FLAGS): experiment switches of the skinny kernel
TRACE = (sys.argv[3] if len(sys.argv) > 3 else "trace") == "trace"
NONORM = len(sys.argv) > 4 and sys.argv[4] == "nonorm"   # q/k/v and gate/up without the fused RMSNorm: A straight from memory, in order with the weights
lib = E.load_library(os.path.join(here, "..", "infinisst_amd", "libinfinisst_hip_trace.so" if TRACE else "libinfinisst_hip.so"))
E._lib = lib
lib.isst_op_set_gemm_tuning(700000 + FLAGS, 0)
P = E._ptr; dev = "cuda"
D, F, QKV = 4096, 14336, 6144
def pk(n, k): return E.op_pack_weight((torch.randn(n, k, device=dev) * 0.02).bfloat16())
layers = [dict(qkv=pk(QKV, D), o=pk(D, D), gu=pk(2 * F, D), down=pk(D, F), n1=torch.ones(D, device=dev).bfloat16(), n2=torch.ones(D, device=dev).bfloat16()) for _ in range(L)]
x = torch.randn(1, D, device=dev).bfloat16(); x2 = torch.empty_like(x); x3 = torch.empty_like(x)
qkv = torch.empty(1, QKV, device=dev, dtype=torch.bfloat16); act = torch.empty(1, F, device=dev, dtype=torch.bfloat16)
def gemm(A, K, W, res, out, N, n_out, epi, nw):
    rc = lib.isst_op_gemm(P(A), K, P(W), None, P(res), n_out, P(out), n_out, 1, N, K, n_out, E.EPI[epi], P(nw), 1e-5, E._stream_ptr()); assert rc == 0
def layer(w):
    gemm(x, D, w["qkv"], None, qkv, QKV, QKV, "none", None if NONORM else w["n1"])
    gemm(qkv, D, w["o"], x, x2, D, D, "res", None)          # (attention skipped: A = the first 4096 of qkv)
    gemm(x2, D, w["gu"], None, act, 2 * F, F, "swiglu", None if NONORM else w["n2"])
    gemm(act, F, w["down"], x2, x3, D, D, "res", None)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for w in layers: layer(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for w in layers: layer(w)
for _ in range(5): g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
per_layer = e0.elapsed_time(e1) / 20 / L * 1e3
bytes_layer = (QKV * D + D * D + 2 * F * D + D * F) * 2
print(f"{'NO NORM ' if NONORM else ''}flags {FLAGS} {'traced' if TRACE else 'production'} build, {L} layers, graph replay: {per_layer:.2f} us per layer ({bytes_layer / 1e6:.1f} MB -> {bytes_layer / per_layer / 1e6:.2f} TB/s; at 6.6 TB/s {bytes_layer / 6.6e6:.2f} us)")
if not TRACE: sys.exit(0)
buf = np.zeros(16384 * 8, dtype=np.uint64)
assert lib.isst_debug_gemv_trace_read(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.nbytes)) == 0
t = buf.reshape(16384, 8)
regions = (("q/k/v", 0, 384), ("o_proj", 2048, 256), ("gate/up", 4096, 896), ("down", 8192, 256))
T0 = min(int(t[b:b + n, 0].min()) for _, b, n in regions)
prev_end = None
def q(a): return "min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f" % tuple(np.percentile(a, [0, 10, 50, 90, 100]))
for name, b, n in regions:
    r = t[b:b + n].astype(np.int64)
    st = (r[:, :5] - T0) / 100.0   # us (100 MHz)
    xcc = (r[:, 5] >> 32) & 0xf
    hw = r[:, 5] & 0xffffffff
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    cuid = xcc * 1000 + se * 100 + sh * 16 + cu
    percu = np.bincount(np.unique(cuid, return_inverse=True)[1])
    print(f"\n{name}: {n} workgroups on {len(percu)} CUs (per CU min {percu.min()} max {percu.max()}); XCC histogram {np.bincount(xcc, minlength=8).tolist()}")
    print(f"  envelope: first entry {st[:, 0].min():8.2f}  last store {st[:, 4].max():8.2f}  = {st[:, 4].max() - st[:, 0].min():6.2f} us" + (f"   (gap after previous launch's last store: {st[:, 0].min() - prev_end:5.2f} us)" if prev_end is not None else ""))
    e = st[:, 0].min()
    print(f"  entry  - first entry : {q(st[:, 0] - e)}")
    print(f"  staged - entry       : {q(st[:, 1] - st[:, 0])}")
    print(f"  first weights - entry: {q(st[:, 2] - st[:, 0])}")
    print(f"  k-loop done - entry  : {q(st[:, 3] - st[:, 0])}")
    print(f"  k-loop done (abs)    : {q(st[:, 3] - e)}")
    print(f"  stored - k-loop done : {q(st[:, 4] - st[:, 3])}")
    print(f"  stored (abs)         : {q(st[:, 4] - e)}")
    prev_end = st[:, 4].max()
