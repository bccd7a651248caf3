"""Probe: gate/up with the fused RMSNorm at 1..8 rows, tile pairs (epi swiglu, 896 workgroups) against self-paired tiles (epi swiglu8, 1792), with the launcher's waves per
workgroup and with 4 forced (isst_op_set_gemm_tuning(4, 0)): from which row count on do the pairs win?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"
def timeit(fn, n=60):
    for _ in range(8): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
I, K = 14336, 4096
Wg = [(torch.randn(I, K, device=dev) * 0.02).bfloat16() for _ in range(3)]
Wu = [(torch.randn(I, K, device=dev) * 0.02).bfloat16() for _ in range(3)]
pairs = [E.op_pack_weight(torch.stack([g.view(I // 16, 16, K), u.view(I // 16, 16, K)], dim=1).reshape(2 * I, K)) for g, u in zip(Wg, Wu)]
self8 = [E.op_pack_gateup8(g, u) for g, u in zip(Wg, Wu)]
del Wg, Wu
nw = torch.ones(K, device=dev).bfloat16()
for M in (1, 2, 3, 4, 5, 6, 8, 10, 12):
    A = torch.randn(M, K, device=dev).bfloat16()
    line = f"M={M}:"
    for wv in (0, 4, 8):
        lib.isst_op_set_gemm_tuning(wv, 0)
        tp = timeit(lambda i: E.op_gemm(A, pairs[i % 3], 2 * I, "swiglu", norm_w=nw))
        ts = timeit(lambda i: E.op_gemm(A, self8[i % 3], 2 * I, "swiglu8", norm_w=nw))
        line += f"   waves {wv or 'auto'}: pairs {tp:6.2f} us, self-paired {ts:6.2f} us"
    lib.isst_op_set_gemm_tuning(0, 0)
    tp = timeit(lambda i: E.op_gemm(A, pairs[i % 3], 2 * I, "swiglu"))
    ts = timeit(lambda i: E.op_gemm(A, self8[i % 3], 2 * I, "swiglu8"))
    line += f"   NO fused norm (the engine's form from 5 rows on): pairs {tp:6.2f} us, self-paired {ts:6.2f} us"
    print(line, flush=True)
