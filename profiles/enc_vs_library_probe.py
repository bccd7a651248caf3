"""Yardstick for the speech encoder's dense GEMMs at many streams (64 streams x 48 frames = 3072 rows, K = 1024 / 4096): gemm_tiled with its fused bias
epilogues against torch.matmul (hipBLASLt) + what the epilogue passes would cost.  Not used by the product."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def timeit(fn, n=40):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
for name, N, K, epi in (("qkv (bias)", 3072, 1024, "bias"), ("out_proj (bias+res)", 1024, 1024, "bias_res"), ("fc1 (bias+gelu)", 4096, 1024, "bias_gelu"), ("fc2 (bias+res)", 1024, 4096, "bias_res")):
    copies = 8
    Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    Wps = [E.op_pack_weight(w) for w in Ws]
    A = torch.randn(M, K, device=dev).bfloat16(); bias = torch.randn(N, device=dev).bfloat16(); res = torch.randn(M, N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def ours(i):
        rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), P(bias), P(res) if "res" in epi else None, N, P(out), N, M, N, K, N, E.EPI[epi], None, 0.0, E._stream_ptr()); assert rc == 0
    def libr(i):
        torch.matmul(A, Ws[i % copies].t(), out=out2)
    def libr_bias(i):
        torch.addmm(bias, A, Ws[i % copies].t(), out=out2)
    t1, t2, t3 = timeit(ours), timeit(libr), timeit(libr_bias)
    fl = 2.0 * M * N * K
    print(f"{name:20s} M={M} N={N:5d} K={K:5d}: gemm_tiled (fused epilogue) {t1:6.1f} us = {fl / t1 / 1e6:5.0f} TFLOP/s   torch.matmul {t2:6.1f} us   torch.addmm (bias) {t3:6.1f} us", flush=True)
