"""Context for gemm_tiled's 0.98-1.02 PFLOP/s: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS, bf16, fp32 accumulate) reaches on the same four
1408-row prefill shapes, random data, weights rotated over copies.  Not used by the product (epilogues, packed weights and split-K slabs are the kernel's
own); a yardstick only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def timeit(fn, n=30):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
import sys
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1408
for name, N, K, epi in (("q/k/v", 6144, 4096, "none"), ("o_proj", 4096, 4096, "none"), ("gate/up", 28672, 4096, "swiglu"), ("down", 4096, 14336, "none")):
    copies = 3
    Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
    Wps = [E.op_pack_weight(w) for w in Ws]
    A = torch.randn(M, K, device=dev).bfloat16()
    n_out = N // 2 if epi == "swiglu" else N
    out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16); out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    def ours(i):
        rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), None, None, 0, P(out), n_out, M, N, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr()); assert rc == 0
    def libr(i):
        torch.matmul(A, Ws[i % copies].t(), out=out2)
    t1, t2 = timeit(ours), timeit(libr)
    fl = 2.0 * M * N * K
    print(f"{name:8s} M={M} N={N:6d} K={K:6d}: gemm_tiled {t1:7.1f} us = {fl / t1 / 1e6:6.0f} TFLOP/s   torch.matmul {t2:7.1f} us = {fl / t2 / 1e6:6.0f} TFLOP/s", flush=True)
