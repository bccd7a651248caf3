"""Probe: K-slice count and tile width of the split-K projections with the in-launch reduction (gemm_mid.hip), 64 / 22 rows: o_proj (N = K = 4096),
down_proj (N = 4096, K = 14336), q/k/v (N = 6144, K = 4096, plain reduction).  Weights rotate over copies."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def timeit(fn, n=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
E.op_gemm_splitk_fused(torch.zeros(16, 256, device=dev).bfloat16(), E.op_pack_weight(torch.zeros(256, 256, device=dev).bfloat16()), torch.zeros(16, 256, device=dev).bfloat16(), 1)  # sets argtypes
E.op_gemm_splitk_plain(torch.zeros(16, 256, device=dev).bfloat16(), E.op_pack_weight(torch.zeros(256, 256, device=dev).bfloat16()), 256, 1)
for M in (64, 22):
    for name, N, K, kss, plain in (("o_proj", 4096, 4096, (1, 2, 4, 8), False), ("down", 4096, 14336, (2, 4, 7, 8, 14), False), ("q/k/v", 6144, 4096, (1, 2, 4), True)):
        copies = max(2, min(6, int(500e6 / (N * K * 2)) + 1))
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        A = torch.randn(M, K, device=dev).bfloat16()
        x = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ssq = torch.zeros(M, N // 32, device=dev); tickets = torch.zeros(N // 16, dtype=torch.int32, device=dev)
        line = f"M={M:3d} {name:7s}:"
        for wn in (2, 4):
            lib.isst_op_set_gemm_tuning(0, wn)
            for ks in kss:
                if K % (256 * ks): continue
                slabs = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
                if plain:
                    def fn(i):
                        rc = lib.isst_op_gemm_splitk_plain(P(A), K, P(Wps[i % copies]), P(out), N, P(slabs), P(tickets), M, N, K, ks, None, 0.0, None, E._stream_ptr()); assert rc == 0, rc
                else:
                    def fn(i):
                        rc = lib.isst_op_gemm_splitk_fused(P(A), K, P(Wps[i % copies]), P(x), P(slabs), P(ssq), P(tickets), M, N, K, ks, E._stream_ptr()); assert rc == 0, rc
                try:
                    t = timeit(fn)
                    line += f"  wn{wn}/ks{ks} {t:5.1f}"
                except AssertionError as e:
                    line += f"  wn{wn}/ks{ks}  n/a"
        print(line + f"   (weights alone {N * K * 2 / 6.6e6:5.1f} us)", flush=True)
        del Wps
lib.isst_op_set_gemm_tuning(0, 0)
