"""The three narrow projections of a 64-stream decode pass in the forms the engine runs them (gemm_mid.hip, 64 rows: q/k/v in 2 K slices with the plain in-launch
reduction, o_proj and down_proj in 4 K slices with the residual + sums-of-squares reduction), a few launches each with rotating weights -- run under
rocprofv3 --pmc to read the L2 <- CU request counters of these launches (the activation re-read ratio as a number).
    rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum --kernel-trace --output-format csv -d out -- python3 profiles/mid_l2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
E.op_gemm_splitk_fused(torch.zeros(16, 256, device=dev).bfloat16(), E.op_pack_weight(torch.zeros(256, 256, device=dev).bfloat16()), torch.zeros(16, 256, device=dev).bfloat16(), 1)  # sets argtypes
E.op_gemm_splitk_plain(torch.zeros(16, 256, device=dev).bfloat16(), E.op_pack_weight(torch.zeros(256, 256, device=dev).bfloat16()), 256, 1)
M = 64
for name, N, K, ks, plain in (("q/k/v", 6144, 4096, 2, True), ("o_proj", 4096, 4096, 4, False), ("down", 4096, 14336, 4, False)):
    copies = max(2, min(6, int(500e6 / (N * K * 2)) + 1))
    Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    A = torch.randn(M, K, device=dev).bfloat16()
    x = torch.randn(M, N, device=dev).bfloat16(); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ssq = torch.zeros(M, N // 32, device=dev); tickets = torch.zeros(N // 16, dtype=torch.int32, device=dev)
    slabs = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
    for i in range(12):
        if plain:
            rc = lib.isst_op_gemm_splitk_plain(P(A), K, P(Wps[i % copies]), P(out), N, P(slabs), P(tickets), M, N, K, ks, None, 0.0, None, E._stream_ptr())
        else:
            rc = lib.isst_op_gemm_splitk_fused(P(A), K, P(Wps[i % copies]), P(x), P(slabs), P(ssq), P(tickets), M, N, K, ks, E._stream_ptr())
        assert rc == 0, rc
    torch.cuda.synchronize()
    print(name, "done", flush=True)
