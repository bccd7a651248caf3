"""Prefill-shaped GEMMs (M = 22 rows): NTB / W variants of the skinny kernel.  Run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(0)
lib = E.load_library()
for name, N, K, epi, res in [("qkv", 6144, 4096, "none", False), ("o", 4096, 4096, "res", True), ("gateup", 28672, 4096, "swiglu", False), ("down", 4096, 14336, "res", True)]:
    copies = max(2, min(12, int(600e6 // (N * K * 2)) + 1))
    packs = []
    for _ in range(copies):
        w = torch.empty((N, K), device=dev, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w)); del w
    x = torch.randn(22, K, device=dev, generator=g).bfloat16()
    r = torch.zeros(22, N, device=dev).bfloat16() if res else None
    for W in (0, 4, 8):
        for ntb in (0, 2, 4):
            if epi == "swiglu" and ntb == 0 and W != 0: continue
            lib.isst_op_set_gemm_tuning(W, ntb)
            for i in range(16): E.op_gemm(x, packs[i % copies], N, epi, res=r)
            torch.cuda.synchronize()
    lib.isst_op_set_gemm_tuning(0, 0)
    del packs
