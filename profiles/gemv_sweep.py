"""Sweep of the GEMV launch parameters (waves per workgroup W, n-tiles per workgroup NTB) for the decode shapes.
Run under   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 profiles/gemv_sweep.py
and reduce OUT/*kernel_trace.csv with profiles/gemv_sweep_reduce.py (durations per (kernel, grid, workgroup))."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E

dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)
lib = E.load_library()
SHAPES = [("qkv", 6144, 4096, "none", True, False), ("o", 4096, 4096, "res", False, True), ("gateup", 28672, 4096, "swiglu", True, False),
          ("down", 4096, 14336, "res", False, True), ("lm_head", 128272, 4096, "f32", True, False)]
for name, N, K, epi, norm, res in SHAPES:
    copies = max(2, min(12, int(600e6 // (N * K * 2)) + 1))
    packs = []
    for _ in range(copies):
        w = torch.empty((N, K), device=dev, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w)); del w
    x = torch.randn(1, K, device=dev, generator=g).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16() if norm else None
    r = torch.zeros(1, N, device=dev).bfloat16() if res else None
    for W in (2, 4, 8, 16):
        for ntb in (1, 2, 4):
            if epi == "swiglu" and ntb == 1:
                continue
            lib.isst_op_set_gemm_tuning(W, ntb)
            for i in range(24):
                E.op_gemm(x, packs[i % copies], N, epi, norm_w=nw, res=r)
            torch.cuda.synchronize()
    lib.isst_op_set_gemm_tuning(0, 0)
    for i in range(24):
        E.op_gemm(x, packs[i % copies], N, epi, norm_w=nw, res=r)
    torch.cuda.synchronize()
    del packs
    print("done", name, flush=True)
