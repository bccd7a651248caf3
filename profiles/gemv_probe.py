"""GEMV micro-benchmark on cold weights: time per launch of the packed-weight kernel for the decode shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E

dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)

def bench(N, K, epi, M=1, norm=None, eps=1e-5, iters=60, res=False):
    w_bytes = N * K * 2
    copies = max(2, int(700e6 // w_bytes) + 1)
    packs = []
    for _ in range(min(copies, 16)):
        w = torch.empty((N, K), device=dev, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w)); del w
    x = torch.randn(M, K, device=dev, generator=g).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16() if norm else None
    r = torch.zeros(M, N, device=dev).bfloat16() if res else None
    for p in packs:
        E.op_gemm(x, p, N, epi, norm_w=nw, norm_eps=eps, res=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        E.op_gemm(x, packs[i % len(packs)], N, epi, norm_w=nw, norm_eps=eps, res=r)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"N={N:6d} K={K:6d} M={M:2d} {epi:7s} norm={str(bool(norm)):5s} eps={eps:8.1e}: {us:7.2f} us  {w_bytes/us/1e3:7.1f} GB/s", flush=True)

for M in (1,):
    bench(6144, 4096, "none", M)
    bench(6144, 4096, "none", M, norm=True)
    bench(6144, 4096, "none", M, norm=True, eps=-1.0)     # no prologue
    bench(6144, 4096, "none", M, norm=True, eps=1000.0)   # no in-loop apply
    bench(28672, 4096, "swiglu", M)
    bench(28672, 4096, "swiglu", M, norm=True)
    bench(28672, 4096, "swiglu", M, norm=True, eps=-1.0)
    bench(28672, 4096, "swiglu", M, norm=True, eps=1000.0)
    bench(4096, 4096, "res", M, res=True)
    bench(4096, 14336, "res", M, res=True)
    bench(128272, 4096, "f32", M)
