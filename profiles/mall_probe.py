"""Is the weight-streaming GEMV faster when its weights already sit in the Infinity Cache?  (cold = cycling many copies,
warm = one copy re-read).  Bounds the gain of prefetching the next weights during latency-bound kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E

dev = torch.device("cuda")
g = torch.Generator(device=dev); g.manual_seed(0)

def bench(N, K, epi, copies, iters=80, norm=False, res=False):
    packs = []
    for _ in range(copies):
        w = torch.empty((N, K), device=dev, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
        packs.append(E.op_pack_weight(w)); del w
    x = torch.randn(1, K, device=dev, generator=g).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16() if norm else None
    r = torch.zeros(1, N, device=dev).bfloat16() if res else None
    for _ in range(3):
        for p in packs:
            E.op_gemm(x, p, N, epi, norm_w=nw, res=r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        E.op_gemm(x, packs[i % copies], N, epi, norm_w=nw, res=r)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"N={N:6d} K={K:6d} {epi:7s} copies={copies:2d} ({copies*N*K*2/1e6:7.0f} MB): {us:7.2f} us  {N*K*2/us/1e3:7.1f} GB/s", flush=True)

for copies in (1, 4):
    bench(28672, 4096, "swiglu", copies, norm=True)
for copies in (1, 6):
    bench(4096, 14336, "res", copies, res=True)
