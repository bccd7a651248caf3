"""Tiled GEMM: plain (column block, row block) grid against the XCD-aware rasterisation (8 x 8 tile patches per XCD): equality and time per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = torch.device("cuda"); g = torch.Generator(device=dev); g.manual_seed(0)
lib = E.load_library()
def bench(name, M, N, K, epi="none", lda=None, iters=20):
    w = torch.empty((N, K), device=dev, dtype=torch.float32).normal_(0, 0.02, generator=g).bfloat16()
    p = E.op_pack_weight(w); del w
    rows = M if lda is None else M * 2 + 8
    x = torch.randn(rows, K if lda is None else lda // 2, device=dev, generator=g).bfloat16()
    kw = dict(lda=lda, M=M, K=K) if lda else {}
    outs = {}
    for mode in ("plain", "raster"):
        lib.isst_op_set_gemm_tuning(100000 + (0 if mode == "plain" else 2), 0)
        for _ in range(3): o = E.op_gemm(x, p, N, epi, **kw)
        torch.cuda.synchronize(); outs[mode] = o.float().clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): E.op_gemm(x, p, N, epi, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name:18s} M={M:5d} N={N:6d} K={K:6d} {mode:6s}: {us:9.1f} us  {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    d = (outs["plain"] - outs["raster"]).abs().max().item()
    print(f"   max |plain - raster| = {d:.3g}  nan={torch.isnan(outs['raster']).any().item()}", flush=True)
    lib.isst_op_set_gemm_tuning(100001, 0)
bench("small ragged", 300, 1000 // 16 * 16, 512)
bench("conv1 (k3 s2)", 1574, 512, 1536, lda=1024)
bench("prefill qkv x64", 1408, 6144, 4096)
bench("prefill gateup x64", 1408, 28672, 4096, "swiglu")
bench("prefill down x64", 1408, 4096, 14336)
bench("encoder fc1 x64", 3072, 4096, 1024, "bias_gelu" if False else "none")
bench("big square", 4096, 8192, 8192)
bench("prefill gateup x16", 352, 28672, 4096, "swiglu")
bench("prefill down x16", 352, 4096, 14336)
bench("prefill qkv x32", 704, 6144, 4096)
bench("encoder qkv x64", 3072, 3072, 1024)
