"""256x128 LDS-DMA dense kernel vs the 128x128 tiled kernel: equality and time per launch on the many-stream shapes."""
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
    for mode in ("tiled", "dense"):
        lib.isst_op_set_gemm_tuning(100000 + (1 << 29 if mode == "tiled" else 65), 0)
        for _ in range(3): o = E.op_gemm(x, p, N, epi, **kw)
        torch.cuda.synchronize(); outs[mode] = o.float().clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): E.op_gemm(x, p, N, epi, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name:18s} M={M:5d} N={N:6d} K={K:6d} {mode:6s}: {us:9.1f} us  {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    d = (outs["tiled"] - outs["dense"]).abs().max().item()
    print(f"   max |tiled - dense| = {d:.3g}  nan={torch.isnan(outs['dense']).any().item()}", flush=True)
    lib.isst_op_set_gemm_tuning(100000 + (1 << 29), 0)
bench("small ragged", 300, 1000 // 16 * 16, 512)
bench("conv1 (k3 s2)", 1574, 512, 1536, lda=1024)
bench("prefill qkv x64", 1408, 6144, 4096)
bench("prefill gateup x64", 1408, 28672, 4096, "swiglu")
bench("prefill down x64", 1408, 4096, 14336)
bench("encoder fc1 x64", 3072, 4096, 1024, "bias_gelu" if False else "none")
bench("big square", 4096, 8192, 8192)
