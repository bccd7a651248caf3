"""Tiled vs skinny GEMM on the dense shapes (conv layer 1, multi-stream prefill): time per launch and TFLOP/s."""
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
    for mode in ("tiled", "skinny"):
        lib.isst_op_set_gemm_tuning(-1 if mode == "skinny" else 0, 0)
        kw = dict(lda=lda, M=M, K=K) if lda else {}
        for _ in range(3): E.op_gemm(x, p, N, epi, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): E.op_gemm(x, p, N, epi, **kw)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / iters * 1e3
        print(f"{name:18s} M={M:5d} N={N:6d} K={K:6d} {mode:6s}: {us:9.1f} us  {2*M*N*K/us/1e6:8.1f} TFLOP/s", flush=True)
    lib.isst_op_set_gemm_tuning(0, 0)
bench("conv1 (k3 s2)", 1574, 512, 1536, lda=1024)
bench("prefill qkv x64", 1408, 6144, 4096)
bench("prefill gateup x64", 1408, 28672, 4096, "swiglu")
bench("prefill down x64", 1408, 4096, 14336)
bench("encoder fc1 x64", 3072, 4096, 1024)
# the conv feature extractor's implicit GEMMs at one stream (wav2vec2-large: 512 channels; k3 s2 x4, k2 s2 x2)
for name, M, K, lda in (("conv1", 1574, 1536, 1024), ("conv2", 786, 1536, 1024), ("conv3", 392, 1536, 1024), ("conv4", 195, 1536, 1024),
                        ("conv5", 97, 1024, 1024), ("conv6", 48, 1024, 1024)):
    bench(name, M, 512, K, lda=lda)
