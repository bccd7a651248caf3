"""Probe: what the fused-RMSNorm prologue (AMODE 2: stage the row in LDS, normalise, two workgroup barriers) costs per launch.
Same GEMV with and without norm_w, weights rotating over copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
K = 4096
MROWS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for name, N, epi in (("gate_up", 28672, "swiglu"), ("qkv", 6144, "none"), ("lm_head", 128272, "f32")):
    copies = 4 if N < 100000 else 2
    Np = (N + 15) // 16 * 16
    packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(MROWS, K, device=dev).bfloat16(); nw = torch.ones(K, device=dev).bfloat16()
    n_out = N // 2 if epi == "swiglu" else N
    out = torch.empty(MROWS, n_out, device=dev, dtype=torch.float32 if epi == "f32" else torch.bfloat16)
    res = {}
    for fused in (True, False):
        def run(i):
            rc = lib.isst_op_gemm(P(x), K, P(packs[i % copies]), None, None, 0, P(out), n_out, MROWS, N, K, n_out, E.EPI[epi], P(nw) if fused else None, 1e-5, E._stream_ptr())
            assert rc == 0, rc
        for i in range(8): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): run(i)
        e1.record(); torch.cuda.synchronize()
        res[fused] = e0.elapsed_time(e1) / 60 * 1e3
    print(f"{name:8s} N={N:6d}: fused norm {res[True]:7.2f} us, plain {res[False]:7.2f} us, prologue {res[True] - res[False]:5.2f} us", flush=True)
    del packs
