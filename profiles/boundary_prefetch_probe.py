"""Premise check for the boundary-prefetch idea: is a decode GEMV shorter when the first window of its weights -- what its resident
workgroups request at t = 0 -- already sits in the Infinity Cache?  Per shape: weights rotate over > 600 MB of copies (cold), the GEMV is timed with an event
pair around it alone, with and without a torch kernel that has just read the first WINDOW_KB of every 16-row tile of that copy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
SH = {"qkv": (6144, 4096, "none"), "o_proj": (4096, 4096, "res"), "gate_up": (28672, 4096, "swiglu"), "down": (4096, 14336, "res")}
M = 1
for name, (N, K, epi) in SH.items():
    copies = max(3, (700 << 20) // (N * K * 2) + 1)
    packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    A = torch.randn(M, K, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16() if epi == "res" else None
    out = torch.empty(M, N // 2 if epi == "swiglu" else N, device=dev, dtype=torch.bfloat16)
    filler = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    def run(i):
        rc = lib.isst_op_gemm(P(A), K, P(packs[i % copies]), None, P(res), 0 if res is None else N, P(out), out.stride(0), M, N, K, out.shape[1], E.EPI[epi], None, 0.0, E._stream_ptr())
        assert rc == 0, rc
    for window_kb in (0, 16, 64):
        ts = []
        for i in range(60):
            w = packs[i % copies].view(N // 16, -1)  # [tiles][KT * 512] bf16
            filler.add_(1)  # 64 MB of other traffic first: whatever the previous iteration left in the L2s is gone
            if window_kb:
                w[:, : window_kb * 512].float().sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(i); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = np.array(ts[10:])
        print(f"{name:8s} window {window_kb:3d} KB per tile ({N // 16 * window_kb / 1024:6.1f} MB): GEMV event bracket median {np.median(ts):7.2f} us  mean {ts.mean():7.2f}", flush=True)
    del packs
