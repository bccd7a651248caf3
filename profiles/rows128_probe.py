"""Probe: decode passes of 65..128 rows (65..128 concurrent streams, or 17..32 streams x beam 4): the default dispatch (gemm_mid as two 64-row blocks for the short
weight streams, gemm_tiled / gemm_dense otherwise) against gemm_mid forced up to 128 rows (isst_op_set_gemm_tuning(0, 128 << 8)), cold rotating weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
SH = {"qkv": (6144, 4096, "none"), "o_proj": (4096, 4096, "res"), "gate_up": (28672, 4096, "swiglu"), "down": (4096, 14336, "res")}
for M in (64, 96, 128):
    line = f"M={M:3d}:"
    for name, (N, K, epi) in SH.items():
        copies = max(3, (700 << 20) // (N * K * 2) + 1)
        packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        A = torch.randn(M, K, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev).bfloat16() if epi == "res" else None
        out = torch.empty(M, N // 2 if epi == "swiglu" else N, device=dev, dtype=torch.bfloat16)
        ts = []
        for tune in ((0, 0), (0, 128 << 8)):
            lib.isst_op_set_gemm_tuning(*tune)
            def run(i):
                rc = lib.isst_op_gemm(P(A), K, P(packs[i % copies]), None, P(res), 0 if res is None else N, P(out), out.stride(0), M, N, K, out.shape[1], E.EPI[epi], None, 0.0, E._stream_ptr())
                assert rc == 0, rc
            for i in range(4): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(40): run(i)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 40 * 1e3)
        lib.isst_op_set_gemm_tuning(0, 64 << 8)
        line += f"  {name} default {ts[0]:6.1f} / mid<=128 {ts[1]:6.1f} us"
        del packs
    print(line, flush=True)
