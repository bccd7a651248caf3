"""Does gemm_mid's gate/up launch at 64 rows leave an eighth of the chip idle?  N = 28672 gives 224 workgroups of 128 columns on 256 CUs; the same launch with
N = 32768 (256 workgroups) moves 14 % more weight bytes -- if it takes the same time, the 32 idle CUs are the loss.  Weights rotate over cold copies."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
K = 4096
for M in (22, 64):
    for N in (24576, 28672, 32768, 36864, 65536):
        copies = max(3, (700 << 20) // (N * K * 2) + 1)
        packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        A = torch.randn(M, K, device=dev).bfloat16()
        out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
        def run(i):
            rc = lib.isst_op_gemm(P(A), K, P(packs[i % copies]), None, None, 0, P(out), out.stride(0), M, N, K, out.shape[1], E.EPI["swiglu"], None, 0.0, E._stream_ptr())
            assert rc == 0, rc
        for i in range(4): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40): run(i)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print(f"M={M:3d} N={N:6d} ({N // 128:4d} workgroups): {us:7.1f} us  {N * K * 2 / us / 1e6:6.2f} TB/s", flush=True)
        del packs
