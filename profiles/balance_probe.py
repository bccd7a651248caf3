"""Probe: does the gate/up GEMV (896 workgroups = 3.5 per CU) lose bandwidth to the uneven workgroup count?  Same kernel at
N = 24576 (768 workgroups, 3 per CU), 28672 (896) and 32768 (1024, 4 per CU); weights rotate over copies > Infinity Cache."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"
lib = E.load_library()
P = E._ptr
K = 4096
for N in (24576, 28672, 32768, 57344):
    copies = 4
    packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(1, K, device=dev).bfloat16()
    nw = torch.ones(K, device=dev).bfloat16()
    out = torch.empty(1, N // 2, device=dev, dtype=torch.bfloat16)
    def run(i):
        rc = lib.isst_op_gemm(P(x), K, P(packs[i % copies]), None, None, 0, P(out), N // 2, 1, N, K, N // 2, E.EPI["swiglu"], P(nw), 1e-5, E._stream_ptr())
        assert rc == 0
    for i in range(8): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60): run(i)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 60 * 1e3
    print(f"N={N:6d} workgroups={N // 32:5d} ({N / 32 / 256:.2f} per CU): {us:7.2f} us  {N * K * 2 / us / 1e6:6.2f} TB/s", flush=True)
    del packs
