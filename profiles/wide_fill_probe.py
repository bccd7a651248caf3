"""Probe: is gemm_wide's gate/up launch (224 workgroups of 128 columns on 256 CUs) bound per CU?  The same kernel on 192 / 224 / 256 column blocks (N = 24576 / 28672 / 32768),
64 / 128 / 256 rows, K = 4096, weights rotating over 4 copies.  If 256 blocks take what 224 take, spreading the 224 blocks' columns over 256 workgroups (112 columns each) would
return the difference."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def timeit(fn, n=40):
    for _ in range(6): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
K = 4096
for M in (64, 128, 256):
    A = torch.randn(M, K, device=dev).bfloat16()
    for N in (24576, 28672, 32768):
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(4)]
        out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
        def run(i):
            rc = lib.isst_op_gemm(P(A), A.stride(0), P(Wps[i % 4]), None, None, 0, P(out), out.stride(0), M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr())
            assert rc == 0, rc
        t = timeit(run)
        mb = N * K * 2 / 1e6
        print(f"M={M:3d} N={N} ({N // 128} column blocks) {mb:6.1f} MB: {t:7.2f} us = {mb / t:5.2f} TB/s, {t / (N // 128) * 1000:6.1f} ns per block", flush=True)
        del Wps
