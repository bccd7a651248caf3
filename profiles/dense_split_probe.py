"""The narrow projections of a many-row pass (o_proj, down_proj: 4096 output columns) on gemm_tiled (128 x 128) and gemm_dense (256 x 256), with
K split over workgroups into fp32 slabs + the reducing residual / RMSNorm kernel (the engine's form) against the unsplit launch + RMSNorm kernel;
q/k/v and gate/up unsplit.  us per (GEMM + the norm that follows), weights rotated over copies.

    python profiles/dense_split_probe.py [rows ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinisst_amd import engine as E

lib = E.load_library()
dev = "cuda"
P = E._ptr


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def mode(m):
    lib.isst_op_set_gemm_tuning(800000 + m, 0)


rows = [int(a) for a in sys.argv[1:]] or [176, 352, 704, 1408]
for M in rows:
    for name, N, K in (("o_proj", 4096, 4096), ("down", 4096, 14336)):
        copies = 3
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        A = torch.randn(M, K, device=dev).bfloat16()
        x = torch.randn(M, N, device=dev).bfloat16()
        nw = torch.ones(N, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        slabs = torch.empty(8 * M * N, device=dev, dtype=torch.float32)
        line = f"{name:7s} M={M:5d}:"
        for mname, m in (("tiled", 0), ("dense", 2)):
            mode(m)
            for ks in (1, 2, 4, 8):
                if K % (64 * ks * 2) != 0:
                    continue

                def run(i):
                    if ks == 1:
                        rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), None, P(x), N, P(x), N, M, N, K, N, E.EPI["res"], None, 0.0, E._stream_ptr())
                        assert rc == 0
                        rc = lib.isst_op_rmsnorm(P(x), P(nw), P(out), M, N, 1e-5, E._stream_ptr())
                    else:
                        rc = lib.isst_op_gemm_splitk_rmsnorm(P(A), K, P(Wps[i % copies]), P(x), P(nw), P(out), P(slabs), M, N, K, ks, 1e-5, E._stream_ptr())
                    assert rc == 0, rc
                line += f"  {mname} ks{ks} {timeit(run):6.1f}"
        mode(1)
        print(line, flush=True)
        del Wps
