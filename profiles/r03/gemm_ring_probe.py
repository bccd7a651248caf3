"""gemm_ring.hip (33..64 rows, every operand through an LDS-DMA ring) against gemm_mid.hip on the Llama-3.1-8B decode projections at 64 and 40 rows, in
the forms the engine launches them: q/k/v as K slices reduced inside the launch (tickets), o_proj / down_proj as K slices with the launch-free
residual (+ sums of squares), gate/up and lm_head unsplit; with and without the rows normalised while they are staged.  Weights rotate over 4 copies.

    python profiles/ring_probe.py [rows ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinisst_amd import engine as E

lib = E.load_library()
dev = "cuda"
P = E._ptr
COPIES = 4


def timeit(fn, n=40):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def ring(on, np=0):
    lib.isst_op_set_gemm_tuning(900000 + int(on), 0)
    lib.isst_op_set_gemm_tuning(910000 + np, 0)


import ctypes as C
lib.isst_op_gemm_splitk_fused.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.isst_op_gemm_splitk_plain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
lib.isst_op_gemm_norm_ssq.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]

rows = [int(a) for a in sys.argv[1:]] or [64, 40]
for M in rows:
    for name, N, K, kind, slices in (("q/k/v", 6144, 4096, "plain", (2, 4)), ("o_proj", 4096, 4096, "fused", (2, 4)), ("gate/up", 28672, 4096, "swiglu", (1,)),
                                      ("down", 4096, 14336, "fused", (4, 8)), ("lm_head", 128272, 4096, "f32", (1,))):
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(COPIES if N < 100000 else 2)]
        nc = len(Wps)
        A = torch.randn(M, K, device=dev).bfloat16()
        x = torch.randn(M, N if kind == "fused" else K, device=dev).bfloat16()
        nw = torch.ones(K, device=dev).bfloat16()
        ssq_in = (x.float() ** 2).view(M, K // 32, 32).sum(-1).contiguous() if kind != "fused" else None
        n_out = N // 2 if kind == "swiglu" else N
        out = torch.empty(M, n_out, device=dev, dtype=torch.float32 if kind == "f32" else torch.bfloat16)
        slabs = torch.empty(8 * M * N if kind in ("fused", "plain") else 1, device=dev, dtype=torch.float32)
        ssq = torch.zeros(M, N // 32, device=dev, dtype=torch.float32)
        tickets = torch.zeros(N // 16 + 64, device=dev, dtype=torch.int32)
        mb = N * K * 2 / 1e6
        line = f"M={M:3d} {name:8s} W={mb:6.1f} MB ({mb / 6.6e6 * 1e6:5.1f} us at 6.6 TB/s):"
        for ks in slices:
            for label, on, np in (("mid", 0, 0), ("ring", 1, 0), ("ring np2", 1, 2), ("ring np3", 1, 3), ("ring np4", 1, 4)):
                if label == "ring np3" and (N // 32) % 3:
                    continue
                ring(on, np)

                def run(i):
                    W = Wps[i % nc]
                    if kind == "fused":
                        rc = lib.isst_op_gemm_splitk_fused(P(A), K, P(W), P(x), P(slabs), P(ssq), P(tickets), M, N, K, ks, E._stream_ptr())
                    elif kind == "plain":   # rows normalised while staged (the q/k/v form)
                        rc = lib.isst_op_gemm_splitk_plain(P(x), K, P(W), P(out), N, P(slabs), P(tickets), M, N, K, ks, P(nw), 1e-5, P(ssq_in), E._stream_ptr())
                    else:
                        rc = lib.isst_op_gemm_norm_ssq(P(x), K, P(W), P(out), n_out, M, N, K, n_out, E.EPI[kind], P(nw), 1e-5, P(ssq_in), E._stream_ptr())
                    assert rc == 0, rc
                line += f"  ks{ks} {label} {timeit(run):6.1f}"
        ring(1, 0)
        print(line, flush=True)
        del Wps
