"""gemm_dense.hip (256 x 256 tile, 8-wave ping-pong, LDS-DMA) against gemm_tiled.hip (128 x 128) and the vendor library (torch.matmul ->
hipBLASLt, a yardstick only) on the many-row shapes of the path; random operands, weights rotated over copies; outputs of the two
hand-written kernels compared bit for bit.

    python profiles/dense_probe.py [rows ...]
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


shapes = [("q/k/v", 6144, 4096, "none"), ("o_proj", 4096, 4096, "none"), ("gate/up", 28672, 4096, "swiglu"), ("down", 4096, 14336, "none"),
          ("enc qkv", 3072, 1024, "bias"), ("enc fc1", 4096, 1024, "bias_gelu"), ("enc fc2", 1024, 4096, "bias_res")]
rows = [int(a) for a in sys.argv[1:]] or [1408, 352]
for M in rows:
    for name, N, K, epi in shapes:
        if name.startswith("enc") and M == rows[0]:
            Mx = 3072
        elif name.startswith("enc"):
            continue
        else:
            Mx = M
        copies = 3
        Ws = [(torch.randn(N, K, device=dev) * 0.02).bfloat16() for _ in range(copies)]
        Wps = [E.op_pack_weight(w) for w in Ws]
        A = torch.randn(Mx, K, device=dev).bfloat16()
        bias = torch.randn(N, device=dev).bfloat16() if "bias" in epi else None
        res = torch.randn(Mx, N, device=dev).bfloat16() if "res" in epi else None
        n_out = N // 2 if epi == "swiglu" else N
        outs = [torch.zeros(Mx, n_out, device=dev, dtype=torch.bfloat16) for _ in range(2)]
        out2 = torch.empty(Mx, N, device=dev, dtype=torch.bfloat16)

        def ours(i, o):
            rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), P(bias), P(res), N if res is not None else 0, P(o), n_out, Mx, N, K, n_out, E.EPI[epi], None, 0.0,
                                  E._stream_ptr())
            assert rc == 0, rc

        mode(0)
        ours(0, outs[0])
        t_tiled = timeit(lambda i: ours(i, outs[0]))
        ours(0, outs[0])
        mode(2)
        ours(0, outs[1])
        t_dense = timeit(lambda i: ours(i, outs[1]))
        ours(0, outs[1])
        torch.cuda.synchronize()
        same = bool(torch.equal(outs[0], outs[1]))
        nbad = int((outs[0] != outs[1]).sum())
        t_lib = timeit(lambda i: torch.matmul(A, Ws[i % copies].t(), out=out2))
        mode(1)
        fl = 2.0 * Mx * N * K
        print(f"{name:8s} M={Mx:5d} N={N:6d} K={K:6d}: tiled {t_tiled:7.1f} us {fl / t_tiled / 1e6:6.0f} TF | dense {t_dense:7.1f} us {fl / t_dense / 1e6:6.0f} TF | "
              f"library {t_lib:7.1f} us {fl / t_lib / 1e6:6.0f} TF | bit-identical {same} ({nbad} differing)", flush=True)
        del Ws, Wps
