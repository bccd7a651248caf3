"""Probe (measurement tool): the 17..64-row GEMM (gemm_mid.hip) against the 16-column skinny kernel on the Llama-3.1-8B
projection shapes, at 12 and 16 rows (9..16 streams decoding; gemm_mid enabled from 9 rows through the tuning hook).
    python profiles/mid_probe.py > gpurun_out/mid_probe.txt
Weights rotate over 4 copies so that nothing is served from L2 / Infinity Cache."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E

lib = E.load_library()
dev = "cuda"
P = E._ptr


def gemm_raw(A, Wp, N, K, epi, res, out):  # no allocation inside the timed loop
    n_out = N // 2 if epi == "swiglu" else N
    rc = lib.isst_op_gemm(P(A), A.stride(0), P(Wp), None, P(res), 0 if res is None else res.stride(0), P(out), out.stride(0), A.shape[0], N, K,
                          n_out, E.EPI[epi], None, 0.0, E._stream_ptr())
    assert rc == 0, rc


def splitk_raw(A, Wp, x, nw, out, slabs, N, K, ks):
    rc = lib.isst_op_gemm_splitk_rmsnorm(P(A), A.stride(0), P(Wp), P(x), P(nw), P(out), P(slabs), A.shape[0], N, K, ks, 1e-5, E._stream_ptr())
    assert rc == 0, rc
SHAPES = {"qkv": (6144, 4096, "none"), "o_proj": (4096, 4096, "res"), "gate_up": (28672, 4096, "swiglu"), "down": (4096, 14336, "res")}
COPIES = 4


def timeit(fn, n=40):
    for _ in range(5):
        fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


lib.isst_op_set_gemm_tuning(200008, 0)  # gemm_mid from 9 rows on
for M in (12, 16):
    for name, (N, K, epi) in SHAPES.items():
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(COPIES)]
        A = torch.randn(M, K, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev).bfloat16() if epi == "res" else None
        mb = N * K * 2 / 1e6
        out = torch.empty(M, N // 2 if epi == 'swiglu' else N, device=dev, dtype=torch.bfloat16)
        slabs = torch.empty(16, M, N, device=dev, dtype=torch.float32)
        line = f"M={M:3d} {name:8s} W={mb:6.1f} MB :"
        for label, tune in (("skinny", (-1, 0)), ("mid wn2", (0, 2)), ("mid wn4", (0, 4)), ("mid wn8", (0, 8)), ("wn4 noA", (0, 4 + 16)), ("wn4 noW", (0, 4 + 32)), ("wn4 none", (0, 4 + 48))):
            lib.isst_op_set_gemm_tuning(*tune)
            t = timeit(lambda i: gemm_raw(A, Wps[i % COPIES], N, K, epi, res, out))
            line += f"  {label} {t:6.1f} us ({mb / t / 1e3 * 1e3:5.0f} GB/s)"
        lib.isst_op_set_gemm_tuning(0, 0)
        if epi == "res":
            x = res.clone()
            nw = torch.ones(N, device=dev).bfloat16()
            for ks in (2, 4, 8, 16):
                if K % (256 * ks):
                    continue
                for wn in (2, 4, 8):
                    lib.isst_op_set_gemm_tuning(0, wn)
                    t = timeit(lambda i: splitk_raw(A, Wps[i % COPIES], x, nw, out, slabs, N, K, ks))
                    line += f"  splitK{ks}/wn{wn}+norm {t:6.1f}"
            lib.isst_op_set_gemm_tuning(0, 0)
            t = timeit(lambda i: E.op_rmsnorm(x, nw))
            line += f"  (rmsnorm alone {t:5.1f})"
        print(line, flush=True)
