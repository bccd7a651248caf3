"""Probe: does the gate/up GEMV (896 workgroups of a (gate, up) tile PAIR = 3.5 per CU) stream slower per byte than the same bytes as 1792 single-tile workgroups (7 per CU)?
One row, K = 4096, weights rotating over 4 copies (nothing served from L2 / the Infinity Cache); with and without the fused RMSNorm.  down_proj (256 workgroups) and a 32768-row padded
pair form (1024 pairs = 4 per CU) beside them."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def run(A, Wp, N, K, epi, res, out, nw):
    n_out = N // 2 if epi == "swiglu" else N
    rc = lib.isst_op_gemm(P(A), A.stride(0), P(Wp), None, P(res), 0 if res is None else res.stride(0), P(out), out.stride(0), A.shape[0], N, K, n_out, E.EPI[epi],
                          P(nw) if nw is not None else None, 1e-5, E._stream_ptr())
    assert rc == 0, rc
def timeit(fn, n=60):
    for _ in range(8): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for M in (1, 4):
    for name, N, K, epi in (("gate_up pairs (896 wgs)", 28672, 4096, "swiglu"), ("gate_up single tiles (1792 wgs)", 28672, 4096, "none"), ("padded pairs (1024 wgs)", 32768, 4096, "swiglu"),
                            ("padded single tiles (2048 wgs)", 32768, 4096, "none"), ("qkv (384 wgs)", 6144, 4096, "none"), ("8192 cols (512 wgs)", 8192, 4096, "none"), ("down (256 wgs)", 4096, 14336, "res")):
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(4)]
        A = torch.randn(M, K, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev).bfloat16() if epi == "res" else None
        out = torch.empty(M, N // 2 if epi == "swiglu" else N, device=dev, dtype=torch.bfloat16)
        nw = torch.ones(K, device=dev).bfloat16()
        mb = N * K * 2 / 1e6
        line = f"M={M} {name:34s} {mb:6.1f} MB:"
        for label, w in (("plain", None), ("fused norm", nw)):
            if epi == "res" and w is not None: continue
            t = timeit(lambda i: run(A, Wps[i % 4], N, K, epi, res, out, w))
            line += f"  {label} {t:6.2f} us = {mb / t:5.2f} TB/s (stream part at 3 us of ramp: {mb / (t - 3):5.2f})"
        print(line, flush=True)
