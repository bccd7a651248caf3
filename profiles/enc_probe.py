"""Probe (measurement tool): the encoder's four projections at one stream (48 rows, wav2vec2-large: D 1024, FFN 4096) --
bias+residual GEMM followed by LayerNorm (what the engine runs) against split-K slabs + the reducing LayerNorm, raw calls, no
allocation in the timed loop, weights rotating over 24 layers' worth of copies.
    python profiles/enc_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
M, D, F, COPIES = 48, 1024, 4096, 24

def timeit(fn, n=96):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n

def packs(N, K): return [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(COPIES)]
x = torch.randn(M, D, device=dev).bfloat16(); xn = torch.empty_like(x); ffn = torch.randn(M, F, device=dev).bfloat16()
lnw = torch.ones(D, device=dev).bfloat16(); lnb = torch.zeros(D, device=dev).bfloat16()
for name, N, K, A, epi in (("qkv", 3 * D, D, x, "bias"), ("fc1", F, D, x, "bias_gelu"), ("out", D, D, x, "bias_res"), ("fc2", D, F, ffn, "bias_res")):
    Wp = packs(N, K); bias = torch.zeros(N, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); res = x.clone() if epi == "bias_res" else None
    def g(i, tune=None):
        rc = lib.isst_op_gemm(P(A), A.stride(0), P(Wp[i % COPIES]), P(bias), P(res) if res is not None else None, D if res is not None else 0,
                              P(res) if res is not None else P(out), N, M, N, K, N, E.EPI[epi], None, 0.0, E._stream_ptr())
        assert rc == 0, rc
    line = f"{name:4s} N={N:5d} K={K:5d}:"
    for label, tune in (("default", (0, 0)), ("skinny", (-1, 0)), ("W4", (4, 0)), ("W8", (8, 0)), ("W16", (16, 0))):
        lib.isst_op_set_gemm_tuning(*tune)
        line += f"  {label} {timeit(g):6.2f} us"
    lib.isst_op_set_gemm_tuning(0, 0)
    if epi == "bias_res":
        def gl(i):
            g(i)
            rc = lib.isst_op_layernorm(P(res), P(lnw), P(lnb), P(xn), M, D, 1e-5, 0, E._stream_ptr()); assert rc == 0
        line += f" | gemm+LN {timeit(gl):6.2f} us"
        for ks in (2, 4, 8):
            if K % (256 * ks): continue
            slabs = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
            def sk(i):
                rc = lib.isst_op_gemm_splitk_layernorm(P(A), A.stride(0), P(Wp[i % COPIES]), P(bias), P(res), P(lnw), P(lnb), P(xn), P(slabs), M, N, K, ks, 1e-5, E._stream_ptr())
                assert rc == 0, rc
            line += f"  splitK{ks}+LNreduce {timeit(sk):6.2f} us"
    print(line, flush=True)
