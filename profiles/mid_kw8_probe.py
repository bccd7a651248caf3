"""Probe: gemm_mid's 8-k-step-wave form (KSW = 1: twice the waves on the same 32- / 64-column tile) against the 4-wave form on the narrow
Llama projections at 22 / 48 / 64 rows.  o_proj and down run as the engine runs them: split-K slabs + the reducing residual/RMSNorm kernel.
Weights rotate over copies."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
lib = E.load_library(); dev = "cuda"; P = E._ptr
def timeit(fn, n=60):
    for i in range(6): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for M in (22, 48, 64):
    for name, N, K, ks in (("q/k/v", 6144, 4096, 0), ("o_proj", 4096, 4096, 2), ("down", 4096, 14336, 4), ("gate/up", 28672, 4096, -1)):
        copies = max(2, min(6, int(500e6 / (N * K * 2)) + 1))
        Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        A = torch.randn(M, K, device=dev).bfloat16()
        line = f"M={M:3d} {name:8s}:"
        outs = []
        for label, tune in (("4 k-waves", 64), ("8 k-waves", 0)):
            lib.isst_op_set_gemm_tuning(0, tune)
            if ks <= 0:
                epi = "swiglu" if ks < 0 else "none"
                n_out = N // 2 if ks < 0 else N
                out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
                def fn(i):
                    rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), None, None, 0, P(out), n_out, M, N, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr()); assert rc == 0
            else:
                torch.manual_seed(3); x0 = torch.randn(M, N, device=dev).bfloat16(); x = x0.clone(); nw = torch.ones(N, device=dev).bfloat16()
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); slabs = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
                def fn(i):
                    rc = lib.isst_op_gemm_splitk_rmsnorm(P(A), K, P(Wps[i % copies]), P(x), P(nw), P(out), P(slabs), M, N, K, ks, 1e-5, E._stream_ptr()); assert rc == 0
            t = timeit(fn)
            if ks > 0: x.copy_(x0)
            fn(0); torch.cuda.synchronize(); outs.append(out.float().clone())
            line += f"   {label} {t:6.2f} us"
        line += f"   max |d| {float((outs[0] - outs[1]).abs().max()):.3g}   (weights alone at 6.6 TB/s {N * K * 2 / 6.6e6:5.2f} us)"
        print(line, flush=True)
        del Wps
lib.isst_op_set_gemm_tuning(0, 0)
