"""Probe: the 64-row decode projections on gemm_mid (A staged through LDS chunk by chunk) against the skinny kernel with 4 / 8 n-tiles per workgroup
(A fragments straight from L2 into registers, no LDS, no barrier in the k-loop).  Weights rotate over copies.
NEEDS profiles/r02/gemm_wide_skinny_experiment.patch applied (the 600000+ntb tuning codes exist only there)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, N, K, epi in (("q/k/v", 6144, 4096, "none"), ("gate/up", 28672, 4096, "swiglu"), ("o_proj", 4096, 4096, "res"), ("down", 4096, 14336, "res")):
    copies = max(2, min(8, int(600e6 / (N * K * 2)) + 1))
    packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    x = torch.randn(M, K, device=dev).bfloat16()
    n_out = N // 2 if epi == "swiglu" else N
    res = torch.randn(M, n_out, device=dev).bfloat16() if epi == "res" else None
    out = torch.empty(M, n_out, device=dev, dtype=torch.bfloat16)
    line, ref = [], None
    for tag, flag in (("mid", 0), ("wide4", 4), ("wide8", 8)):
        lib.isst_op_set_gemm_tuning(600000 + flag, 0)
        def run(i):
            rc = lib.isst_op_gemm(P(x), K, P(packs[i % copies]), None, P(res), n_out, P(out), n_out, M, N, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr())
            assert rc == 0, rc
        for i in range(8): run(i)
        run(0); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        d = (out.float() - ref.float()).abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): run(i)
        e1.record(); torch.cuda.synchronize()
        line.append(f"{tag} {e0.elapsed_time(e1) / 100 * 1e3:6.2f} us (max |d| vs mid {d:.3g})")
    print(f"{name:8s} M={M} N={N:6d} K={K:6d}: " + "   ".join(line) + f"   weights alone at 6.6 TB/s {N * K * 2 / 6.6e6:5.2f} us", flush=True)
    del packs
lib.isst_op_set_gemm_tuning(600000, 0)
