"""PMC probe (round 4): gate/up at 128 rows on gemm_wide (variant from VARIANT), 24 launches over rotating weight copies, for rocprofv3 --pmc passes.
Also 64 rows on gemm_mid for comparison (the same question: do the activation re-reads hit in L2 or go out over the fabric?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
N, K = 28672, 4096
copies = 4
packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
var = int(os.environ.get("VARIANT", "1"))
for M in (128, 64):
    A = torch.randn(M, K, device=dev).bfloat16()
    out = torch.zeros(M, N // 2, device=dev, dtype=torch.bfloat16)
    lib.isst_op_set_gemm_tuning(900000 + 20 + var, 0)
    for i in range(24):
        rc = lib.isst_op_gemm(P(A), K, P(packs[i % copies]), None, None, 0, P(out), out.stride(0), M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr())
        assert rc == 0
    torch.cuda.synchronize()
lib.isst_op_set_gemm_tuning(900000 + 10, 0)
