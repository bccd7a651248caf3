"""A few launches of gemm_dense.hip on the 64-stream prefill's gate/up shape (1408 x 28672 x 4096, SwiGLU), weights rotating over 3 copies -- run under
rocprofv3 to read the kernel's counters:
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 profiles/dense_pmc_probe.py            (fabric reads; x2 on gfx950)
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d OUT -- python3 profiles/dense_pmc_probe.py
profiles/dense_pmc_reduce.py turns the counter CSVs into profiles/rNN/dense_pmc.json."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinisst_amd import engine as E

lib = E.load_library()
dev = "cuda"
M, N, K = 1408, 28672, 4096
Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(3)]
A = torch.randn(M, K, device=dev).bfloat16()
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
for i in range(12):
    rc = lib.isst_op_gemm(E._ptr(A), K, E._ptr(Wps[i % 3]), None, None, 0, E._ptr(out), N // 2, M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr())
    assert rc == 0
torch.cuda.synchronize()
print("done")
