"""gemm_mid at 64 rows on the q/k/v shape (N 6144, K 4096), a few launches -- run under rocprofv3 --pmc to see where its waves spend their cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = torch.device("cuda"); lib = E.load_library(); P = E._ptr
M, N, K = 64, 6144, 4096
ps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(4)]
x = torch.randn(M, K, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for i in range(12):
    rc = lib.isst_op_gemm(P(x), K, P(ps[i % 4]), None, None, 0, P(out), N, M, N, K, N, E.EPI["none"], None, 0.0, E._stream_ptr()); assert rc == 0
torch.cuda.synchronize()
