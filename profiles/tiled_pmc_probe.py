import os, sys
sys.path.insert(0, "/root/repo")
import torch
from infinisst_amd import engine as E
dev = torch.device("cuda"); lib = E.load_library(); P = E._ptr
M, N, K = 1408, 28672, 4096
p = E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16())
x = torch.randn(M, K, device=dev).bfloat16()
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
for i in range(6):
    rc = lib.isst_op_gemm(P(x), K, P(p), None, None, 0, P(out), N // 2, M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr()); assert rc == 0
torch.cuda.synchronize()
