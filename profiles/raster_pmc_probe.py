"""The tiled GEMM on the 64-stream prefill gate/up shape (1408 x 28672 x 4096), plain grid (argv 0) or XCD-aware rasterisation (argv 2): run under
rocprofv3 --pmc FETCH_SIZE to read the memory-side traffic per launch (algorithmic: 235 MB of weights + 11.5 MB of activations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = torch.device("cuda"); lib = E.load_library(); P = E._ptr
lib.isst_op_set_gemm_tuning(100000 + int(sys.argv[1]), 0)
M, N, K = 1408, 28672, 4096
p = E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16())
x = torch.randn(M, K, device=dev).bfloat16()
out = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
for i in range(8):
    rc = lib.isst_op_gemm(P(x), K, P(p), None, None, 0, P(out), N // 2, M, N, K, N // 2, E.EPI["swiglu"], None, 0.0, E._stream_ptr()); assert rc == 0
torch.cuda.synchronize()
