"""What would the launch-free residual + RMSNorm cost at 65..256 rows?  Its CONSUMER half exists -- gate/up / q/k/v through gemm_wide
normalising the staged rows (GemmArgs::ssq; the engine uses it at 49..64 rows) -- and is timed here at 128 and 256 rows against the plain launch on pre-normalised rows
and against the norm launch it would make unnecessary (cold rotating weights; every call also pays the wrapper's output allocation, the same in all columns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); E._lib = lib
def timeit(run, n=40):
    for i in range(4): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
K = 4096
for M in (128, 256):
    x = (torch.randn(M, K, device=dev) * 0.5).bfloat16()
    nw = (1 + 0.05 * torch.randn(K, device=dev)).bfloat16()
    ssq = x.float().view(M, K // 32, 32).pow(2).sum(-1).contiguous()
    xn = E.op_rmsnorm(x, nw)
    for name, N, epi in (("gate_up", 28672, "swiglu"), ("q/k/v  ", 6144, "none")):
        copies = max(3, (700 << 20) // (N * K * 2) + 1)
        packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        a = E.op_gemm(xn, packs[0], N, epi); b = E.op_gemm_norm_ssq(x, packs[0], N, nw, ssq, epi)
        same = bool((a == b).all())
        t_plain = timeit(lambda i: E.op_gemm(xn, packs[i % copies], N, epi))
        t_norm = timeit(lambda i: E.op_gemm_norm_ssq(x, packs[i % copies], N, nw, ssq, epi))
        t_rms = timeit(lambda i: E.op_rmsnorm(x, nw))
        print(f"M={M:3d} {name}: plain {t_plain:6.1f} us   normalising on stage {t_norm:6.1f} us ({t_norm - t_plain:+5.1f})   a norm launch alone {t_rms:5.1f} us   bit-identical {same}", flush=True)
        del packs
