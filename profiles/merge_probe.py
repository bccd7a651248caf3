"""Probe: the split-KV merge as a launch of its own + the o_proj GEMV, against the merge fused into the o_proj GEMV's A-staging prologue
(gemm.hip AMODE 3), Llama-3.1-8B decode shape (32 heads, 19 splits, N = K = 4096), weights rotating over copies.
    python profiles/merge_probe.py [waves per workgroup of the fused kernel ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(); P = E._ptr
H, S, K, N = 32, 19, 4096, 4096
copies = 8
packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
part = torch.randn(1, H, S, 132, device=dev)  # slab: O[128], max, sum, pad
part[..., 129] = part[..., 129].abs() + 1.0
res = torch.randn(1, N, device=dev).bfloat16()
attn = torch.empty(1, K, device=dev, dtype=torch.bfloat16)
out = torch.empty(1, N, device=dev, dtype=torch.bfloat16)


def timeit(fn, n=200):
    for i in range(16): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def separate(i):
    assert lib.isst_op_attn_combine(P(part), P(attn), H, 1, S, E._stream_ptr()) == 0
    assert lib.isst_op_gemm(P(attn), K, P(packs[i % copies]), None, P(res), N, P(out), N, 1, N, K, N, E.EPI["res"], None, 0.0, E._stream_ptr()) == 0


def gemv_only(i):
    assert lib.isst_op_gemm(P(attn), K, P(packs[i % copies]), None, P(res), N, P(out), N, 1, N, K, N, E.EPI["res"], None, 0.0, E._stream_ptr()) == 0


def fused(i):
    assert lib.isst_op_gemm_attn_merge(P(part), S, P(packs[i % copies]), P(res), N, P(out), N, 1, N, K, E._stream_ptr()) == 0


separate(0); ref = out.clone(); fused(0); torch.cuda.synchronize()
print("fused == separate (bits):", bool(torch.equal(ref, out)))
print(f"o_proj GEMV alone            {timeit(gemv_only):7.2f} us")
print(f"combine launch + o_proj GEMV {timeit(separate):7.2f} us")
for w in [int(a) for a in sys.argv[1:]] or [16, 8, 4]:
    lib.isst_op_set_gemm_tuning(300000 + w, 0)
    print(f"fused, {w:2d} waves              {timeit(fused):7.2f} us")
lib.isst_op_set_gemm_tuning(300000, 0)
