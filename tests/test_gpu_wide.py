"""gemm_wide.hip (65..256 rows, every weight byte read once: 4 waves with one wave per SIMD, statically indexed register rings for the weight
fragments and the activation lines) on a real MI355X, through the C ABI's isst_op_gemm / isst_op_gemm_splitk_rmsnorm: bit-identical to the path
those shapes ran on before (gemm_tiled.hip: same MFMA, same ascending K order per accumulator, same epilogue rounding points) and within
bf16 rounding of the fp32 reference arithmetic of the call sites it serves (patch_llm.py:260-262,334, HF LlamaMLP [3P], model/llm.py:236-262)."""
import numpy as np
import pytest
import torch

from infinisst_amd import engine as E
from oracle import llm as ollm
from test_gpu_kernels import bf, close_bf16, ref_linear

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _wide_mode(m, variant=0):
    E.load_library().isst_op_set_gemm_tuning(900000 + 10 * m + variant, 0)  # gemm_wide.hip: 0 never, 1 heuristic, 2 wherever it can run


@pytest.mark.parametrize("M,N,K", [(65, 256, 128), (100, 272, 256), (128, 512, 1536), (129, 1040, 512), (200, 96, 4096), (256, 1024, 2048), (255, 48, 192), (77, 2048, 896)])
def test_gemm_wide_is_bit_identical_to_gemm_tiled(M, N, K):
    """Ragged rows (rows past M read zeros through the descriptor), n-tiles past N, 2..64 K-steps incl. counts that are no multiple of the ring's unroll
    period (the padded iterations multiply zeros), every epilogue."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    res = bf(torch.randn(M, N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        for epi in ("none", "bias", "bias_gelu", "res", "bias_res", "f32"):
            kw = dict(bias=bias.to(DEV) if "bias" in epi else None, res=res.to(DEV) if "res" in epi else None)
            _wide_mode(0)
            want = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
            _wide_mode(2)
            got = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
            torch.cuda.synchronize()
            atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
            close_bf16(got, ref_linear(A, W, epi, bias, res), f"wide {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)
            if M > 128 or (N * K > (32 << 20)):  # (up to 128 rows the old path of a short weight stream is gemm_mid's two 64-row blocks: another K order)
                assert torch.equal(got, want), f"{epi} M{M} N{N} K{K}: {int((got != want).sum())} elements differ from gemm_tiled"
        if N % 32 == 0:
            _wide_mode(0)
            want = E.op_gemm(A.to(DEV), Wp, N, "swiglu")
            _wide_mode(2)
            got = E.op_gemm(A.to(DEV), Wp, N, "swiglu")
            if M > 128:
                assert torch.equal(got, want), f"swiglu M{M} N{N} K{K}"
            close_bf16(got, want, f"wide swiglu M{M} N{N} K{K}", ulps=4.0, atol=3.2e-2)
    finally:
        _wide_mode(1)


@pytest.mark.parametrize("M,N,K,ks", [(128, 1024, 2048, 4), (256, 512, 4096, 2), (130, 512, 512, 2), (100, 256, 1024, 8), (192, 4096, 1792, 4)])
def test_gemm_wide_split_k_slabs(M, N, K, ks):
    """K slices (EPI_PARTIAL: fp32 slabs) + the reducing residual / RMSNorm kernel against the same pair on gemm_tiled (bit-identical: the slices cut K at
    the same places) and against the oracle arithmetic x = bf16(x + bf16(A @ W^T)), LlamaRMSNorm(x)."""
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(M, N, generator=g))
    nw = bf(1 + 0.2 * torch.randn(N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        _wide_mode(0)
        x0, n0 = E.op_gemm_splitk_rmsnorm(A.to(DEV), Wp, x.to(DEV), ks, nw.to(DEV))
        _wide_mode(2)
        x1, n1 = E.op_gemm_splitk_rmsnorm(A.to(DEV), Wp, x.to(DEV), ks, nw.to(DEV))
        torch.cuda.synchronize()
    finally:
        _wide_mode(1)
    want = bf(x.float() + bf(A.float() @ W.float().t()).float())
    close_bf16(x1, want, f"wide split-K M{M} N{N} K{K} ks{ks}", ulps=2.5, atol=3.2e-2)
    close_bf16(n1, ollm.rmsnorm(x1.cpu(), nw, 1e-5), "norm of the updated rows", ulps=2.0, atol=1e-3)
    assert torch.equal(x0, x1) and torch.equal(n0, n1), f"{int((x0 != x1).sum())} elements differ from gemm_tiled's slabs"


@pytest.mark.parametrize("M,N,K,epi", [(33, 16384, 4096, "swiglu"), (64, 16400, 4096, "f32"), (48, 16384, 4096, "none"), (100, 4096, 4096, "swiglu"), (200, 1040, 2048, "f32")])
def test_gemm_wide_normalises_on_stage(M, N, K, epi):
    """The consumer half of the launch-free residual + RMSNorm on gemm_wide (round 4): A = x is normalised by the LOADER waves, in LDS, with 1/rms from
    the producer's per-32-column sums of squares and HF LlamaRMSNorm's rounding points [3P] -- gate/up and lm_head of a 33..64-stream decode pass take
    this form (the widest weight streams; rows 65+ through mode 2).  Against the oracle arithmetic bf16(w * bf16(x / rms)) @ W^T with the epilogue's
    rounding points, and against the unfused pair (RMSNorm launch + plain projection)."""
    g = torch.Generator().manual_seed(M + N + K)
    x = bf(torch.randn(M, K, generator=g) * 3.0)
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    nw = bf(1 + 0.2 * torch.randn(K, generator=g))
    ssq = (x.float() ** 2).view(M, K // 32, 32).sum(-1).contiguous()
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        _wide_mode(2)
        got = E.op_gemm_norm_ssq(x.to(DEV), Wp, N, nw.to(DEV), ssq.to(DEV), epi)
        _wide_mode(1)
        xn_dev = E.op_rmsnorm(x.to(DEV), nw.to(DEV))
        unfused = E.op_gemm(xn_dev, Wp, N, epi)
        torch.cuda.synchronize()
    finally:
        _wide_mode(1)
    xn = ollm.rmsnorm(x, nw, 1e-5)
    acc = xn.float() @ W.float().t()
    if epi == "swiglu":
        gg, uu = acc.view(M, N // 32, 2, 16)[:, :, 0].reshape(M, -1), acc.view(M, N // 32, 2, 16)[:, :, 1].reshape(M, -1)
        r = lambda t: t.to(torch.bfloat16).float()
        want = bf(r(torch.nn.functional.silu(r(gg))) * r(uu))
    elif epi == "f32":
        want = acc.to(torch.bfloat16).float()
    else:
        want = bf(acc)
    close_bf16(got, want, f"wide norm-on-stage {epi} M{M} N{N} vs oracle", ulps=3.0, atol=6e-2)
    close_bf16(got, unfused, f"wide norm-on-stage {epi} M{M} N{N} vs the unfused pair", ulps=3.0, atol=6e-2)
