"""Per-kernel parity on a real MI355X: every HIP kernel entry point of the C ABI against the CPU oracle / the
same torch op the reference calls, on seeded inputs.  bf16 outputs must agree within one bf16 rounding of the
fp32 result (stated per test)."""
import numpy as np
import pytest
import torch

from infinisst_amd import engine as E
from infinisst_amd import synth
from infinisst_amd.config import toy_config
from oracle import generate as ogen
from oracle import llm as ollm
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def close_bf16(got, ref, what, ulps=2.0, atol=1e-6):
    """|got - ref| <= ulps * 2^-8 * |ref| + atol   (bf16 has 8 significant bits)."""
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, f"{what}: shape {got.shape} vs {ref.shape}"
    err = (got - ref).abs()
    tol = ulps * (2.0 ** -8) * ref.abs() + atol
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, worst {float(err.max()):.5g} at {np.unravel_index(int(err.argmax()), err.shape)}"


def ref_linear(A, W, epi, bias=None, res=None):
    acc = A.float() @ W.float().t()
    r = lambda t: t.to(torch.bfloat16).float()
    if epi == "none":
        return bf(acc)
    if epi == "bias":
        return bf(acc + bias.float())
    if epi == "bias_gelu":
        return bf(torch.nn.functional.gelu(r(acc + bias.float())))
    if epi == "res":
        return bf(res.float() + r(acc))
    if epi == "bias_res":
        return bf(res.float() + r(acc + bias.float()))
    if epi == "f32":
        return r(acc)
    raise ValueError(epi)


@pytest.mark.parametrize("M", [1, 3, 5, 8, 9, 16, 22, 48, 64, 100, 200])
@pytest.mark.parametrize("N,K", [(256, 128), (1040, 512), (64, 4096)])
def test_gemm_plain_and_epilogues(M, N, K):
    g = torch.Generator().manual_seed(M * 1000 + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    res = bf(torch.randn(M, N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    for epi in ("none", "bias", "bias_gelu", "res", "bias_res", "f32"):
        out = E.op_gemm(A.to(DEV), Wp, N, epi, bias=bias.to(DEV) if "bias" in epi else None,
                        res=res.to(DEV) if "res" in epi else None)
        torch.cuda.synchronize()
        # compound epilogues round twice (inner Linear output, then residual add / GELU): a one-ulp flip of the inner
        # rounding moves the result by an ulp of the INNER magnitude, hence the absolute slack
        atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
        close_bf16(out, ref_linear(A, W, epi, bias, res), f"gemm {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)


def test_gemm_random_shapes_across_every_kernel_boundary():
    """70 seeded random (rows, columns, depth, epilogue) draws -- rows clustered around the dispatch boundaries of launch_gemm (12 | 13: GEMV -> gemm_mid,
    64 | 65: -> the tiled / dense kernels, 128, 256, 320, 640: tile and heuristic edges), ragged column counts, depths from one 32-wide MFMA step up --
    through the one entry point the engine uses, each against the fp32 reference with the epilogue's rounding points."""
    rng = np.random.default_rng(20261003)
    edges = [1, 2, 8, 12, 13, 16, 17, 32, 33, 48, 63, 64, 65, 66, 127, 128, 129, 255, 256, 257, 319, 320, 321, 639, 640, 641, 1023, 1408]
    epis = ("none", "bias", "bias_gelu", "res", "bias_res", "f32", "swiglu")
    for trial in range(70):
        M = int(rng.choice(edges)) if trial % 3 else int(rng.integers(1, 900))
        N = 16 * int(rng.integers(1, 70))
        K = 32 * int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128]))
        epi = epis[trial % len(epis)]
        if epi == "swiglu":
            N = max(32, N - N % 32)
        g = torch.Generator().manual_seed(trial)
        A = bf(torch.randn(M, K, generator=g))
        W = bf(torch.randn(N, K, generator=g) * (0.4 / K ** 0.5))
        bias, res = bf(torch.randn(N, generator=g)), bf(torch.randn(M, N, generator=g))
        out = E.op_gemm(A.to(DEV), E.op_pack_weight(W.to(DEV)), N, epi, bias=bias.to(DEV) if "bias" in epi else None, res=res.to(DEV) if "res" in epi else None)
        torch.cuda.synchronize()
        if epi == "swiglu":  # packed rows alternate (gate tile, up tile): output column j of tile pair p = silu(gate) * up (HF LlamaMLP, patch_llm.py call site)
            acc = (A.float() @ W.float().t()).view(M, N // 32, 2, 16)
            r = lambda t: t.to(torch.bfloat16).float()
            want = bf(r(torch.nn.functional.silu(r(acc[:, :, 0]))) * r(acc[:, :, 1])).reshape(M, N // 2)
            close_bf16(out, want, f"trial {trial}: swiglu M{M} N{N} K{K}", ulps=4.0, atol=3.2e-2)
        else:
            atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
            close_bf16(out, ref_linear(A, W, epi, bias, res), f"trial {trial}: {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)


def test_gemm_asymmetric_identity():
    """A = I against an asymmetric W catches swapped fragment maps (cdna_hip_programming.md section 3)."""
    K = N = 64
    W = bf(torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 100)
    A = bf(torch.eye(K))[:48]
    out = E.op_gemm(A.to(DEV), E.op_pack_weight(W.to(DEV)), N, "none")
    assert torch.equal(out.float().cpu(), W.float().t()[:48])


def test_gemm_odd_vocab_f32():
    g = torch.Generator().manual_seed(5)
    A, W = bf(torch.randn(3, 256, generator=g)), bf(torch.randn(1031, 256, generator=g) * 0.05)
    out = E.op_gemm(A.to(DEV), E.op_pack_weight(W.to(DEV)), 1031, "f32")
    assert out.shape == (3, 1031)
    close_bf16(out, ref_linear(A, W, "f32"), "lm_head-like", ulps=2.5, atol=2e-3)


@pytest.mark.parametrize("M", [1, 7, 33, 70])
def test_gemm_swiglu(M):
    g = torch.Generator().manual_seed(M)
    I, K = 512, 256
    A = bf(torch.randn(M, K, generator=g))
    Wg, Wu = bf(torch.randn(I, K, generator=g) * 0.1), bf(torch.randn(I, K, generator=g) * 0.1)
    # interleave gate/up 16-row tiles the way the engine packs them
    inter = torch.stack([Wg.view(I // 16, 16, K), Wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K)
    out = E.op_gemm(A.to(DEV), E.op_pack_weight(inter.to(DEV)), 2 * I, "swiglu")
    gg, uu = bf(A.float() @ Wg.float().t()), bf(A.float() @ Wu.float().t())
    ref = torch.nn.functional.silu(gg) * uu
    close_bf16(out, ref, f"swiglu M{M}", ulps=3, atol=2e-3)


@pytest.mark.parametrize("M", [1, 2, 4, 8])
@pytest.mark.parametrize("norm", [False, True])
@pytest.mark.parametrize("I,K", [(512, 256), (14336, 4096)])
def test_gemm_swiglu_self_paired_tiles_are_bit_identical_to_tile_pairs(M, norm, I, K):
    """gate / up as SELF-PAIRED tiles (isst_op_pack_gateup8 + epi swiglu8: tile t = gate rows 8t..8t+7 | up rows 8t..8t+7; the copy a one-row decode pass streams,
    engine_llm.hip) against the tile-pair form (epi swiglu) on the same weights: the same bits, with and without the fused RMSNorm, toy and Llama-3.1-8B widths --
    both forms run the register-A GEMV with the same waves per workgroup at one and two rows, i.e. one summation order per output -- and the oracle's HF LlamaMLP value at every row count."""
    g = torch.Generator().manual_seed(M + I)
    A = bf(torch.randn(M, K, generator=g))
    nw = bf(1 + 0.2 * torch.randn(K, generator=g))
    Wg, Wu = bf(torch.randn(I, K, generator=g) * 0.05), bf(torch.randn(I, K, generator=g) * 0.05)
    inter = torch.stack([Wg.view(I // 16, 16, K), Wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K)
    kw = dict(norm_w=nw.to(DEV), norm_eps=1e-5) if norm else {}
    pairs = E.op_gemm(A.to(DEV), E.op_pack_weight(inter.to(DEV)), 2 * I, "swiglu", **kw)
    self8 = E.op_gemm(A.to(DEV), E.op_pack_gateup8(Wg.to(DEV), Wu.to(DEV)), 2 * I, "swiglu8", **kw)
    torch.cuda.synchronize()
    assert self8.shape == pairs.shape == (M, I)
    if M <= 2:  # (from 3 rows on the fused-norm launcher gives one-tile workgroups 8 waves and tile pairs 4: another order of the same sum; the engine takes this form at ONE row)
        assert torch.equal(self8.view(torch.int16), pairs.view(torch.int16)), f"M{M} norm {norm} I{I}: max |d| {float((self8.float() - pairs.float()).abs().max())}"
    x = ollm.rmsnorm(A, nw, 1e-5) if norm else A
    ref = torch.nn.functional.silu(bf(x.float() @ Wg.float().t())) * bf(x.float() @ Wu.float().t())
    close_bf16(self8, bf(ref.float()), f"swiglu8 vs oracle M{M} norm {norm} I{I}", ulps=3, atol=4e-3 if K <= 512 else 2e-2)


@pytest.mark.parametrize("M", [1, 2, 5, 8])
@pytest.mark.parametrize("epi", ["none", "f32", "swiglu"])
def test_gemm_fused_rmsnorm(M, epi):
    """RMSNorm applied inside the A-fragment load against the ORACLE end to end -- oracle LlamaRMSNorm (pinned to HF's class,
    tests/golden/llama_blocks.npz), then the projection in fp32 with the reference's bf16 rounding points -- and against the two-step HIP
    path (oracle norm, then the plain HIP projection)."""
    g = torch.Generator().manual_seed(M + len(epi))
    K, N = 512, 256
    x = bf(torch.randn(M, K, generator=g) * 2)
    nw = bf(1 + 0.2 * torch.randn(K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    if epi == "swiglu":
        W = torch.stack([W[: N // 2].view(N // 32, 16, K), W[N // 2:].view(N // 32, 16, K)], dim=1).reshape(N, K)
    Wp = E.op_pack_weight(W.to(DEV))
    fused = E.op_gemm(x.to(DEV), Wp, N, epi, norm_w=nw.to(DEV), norm_eps=1e-5)
    xn = ollm.rmsnorm(x, nw, 1e-5)
    two_step = E.op_gemm(xn.to(DEV), Wp, N, epi)
    torch.cuda.synchronize()
    close_bf16(fused, two_step, f"fused norm {epi} M{M}", ulps=2.5, atol=4e-3)
    if epi == "swiglu":
        Wg = W.view(N // 32, 2, 16, K)[:, 0].reshape(N // 2, K)
        Wu = W.view(N // 32, 2, 16, K)[:, 1].reshape(N // 2, K)
        ref = torch.nn.functional.silu(bf(xn.float() @ Wg.float().t())) * bf(xn.float() @ Wu.float().t())
        close_bf16(fused, bf(ref.float()), f"fused norm swiglu vs oracle M{M}", ulps=3, atol=4e-3)
    else:
        ref = bf(xn.float() @ W.float().t())
        close_bf16(fused, ref, f"fused norm {epi} vs oracle M{M}", ulps=2.5, atol=4e-3)


@pytest.mark.parametrize("k,stride,T", [(3, 2, 157), (2, 2, 48)])
def test_conv1d_as_gemm(k, stride, T):
    """Conv1d (no padding) over time-major activations == GEMM with overlapping rows (lda = stride*C)."""
    Cc = 64
    g = torch.Generator().manual_seed(k * 10 + T)
    x = bf(torch.randn(T, Cc, generator=g))
    w = bf(torch.randn(Cc, Cc, k, generator=g) * 0.1)
    bias = bf(torch.randn(Cc, generator=g))
    To = (T - k) // stride + 1
    out = E.op_gemm(x.to(DEV), E.op_pack_weight(w.to(DEV), conv_k=k), Cc, "bias", bias=bias.to(DEV), lda=stride * Cc, M=To,
                    K=k * Cc)
    ref = torch.nn.functional.conv1d(x.t().unsqueeze(0).float(), w.float(), bias.float(), stride=stride)[0].t()
    close_bf16(out, bf(ref), f"conv k{k}", ulps=2.5, atol=2e-3)


@pytest.mark.parametrize("M", [1, 4, 15, 20, 37, 64])
@pytest.mark.parametrize("N,K", [(48, 160), (272, 96), (16, 32)])
def test_gemm_skinny_ring_edges_and_poison(M, N, K):
    """The skinny kernel's weight / A ring issues loads past the end of K unconditionally (bounds-checked buffer loads that must
    read zeros): K with a partial last stage, fewer k-tiles than the ring is deep, and NaN-filled memory right behind the A rows and
    the packed weight must not reach any output row < M."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    rows = M + 40
    Abig = torch.full((rows, K), float("nan"), dtype=torch.bfloat16)
    Abig[:M] = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    res = bf(torch.randn(M, N, generator=g))
    Wp_exact = E.op_pack_weight(W.to(DEV))
    Wbig = torch.full((Wp_exact.numel() + 4096,), float("nan"), dtype=torch.bfloat16, device=DEV)
    Wbig[:Wp_exact.numel()] = Wp_exact.reshape(-1)
    Wp = Wbig[:Wp_exact.numel()].view_as(Wp_exact)
    A = Abig.to(DEV)[:M]
    E.load_library().isst_op_set_gemm_tuning(-1, 0)  # the skinny kernel at every M
    try:
        for epi in ("none", "res"):
            out = E.op_gemm(A, Wp, N, epi, res=res.to(DEV) if epi == "res" else None)
            torch.cuda.synchronize()
            assert not torch.isnan(out.float()).any(), f"NaN leaked into {epi} M{M} N{N} K{K}"
            close_bf16(out, ref_linear(Abig[:M], W, epi, None, res), f"ring edges {epi} M{M} N{N} K{K}", ulps=2.5, atol=2e-3 if epi == "none" else 3.2e-2)
    finally:
        E.load_library().isst_op_set_gemm_tuning(0, 0)


@pytest.mark.parametrize("T", [9, 41, 100])
def test_conv1d_as_gemm_few_rows_overlapping(T):
    """k=3, stride 2 with few output rows: the skinny kernel reads A through a descriptor whose rows overlap (lda < K)."""
    Cc, k, stride = 64, 3, 2
    g = torch.Generator().manual_seed(T)
    x = bf(torch.randn(T, Cc, generator=g))
    w = bf(torch.randn(Cc, Cc, k, generator=g) * 0.1)
    bias = bf(torch.randn(Cc, generator=g))
    To = (T - k) // stride + 1
    out = E.op_gemm(x.to(DEV), E.op_pack_weight(w.to(DEV), conv_k=k), Cc, "bias", bias=bias.to(DEV), lda=stride * Cc, M=To, K=k * Cc)
    ref = torch.nn.functional.conv1d(x.t().unsqueeze(0).float(), w.float(), bias.float(), stride=stride)[0].t()
    close_bf16(out, bf(ref), f"conv k3 T{T}", ulps=2.5, atol=2e-3)


@pytest.mark.parametrize("C,gelu", [(64, True), (512, True), (512, False), (1024, False), (1024, True), (128, False)])
def test_layernorm(C, gelu):
    g = torch.Generator().manual_seed(C)
    x = bf(torch.randn(37, C, generator=g) * 2 + 0.3)
    w, b = bf(1 + 0.1 * torch.randn(C, generator=g)), bf(0.1 * torch.randn(C, generator=g))
    out = E.op_layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, gelu)
    ref = torch.nn.functional.layer_norm(x, (C,), w, b, 1e-5)
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    close_bf16(out, ref, f"layernorm C{C}", ulps=2.0, atol=1e-3)


@pytest.mark.parametrize("D", [256, 4096])
def test_rmsnorm(D):
    g = torch.Generator().manual_seed(D)
    x = bf(torch.randn(9, D, generator=g) * 3)
    w = bf(1 + 0.1 * torch.randn(D, generator=g))
    out = E.op_rmsnorm(x.to(DEV), w.to(DEV), 1e-5)
    close_bf16(out, ollm.rmsnorm(x, w, 1e-5), f"rmsnorm D{D}", ulps=2.0, atol=1e-3)


@pytest.mark.parametrize("C,n_samples", [(64, 399 + 5120), (512, 399 + 5120), (512, 5 * 33001 + 7), (512, 10 + 5 * 2)])
def test_conv0_ln_gelu(C, n_samples):
    """(512 channels: the register-resident form, 4 frames per wave below 32768 frames and 16 from there on; frame counts that are no multiple of either)"""
    cfg = toy_config().replace(conv_layers=[(C, 10, 5)])
    w = synth.random_weights(cfg.replace(enc_layers=0, llm_layers=0), dtype=torch.bfloat16, std=0.3, norm_jitter=0.1, seed=C)
    audio = bf(torch.from_numpy(synth.synthetic_audio(n_samples)))
    p = oenc.ENC + "feature_extractor.conv_layers.0."
    out = E.op_conv0(audio.to(DEV), w[p + "0.weight"].to(DEV), w[p + "0.bias"].to(DEV), w[p + "2.1.weight"].to(DEV),
                     w[p + "2.1.bias"].to(DEV), 10, 5)
    ref = oenc.conv_feature_extractor(w, cfg, audio.unsqueeze(0))[0].t()
    close_bf16(out, ref, f"conv0 C{C}", ulps=3.0, atol=4e-3)


def test_sample_processors_and_argmax():
    rng = np.random.default_rng(0)
    V = 5000
    for trial in range(20):
        logits = torch.from_numpy(rng.standard_normal(V).astype(np.float32)).bfloat16().float()
        n_ids = int(rng.integers(6, 40))
        ids = rng.integers(0, 50, size=n_ids).tolist()  # small alphabet -> repeated n-grams
        enc = rng.integers(0, 50, size=int(rng.integers(0, 60))).tolist()
        sup = rng.integers(0, V, size=int(rng.integers(0, 5))).tolist()
        n = int(rng.integers(2, 5))
        if trial % 3 == 0:  # force a tie: lowest index must win
            logits[100] = logits[4000] = 50.0
        ref_scores = ogen.process_logits(logits.clone(), ids, enc, 1.2, n, n, sup)
        ref_tok = int(torch.argmax(ref_scores))
        dl = logits.clone().to(DEV)
        tok = E.op_sample(dl, ids, enc, sup, 1.2, n, n)
        assert tok == ref_tok, f"trial {trial}: {tok} vs {ref_tok}"
        got = dl.cpu()
        assert torch.equal(torch.isinf(got), torch.isinf(ref_scores))
        fin = ~torch.isinf(ref_scores)
        assert torch.allclose(got[fin], ref_scores[fin], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("M,N,K", [(130, 256, 128), (1574, 512, 1536), (700, 1040, 512), (257, 96, 192)])
def test_gemm_tiled_dense_shapes(M, N, K):
    """The LDS-tiled kernel (M > 64): every epilogue, ragged M/N edges, against the fp32 reference."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias, res = bf(torch.randn(N, generator=g)), bf(torch.randn(M, N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    for epi in ("none", "bias", "bias_gelu", "res", "bias_res", "f32"):
        out = E.op_gemm(A.to(DEV), Wp, N, epi, bias=bias.to(DEV) if "bias" in epi else None, res=res.to(DEV) if "res" in epi else None)
        atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
        close_bf16(out, ref_linear(A, W, epi, bias, res), f"tiled gemm {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)
    # the skinny kernel on the same problem must agree with the tiled one to fp32 summation order
    lib = E.load_library()
    tiled = E.op_gemm(A.to(DEV), Wp, N, "none")
    lib.isst_op_set_gemm_tuning(-1, 0)
    try:
        skinny = E.op_gemm(A.to(DEV), Wp, N, "none")
    finally:
        lib.isst_op_set_gemm_tuning(0, 0)
    close_bf16(tiled, skinny, "tiled vs skinny", ulps=2.5, atol=4e-3)  # one bf16 rounding flip from the different fp32 summation order


def test_gemm_tiled_swiglu():
    g = torch.Generator().manual_seed(77)
    M, I, K = 300, 512, 256
    A = bf(torch.randn(M, K, generator=g))
    Wg, Wu = bf(torch.randn(I, K, generator=g) * 0.1), bf(torch.randn(I, K, generator=g) * 0.1)
    inter = torch.stack([Wg.view(I // 16, 16, K), Wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K)
    out = E.op_gemm(A.to(DEV), E.op_pack_weight(inter.to(DEV)), 2 * I, "swiglu")
    gg, uu = bf(A.float() @ Wg.float().t()), bf(A.float() @ Wu.float().t())
    close_bf16(out, torch.nn.functional.silu(gg) * uu, "tiled swiglu", ulps=3, atol=2e-3)


@pytest.mark.parametrize("M", [13, 16, 17, 22, 40, 64])
@pytest.mark.parametrize("wn", [2, 4, 8])
def test_gemm_mid_rows_both_widths(M, wn):
    """13..64 rows go through gemm_mid.hip (A staged in LDS); all workgroup widths, every epilogue, ragged N; and the
    result must agree with the 1..16-row kernel's arithmetic (same products, different fp32 summation order)."""
    lib = E.load_library()
    g = torch.Generator().manual_seed(M * 10 + wn)
    try:
        for N, K in ((1040, 512), (256, 1024), (48, 128)):
            A = bf(torch.randn(M, K, generator=g))
            W = bf(torch.randn(N, K, generator=g) * 0.05)
            bias = bf(torch.randn(N, generator=g))
            res = bf(torch.randn(M, N, generator=g))
            Wp = E.op_pack_weight(W.to(DEV))
            for epi in ("none", "bias", "bias_gelu", "res", "bias_res", "f32"):
                kw = dict(bias=bias.to(DEV) if "bias" in epi else None, res=res.to(DEV) if "res" in epi else None)
                lib.isst_op_set_gemm_tuning(0, wn)
                out = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
                lib.isst_op_set_gemm_tuning(-1, 0)
                skinny = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
                torch.cuda.synchronize()
                atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
                close_bf16(out, ref_linear(A, W, epi, bias, res), f"mid {epi} M{M} N{N} K{K} wn{wn}", ulps=2.5, atol=atol)
                close_bf16(out, skinny, f"mid vs skinny {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)
        # SwiGLU pairs
        I, K = 512, 256
        A = bf(torch.randn(M, K, generator=g))
        Wg, Wu = bf(torch.randn(I, K, generator=g) * 0.1), bf(torch.randn(I, K, generator=g) * 0.1)
        inter = torch.stack([Wg.view(I // 16, 16, K), Wu.view(I // 16, 16, K)], dim=1).reshape(2 * I, K)
        lib.isst_op_set_gemm_tuning(0, wn)
        out = E.op_gemm(A.to(DEV), E.op_pack_weight(inter.to(DEV)), 2 * I, "swiglu")
        ref = torch.nn.functional.silu(bf(A.float() @ Wg.float().t())) * bf(A.float() @ Wu.float().t())
        close_bf16(out, ref, f"mid swiglu M{M} wn{wn}", ulps=3, atol=2e-3)
    finally:
        lib.isst_op_set_gemm_tuning(0, 0)


@pytest.mark.parametrize("M,N,K,ks", [(13, 256, 1024, 2), (16, 512, 2048, 4), (22, 256, 1024, 2), (64, 512, 2048, 4), (33, 256, 4096, 8), (17, 4096, 1024, 1),
                                      (88, 512, 2048, 8), (200, 256, 1024, 4), (130, 1040, 512, 2)])  # > 64 rows: the dense kernel's K slices
@pytest.mark.parametrize("with_norm", [True, False])
def test_gemm_splitk_rmsnorm(M, N, K, ks, with_norm):
    """o_proj / down_proj at 17..64 rows: K split over workgroups into fp32 slabs, reduced by the residual + RMSNorm kernel.
    Reference = HF LlamaDecoderLayer: hidden = residual + Linear(x) (bf16 each), then LlamaRMSNorm."""
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(M, N, generator=g))
    nw = bf(1 + 0.2 * torch.randn(N, generator=g))
    x_new, normed = E.op_gemm_splitk_rmsnorm(A.to(DEV), E.op_pack_weight(W.to(DEV)), x.to(DEV), ks, nw.to(DEV) if with_norm else None, 1e-5)
    torch.cuda.synchronize()
    ref_x = ref_linear(A, W, "res", res=x)
    close_bf16(x_new, ref_x, f"splitk x M{M} N{N} K{K} S{ks}", ulps=2.5, atol=3.2e-2)
    if with_norm:
        # the norm is checked on the kernel's own x (a 1-ulp difference in x is amplified by nothing, but keep it exact)
        close_bf16(normed, ollm.rmsnorm(x_new.cpu(), nw, 1e-5), f"splitk norm M{M}", ulps=2.0, atol=1e-3)
    else:
        assert normed is None


@pytest.mark.parametrize("N,K,ks", [(4096, 2048, 4), (4096, 4096, 8), (1024, 2048, 2), (1040, 512, 2), (2048, 2048, 8), (4096, 1024, 1)])
def test_residual_rmsnorm_reduce_all_loads_first_carries_the_stepwise_kernels_bits(N, K, ks):
    """The residual + RMSNorm reduce asks for every load of a row before its first add and stores nothing before the last one has landed
    (rowops.hip rmsnorm_reduce_lf_kernel; the stepwise kernel remains for 5..8 slices above 256 rows and for D > 4096).  Same slabs, same
    arithmetic, same order: both forms (isst_op_set_reduce_tuning) carry the same bits, x and normalised rows, at every row count; held to the oracle."""
    g = torch.Generator().manual_seed(N + K + ks)
    A = bf(torch.randn(300, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(300, N, generator=g))
    nw = bf(1 + 0.2 * torch.randn(N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    lib = E.load_library()
    try:
        for M in (300, 256, 131, 66, 22):
            lib.isst_op_set_reduce_tuning(0, 0)
            x_st, n_st = E.op_gemm_splitk_rmsnorm(A[:M].to(DEV), Wp, x[:M].to(DEV), ks, nw.to(DEV), 1e-5)
            xp_st, none = E.op_gemm_splitk_rmsnorm(A[:M].to(DEV), Wp, x[:M].to(DEV), ks, None, 1e-5)
            lib.isst_op_set_reduce_tuning(1 << 30, 1 << 30)
            x_lf, n_lf = E.op_gemm_splitk_rmsnorm(A[:M].to(DEV), Wp, x[:M].to(DEV), ks, nw.to(DEV), 1e-5)
            xp_lf, none_lf = E.op_gemm_splitk_rmsnorm(A[:M].to(DEV), Wp, x[:M].to(DEV), ks, None, 1e-5)
            torch.cuda.synchronize()
            assert none is None and none_lf is None
            assert torch.equal(x_lf.view(torch.int16), x_st.view(torch.int16)), f"x bits M{M} N{N} K{K} S{ks}"
            assert torch.equal(n_lf.view(torch.int16), n_st.view(torch.int16)), f"norm bits M{M} N{N} K{K} S{ks}"
            assert torch.equal(xp_lf.view(torch.int16), x_st.view(torch.int16)) and torch.equal(xp_st.view(torch.int16), x_st.view(torch.int16)), f"no-norm x bits M{M}"
            close_bf16(x_lf, ref_linear(A[:M], W, "res", res=x[:M]), f"reduce x M{M} N{N} K{K} S{ks}", ulps=2.5, atol=3.2e-2)
            close_bf16(n_lf, ollm.rmsnorm(x_lf.cpu(), nw, 1e-5), f"reduce norm M{M} N{N}", ulps=2.0, atol=1e-3)
    finally:
        lib.isst_op_set_reduce_tuning(-1, -1)


@pytest.mark.parametrize("M,N,K,ks,N2,epi2", [(13, 256, 1024, 2, 512, "none"), (22, 4096, 4096, 2, 1024, "swiglu"), (64, 4096, 2048, 4, 768, "none"),
                                              (64, 512, 2048, 4, 512, "f32"), (48, 1024, 14336, 4, 2048, "swiglu"), (33, 256, 4096, 8, 256, "none")])
def test_launch_free_residual_rmsnorm_equals_the_reduce_launch(M, N, K, ks, N2, epi2):
    """13..64 rows without a residual + RMSNorm launch (gemm_mid.hip): the last K-slice workgroup of a column block reduces the slabs, writes x and the
    per-(row, 32 columns) sums of squares; the next projection normalises its rows while it stages them.  x must carry the SAME BITS as the
    reduce launch writes (same arithmetic, same slice order); the second projection is held to the oracle's RMSNorm + Linear on that x and to the
    unfused pair; the counters must be re-armed (second call on the same tickets gives the same bits)."""
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(M, N, generator=g))
    nw = bf(1 + 0.2 * torch.randn(N, generator=g))
    W2 = bf(torch.randn(N2, N, generator=g) * 0.05)
    Wp, W2p = E.op_pack_weight(W.to(DEV)), E.op_pack_weight(W2.to(DEV))
    x_ref, normed_ref = E.op_gemm_splitk_rmsnorm(A.to(DEV), Wp, x.to(DEV), ks, nw.to(DEV), 1e-5)
    x_new, ssq, tickets = E.op_gemm_splitk_fused(A.to(DEV), Wp, x.to(DEV), ks)
    torch.cuda.synchronize()
    assert torch.equal(x_new, x_ref), f"x differs from the reduce launch: max |d| {float((x_new.float() - x_ref.float()).abs().max())}"
    assert int(tickets.abs().sum()) == 0, "arrival counters not re-armed"
    ssq_ref = (x_new.float() ** 2).reshape(M, N // 32, 32).sum(-1)
    assert torch.allclose(ssq, ssq_ref, rtol=1e-5, atol=1e-6), float((ssq - ssq_ref).abs().max())
    x_again, ssq_again, _ = E.op_gemm_splitk_fused(A.to(DEV), Wp, x.to(DEV), ks, tickets=tickets)
    assert torch.equal(x_again, x_new) and torch.equal(ssq_again, ssq)
    out = E.op_gemm_norm_ssq(x_new, W2p, N2, nw.to(DEV), ssq, epi2, 1e-5)
    ref2 = E.op_gemm(normed_ref, W2p, N2, epi2)  # the unfused pair: norm launch output -> plain projection
    torch.cuda.synchronize()
    xn = ollm.rmsnorm(x_new.cpu(), nw, 1e-5)
    if epi2 == "swiglu":  # W2's 16-row tiles alternate gate, up (the engine's packing)
        Wg = W2.view(N2 // 32, 2, 16, N)[:, 0].reshape(N2 // 2, N)
        Wu = W2.view(N2 // 32, 2, 16, N)[:, 1].reshape(N2 // 2, N)
        want = bf((torch.nn.functional.silu(bf(xn.float() @ Wg.float().t())) * bf(xn.float() @ Wu.float().t())).float())
    else:
        want = bf(xn.float() @ W2.float().t())
    close_bf16(out, want, f"norm-on-stage {epi2} M{M} N{N} vs oracle", ulps=3.0, atol=6e-2)
    close_bf16(out, ref2, f"norm-on-stage {epi2} M{M} N{N} vs the unfused pair", ulps=3.0, atol=6e-2)
    if epi2 == "none" and N % 512 == 0:  # the q/k/v form: the second projection itself in two K slices, reduced in the launch, rows normalised while staged
        out2 = E.op_gemm_splitk_plain(x_new, W2p, N2, 2, nw.to(DEV), ssq, 1e-5)
        close_bf16(out2, want, f"split-K plain + norm-on-stage M{M} N{N} vs oracle", ulps=3.0, atol=6e-2)
        out3 = E.op_gemm_splitk_plain(normed_ref, W2p, N2, 2)  # without the norm: bf16(sum of two slices) against the unsplit launch
        close_bf16(out3, ref2, f"split-K plain M{M} N{N} vs the unsplit launch", ulps=2.0, atol=3e-2)


def _dense_mode(m):
    E.load_library().isst_op_set_gemm_tuning(800000 + m, 0)  # gemm_dense.hip: 0 never, 1 heuristic, 2 wherever it can run


@pytest.mark.parametrize("M,N,K", [(65, 256, 128), (130, 256, 256), (257, 512, 1536), (700, 1040, 512), (1408, 1024, 2048), (300, 96, 192), (513, 272, 4096)])
def test_gemm_dense_is_bit_identical_to_gemm_tiled(M, N, K):
    """gemm_dense.hip (256 x 256 tile, 8-wave ping-pong, LDS-DMA staging, LDS-staged bf16 epilogue) against gemm_tiled.hip (128 x 128) -- the kernel
    every dense shape ran on before -- on ragged row / column counts (rows past M, n-tiles past N, partial last tiles), 2 .. 64 K-tiles, every
    epilogue: the same MFMA shape and the same ascending K order per accumulator, so every output bit must agree; and both against the oracle
    arithmetic (reference call sites: patch_llm.py:260-262,334, HF LlamaMLP, patch_speech_encoder.py:741-743,923,586-589)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    res = bf(torch.randn(M, N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        for epi in ("none", "bias", "bias_gelu", "res", "bias_res", "f32"):
            kw = dict(bias=bias.to(DEV) if "bias" in epi else None, res=res.to(DEV) if "res" in epi else None)
            _dense_mode(0)
            want = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
            _dense_mode(2)
            got = E.op_gemm(A.to(DEV), Wp, N, epi, **kw)
            torch.cuda.synchronize()
            assert torch.equal(got, want), f"{epi} M{M} N{N} K{K}: {int((got != want).sum())} elements differ from gemm_tiled"
            atol = 2e-3 if epi in ("none", "bias", "f32") else 3.2e-2
            close_bf16(got, ref_linear(A, W, epi, bias, res), f"dense {epi} M{M} N{N} K{K}", ulps=2.5, atol=atol)
        if N % 32 == 0:
            _dense_mode(0)
            want = E.op_gemm(A.to(DEV), Wp, N, "swiglu")
            _dense_mode(2)
            got = E.op_gemm(A.to(DEV), Wp, N, "swiglu")
            assert torch.equal(got, want), f"swiglu M{M} N{N} K{K}"
    finally:
        _dense_mode(1)


@pytest.mark.parametrize("M,N,K,lda,off", [(300, 512, 256, 448, 64), (1000, 272, 512, 512 + 8, 8), (200, 256, 384, 128, 0)])
def test_gemm_dense_strided_and_overlapping_rows(M, N, K, lda, off):
    """A given as a window of a wider matrix (lda > K, base not at a row start) and as OVERLAPPING rows (lda < K: the implicit-GEMM form of a strided
    convolution, patch_speech_encoder.py:245-251 / model/speech_encoder.py:233): the LDS-DMA staging addresses rows by lda like every other kernel."""
    g = torch.Generator().manual_seed(M + N + K + lda)
    flat = bf(torch.randn(off + (M - 1) * lda + K + 64, generator=g)).to(DEV)
    A = torch.as_strided(flat, (M, K), (lda, 1), off)
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        _dense_mode(0)
        want = E.op_gemm(A, Wp, N, "bias", bias=bias.to(DEV))
        _dense_mode(2)
        got = E.op_gemm(A, Wp, N, "bias", bias=bias.to(DEV))
        torch.cuda.synchronize()
    finally:
        _dense_mode(1)
    assert torch.equal(got, want), f"{int((got != want).sum())} elements differ from gemm_tiled"
    close_bf16(got, ref_linear(A.cpu().contiguous(), W, "bias", bias, None), f"dense strided M{M} N{N} K{K} lda{lda}", ulps=2.5, atol=2e-3)


@pytest.mark.parametrize("M,N,K,ks", [(384, 1024, 2048, 4), (1408, 512, 4096, 2), (130, 512, 512, 2), (700, 256, 1024, 8)])
def test_gemm_dense_split_k_slabs(M, N, K, ks):
    """K slices of the dense kernel (EPI_PARTIAL: fp32 slabs) + the reducing residual / RMSNorm kernel, against the same pair on gemm_tiled (bit-identical:
    the slices cut K at the same places) and against the oracle arithmetic x = bf16(x + bf16(A @ W^T)), LlamaRMSNorm(x)."""
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(M, N, generator=g))
    nw = bf(1 + 0.2 * torch.randn(N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        _dense_mode(0)
        x0, n0 = E.op_gemm_splitk_rmsnorm(A.to(DEV), Wp, x.to(DEV), ks, nw.to(DEV))
        _dense_mode(2)
        x1, n1 = E.op_gemm_splitk_rmsnorm(A.to(DEV), Wp, x.to(DEV), ks, nw.to(DEV))
        assert torch.equal(x0, x1) and torch.equal(n0, n1)
    finally:
        _dense_mode(1)
    want = bf(x.float() + bf(A.float() @ W.float().t()).float())
    close_bf16(x1, want, f"dense split-K M{M} N{N} K{K} ks{ks}", ulps=2.5, atol=3.2e-2)
    close_bf16(n1, ollm.rmsnorm(x1.cpu(), nw, 1e-5), "norm of the updated rows", ulps=2.0, atol=1e-3)


def _dense_mix(mix):
    E.load_library().isst_op_set_gemm_tuning(800000 + 2 + 10 * mix, 0)  # gemm_dense.hip wherever it can run; tile mix: 0 = the model, 1 = 256-row tiles only, 2 = 128-row only, 100 + h


@pytest.mark.parametrize("M,N,K", [(1408, 1024, 1024), (700, 2304, 512), (129, 256, 256), (1000, 272, 192)])
def test_gemm_dense_tile_mixes_are_bit_identical(M, N, K):
    """Round 6: one launch may hold row blocks of 256 rows (8-wave ping-pong over four quadrants) AND of 128 rows (two quadrants per wave, three K-tile buffers),
    dealt to the XCDs longest first through a folded 1-D grid.  Every mix -- the launcher's own choice, 128-row tiles only, h blocks of 128 behind blocks of
    256, ragged last blocks, column counts that do not divide by the 8 XCD shares -- must give the bits of the 256-row-tiles-only form (and so of gemm_tiled)."""
    g = torch.Generator().manual_seed(M * 3 + N + K)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g)).to(DEV)
    res = bf(torch.randn(M, N, generator=g)).to(DEV)
    Wp = E.op_pack_weight(W.to(DEV))
    nb = (M + 127) // 128
    try:
        for epi in ("none", "bias_gelu", "bias_res", "f32") + (("swiglu",) if N % 32 == 0 else ()):
            kw = dict(bias=bias if "bias" in epi else None, res=res if "res" in epi else None)
            _dense_mode(0)
            want = E.op_gemm(A, Wp, N, epi, **kw)
            for mix in [1, 0, 2] + [100 + h for h in sorted({1, 2, 3, nb - 1, nb}) if 1 <= h <= nb]:
                _dense_mix(mix)
                got = E.op_gemm(A, Wp, N, epi, **kw)
                torch.cuda.synchronize()
                assert torch.equal(got, want), f"mix {mix} {epi} M{M} N{N} K{K}: {int((got != want).sum())} elements differ from gemm_tiled"
    finally:
        _dense_mode(1)


@pytest.mark.parametrize("M,N,K,ks", [(1408, 512, 4096, 3), (700, 256, 1536, 5), (384, 1024, 2048, 7), (1408, 512, 1024, 2)])
def test_gemm_dense_uneven_k_slices_and_mixes(M, N, K, ks):
    """K slices of unequal length (K / 64 not a multiple of the slice count: the first (K / 64) % ks slices take one K-tile more) in the dense kernel's folded
    grid, for every tile mix, against gemm_tiled.hip cutting K at the same places: slabs summed in slice order by the reducing RMSNorm -> identical bits."""
    g = torch.Generator().manual_seed(M + N + K + ks)
    A = bf(torch.randn(M, K, generator=g)).to(DEV)
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    x = bf(torch.randn(M, N, generator=g)).to(DEV)
    nw = bf(1 + 0.2 * torch.randn(N, generator=g)).to(DEV)
    Wp = E.op_pack_weight(W.to(DEV))
    try:
        _dense_mode(0)
        x0, n0 = E.op_gemm_splitk_rmsnorm(A, Wp, x, ks, nw)
        for mix in (1, 0, 2, 101):
            _dense_mix(mix)
            x1, n1 = E.op_gemm_splitk_rmsnorm(A, Wp, x, ks, nw)
            assert torch.equal(x0, x1) and torch.equal(n0, n1), f"mix {mix}"
    finally:
        _dense_mode(1)
    want = bf(x.cpu().float() + bf(A.cpu().float() @ W.float().t()).float())
    close_bf16(x1.cpu(), want, f"dense split-K M{M} N{N} K{K} ks{ks}", ulps=2.5, atol=3.2e-2)


def test_in_launch_reduction_under_uneven_load():
    """The cross-workgroup hand-off of the in-launch reduction (sc1 slabs, drained, one agent-scope ticket per workgroup, the last arriver reads every
    slice with sc1 loads) checked the way MI355X_MICROARCH.md asks for hand-offs: under UNEVEN load, every word, many times.  A second stream keeps the
    chip busy with a bandwidth-bound copy and a compute-bound matmul of varying size while 300 fused launches run (o_proj and q/k/v forms at 64 and 22
    rows, rotating weights so that nothing is L2-warm by accident); every result must carry the bits of the quiet-chip run."""
    g = torch.Generator().manual_seed(77)
    cases = []
    for (M, N, K, ks, plain) in [(64, 4096, 4096, 4, False), (22, 4096, 4096, 2, False), (64, 6144, 4096, 2, True), (48, 4096, 14336, 4, False)]:
        A = bf(torch.randn(M, K, generator=g)).to(DEV)
        Ws = [E.op_pack_weight(bf(torch.randn(N, K, generator=g) * 0.05).to(DEV)) for _ in range(3)]
        x = bf(torch.randn(M, N, generator=g)).to(DEV)
        quiet = [E.op_gemm_splitk_plain(A, w, N, ks) if plain else E.op_gemm_splitk_fused(A, w, x, ks)[:2] for w in Ws]
        cases.append((A, Ws, x, N, ks, plain, quiet))
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    mm = torch.randn(2048, 2048, device=DEV)
    bad = 0
    for it in range(300):
        with torch.cuda.stream(side):  # uneven background load: alternating copy / matmul bursts of changing size
            if it % 3 == 0:
                n_el = (4 << 20) * (1 + it % 7)
                big[:n_el].copy_(big[n_el:2 * n_el])
            else:
                mm = (mm @ mm).clamp_(-1, 1)
        A, Ws, x, N, ks, plain, quiet = cases[it % len(cases)]
        w = it % 3
        if plain:
            got = E.op_gemm_splitk_plain(A, Ws[w], N, ks)
            bad += int(not torch.equal(got, quiet[w]))
        else:
            xn, ssq, _ = E.op_gemm_splitk_fused(A, Ws[w], x, ks)
            bad += int(not (torch.equal(xn, quiet[w][0]) and torch.equal(ssq, quiet[w][1])))
    torch.cuda.synchronize()
    assert bad == 0, f"{bad} of 300 fused launches under load differ from the quiet-chip bits"


@pytest.mark.parametrize("M,N,K,ks", [(96, 256, 1024, 2), (384, 1024, 2048, 4), (130, 512, 512, 1)])
def test_gemm_splitk_layernorm(M, N, K, ks):
    """Encoder out_proj / fc2 at 65..1024 rows: K slices on the dense kernel, summed (+ bias, + residual) by the LayerNorm kernel.
    Reference = fairseq TransformerSentenceEncoderLayer: x = residual + Linear(x) (bf16 each), then LayerNorm (fp32 statistics)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = bf(torch.randn(M, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    x = bf(torch.randn(M, N, generator=g))
    lw, lb = bf(1 + 0.2 * torch.randn(N, generator=g)), bf(0.1 * torch.randn(N, generator=g))
    x_new, normed = E.op_gemm_splitk_layernorm(A.to(DEV), E.op_pack_weight(W.to(DEV)), bias.to(DEV), x.to(DEV), ks, lw.to(DEV), lb.to(DEV), 1e-5)
    torch.cuda.synchronize()
    close_bf16(x_new, ref_linear(A, W, "bias_res", bias, x), f"splitk-ln x M{M}", ulps=2.5, atol=3.2e-2)
    ref_ln = torch.nn.functional.layer_norm(x_new.float().cpu(), (N,), lw.float(), lb.float(), 1e-5)
    close_bf16(normed, bf(ref_ln), f"splitk-ln norm M{M}", ulps=2.0, atol=4e-3)


@pytest.mark.parametrize("N,K,ks", [(1024, 1024, 2), (1024, 4096, 4), (768, 2048, 4), (1024, 4096, 8), (1024, 1024, 1)])
def test_layernorm_reduce_all_loads_first_carries_the_stepwise_kernels_bits(N, K, ks):
    """The LayerNorm that sums the K slices of the encoder's out_proj / fc2 asks for every load of a row before its first add (rowops.hip
    layernorm_reduce_lf_kernel, 512 < C <= 1024, <= 8 slices).  Both forms (isst_op_set_reduce_tuning) carry the same bits, x and normalised
    rows, from one stream's 48 rows to 300; held to the oracle."""
    g = torch.Generator().manual_seed(N + K + ks)
    A = bf(torch.randn(300, K, generator=g))
    W = bf(torch.randn(N, K, generator=g) * 0.05)
    bias = bf(torch.randn(N, generator=g))
    x = bf(torch.randn(300, N, generator=g))
    lw, lb = bf(1 + 0.2 * torch.randn(N, generator=g)), bf(0.1 * torch.randn(N, generator=g))
    Wp = E.op_pack_weight(W.to(DEV))
    lib = E.load_library()
    try:
        for M in (300, 256, 130, 48, 13):
            lib.isst_op_set_reduce_tuning(0, 0)
            x_st, n_st = E.op_gemm_splitk_layernorm(A[:M].to(DEV), Wp, bias.to(DEV), x[:M].to(DEV), ks, lw.to(DEV), lb.to(DEV), 1e-5)
            lib.isst_op_set_reduce_tuning(1 << 30, 1 << 30)
            x_lf, n_lf = E.op_gemm_splitk_layernorm(A[:M].to(DEV), Wp, bias.to(DEV), x[:M].to(DEV), ks, lw.to(DEV), lb.to(DEV), 1e-5)
            torch.cuda.synchronize()
            assert torch.equal(x_lf.view(torch.int16), x_st.view(torch.int16)), f"x bits M{M} N{N} K{K} S{ks}"
            assert torch.equal(n_lf.view(torch.int16), n_st.view(torch.int16)), f"norm bits M{M} N{N} K{K} S{ks}"
            close_bf16(x_lf, ref_linear(A[:M], W, "bias_res", bias, x[:M]), f"ln-reduce x M{M} N{N}", ulps=2.5, atol=3.2e-2)
            close_bf16(n_lf, bf(torch.nn.functional.layer_norm(x_lf.float().cpu(), (N,), lw.float(), lb.float(), 1e-5)), f"ln-reduce norm M{M} N{N}", ulps=2.0, atol=4e-3)
    finally:
        lib.isst_op_set_reduce_tuning(-1, -1)


def test_split_kv_merge_fused_into_oproj_equals_combine_then_gemm():
    """The flash-decoding merge of the decode attention's split-KV partials exists in three places that must agree bit for bit (one
    definition: csrc/common.h attn_merge_*): the combine launch, the o_proj GEMV's merge-on-load prologue (gemm.hip AMODE 3) and -- compared
    end to end in test_gpu_engine -- the attention kernel's own last-arriver combine.  Here: combine + GEMV against the fused GEMV, and the
    combine against a straightforward fp32 restatement (softmax re-weighting of the partial outputs)."""
    torch.manual_seed(0)
    H, S, K, N = 8, 19, 1024, 512
    part = torch.randn(2, H, S, 132, device="cuda")
    part[..., 128] *= 3.0                       # running max of each split
    part[..., 129] = part[..., 129].abs() + 0.5  # sum of each split
    part[0, 3, 5:, 128] = float("-inf")          # splits without a live key for a row (weight 0)
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    packed = E.op_pack_weight(w)
    res = torch.randn(2, N, device="cuda").bfloat16()
    lib = E.load_library()
    P = E._ptr
    for M in (1, 2):
        attn = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
        assert lib.isst_op_attn_combine(P(part), P(attn), H, M, S, E._stream_ptr()) == 0
        m, l, o = part[:M, :, :, 128].double(), part[:M, :, :, 129].double(), part[:M, :, :, :128].double()
        wgt = torch.exp(m - m.max(dim=-1, keepdim=True).values)
        ref = (o * wgt.unsqueeze(-1)).sum(2) / (l * wgt).sum(-1, keepdim=True)
        d = (attn.view(M, H, 128).double() - ref).abs().max().item()
        assert d <= 0.01 * ref.abs().max().item() + 1e-3, d
        sep = E.op_gemm(attn, packed, N, "res", res=res[:M].contiguous())
        fused = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        assert lib.isst_op_gemm_attn_merge(P(part), S, P(packed), P(res), N, P(fused), N, M, N, K, E._stream_ptr()) == 0
        torch.cuda.synchronize()
        assert torch.equal(sep, fused), f"M={M}: merge-on-load differs from combine + GEMV"


@pytest.mark.parametrize("ties", [True, False])
@pytest.mark.parametrize("rows,k", [(4, 8), (16, 8), (80, 16), (256, 8), (256, 16)])
def test_topk_rows_matches_a_sort_with_ties_to_the_lowest_index(rows, k, ties):
    """csrc/beam.hip: the beam search's candidate selection (reference: torch.topk over the processed scores, patch_hf.py:877-879).  Stage 1 works on
    (slice, row) workgroups: by threshold (round 5: a thread keeps its share of the slice in registers, the k-th largest of the 256 thread maxima bounds the
    slice's k-th largest entry from below, the few entries at or above it are collected and sorted) with the exact per-thread-list scan as the fallback when
    ties or bans make that collection long; stage 2 merges a row's slices in one wave.  Both must return exactly the k best (value, lowest index first on
    ties): checked on scores quantised so coarsely that ties are everywhere (`ties`: every workgroup takes the fallback) and on continuous scores (the
    threshold path), with -inf entries (suppressed tokens) mixed in and a row with fewer finite entries than k."""
    from infinisst_amd.engine import load_library, _ptr, _stream_ptr
    lib = load_library()
    V, ld = 128263, 128272
    g = torch.Generator(device="cuda").manual_seed(rows * 31 + k)
    sc = torch.randn(rows, ld, device="cuda", generator=g)
    if ties:
        sc = torch.round(sc * 3.0) / 3.0  # ~20 distinct values: the top k are all ties
    else:
        sc[0, 1000:1040] = sc[0].max() + 1.0  # ... and one run of equal leaders inside one slice
    sc[:, ::7] = float("-inf")
    sc[rows // 2, :V - 5] = float("-inf")  # a row with five finite entries
    out_val = torch.full((rows, 32), float("nan"), device="cuda")
    out_idx = torch.full((rows, 32), -7, device="cuda", dtype=torch.int32)
    rc = lib.isst_op_topk_rows(_ptr(sc), ld, V, k, rows, _ptr(out_val), _ptr(out_idx), _stream_ptr())
    assert rc == 0
    s = sc[:, :V].cpu().numpy()
    gv, gi = out_val.cpu().numpy(), out_idx.cpu().numpy()
    for r in range(rows):
        finite = int(np.isfinite(s[r]).sum())
        order = np.lexsort((np.arange(V), -s[r]))[:k]  # value descending, index ascending
        n = min(k, finite)  # (below that, -inf candidates: which of them come back is not specified)
        assert np.array_equal(gi[r, :n], order[:n]), f"row {r}: indices {gi[r, :k]} vs {order}"
        assert np.array_equal(gv[r, :n], s[r][order[:n]])
        assert np.all(np.isneginf(gv[r, n:k]))
