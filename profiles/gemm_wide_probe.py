"""Probe (round 4): the decode GEMMs at 65..256 rows -- the old dispatch (gemm_mid as two 64-row blocks / gemm_tiled / gemm_dense; isst_op_set_gemm_tuning(900000 + 0, 0)
switches gemm_wide off) against gemm_wide.hip's ring-depth variants (build with `make EXTRA=-DISST_WIDE_PROBE`), cold rotating weights (> 700 MB of copies per shape),
outputs compared bit for bit with the old path.  K-sliced shapes run GEMM (fp32 slabs) + the reducing residual / RMSNorm launch, as the engine does."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from infinisst_amd import engine as E
dev = "cuda"; lib = E.load_library(os.environ.get("LIB")); E._lib = lib; P = E._ptr  # LIB=path: a variant build of the library
ROWS = [int(x) for x in os.environ.get("ROWS", "128,256,96,192").split(",")]
VARIANTS = [int(x) for x in os.environ.get("VARIANTS", "0,1,2,3").split(",")]
SH = {"qkv": (6144, 4096, "none", (1, 2, 4, 8)), "o_proj": (4096, 4096, "res", (4, 8)), "gate_up": (28672, 4096, "swiglu", (1,)), "down": (4096, 14336, "res", (4, 8)),
      "lm_head": (128272, 4096, "f32", (1,))}
ONLY = os.environ.get("SHAPES")
def timeit(run, n=30):
    for i in range(3): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): run(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in ROWS:
    for name, (N, K, epi, splits) in SH.items():
        if ONLY and name not in ONLY.split(","): continue
        Np = (N + 15) // 16 * 16
        copies = max(3, (700 << 20) // (Np * K * 2) + 1)
        packs = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
        PAD = int(os.environ.get("LDA_PAD", "0"))  # elements added to the row stride of A (rows of 8 KB all start in the same L2 channel)
        A = torch.randn(M, K + PAD, device=dev).bfloat16()
        LDA = K + PAD
        for ks in splits:
            line = f"M={M:3d} {name:8s} ks={ks}  W={Np * K * 2 / 1e6:7.1f} MB ({Np * K * 2 / 6.6e6:6.1f} us at 6.6 TB/s):"
            ref = None
            DBG = [int(x) for x in os.environ.get("DBG", "0").split(",")]  # 1: W descriptor emptied, 2: A descriptor emptied, 3: both (timing only)
            LOW = 100 if os.environ.get("LOWROWS") else 0  # + 100: gemm_wide also takes 17..64 rows (MT = 4 form)
            for mode, var in [(0, 0)] + [(LOW + 2 + 10 * d, v) for v in VARIANTS for d in DBG]:
                lib.isst_op_set_gemm_tuning(900000 + mode * 10 + var, 0)
                if ks == 1:
                    n_out = N // 2 if epi == "swiglu" else N
                    out = torch.zeros(M, n_out, device=dev, dtype=torch.float32 if epi == "f32" else torch.bfloat16)
                    def run(i):
                        rc = lib.isst_op_gemm(P(A), LDA, P(packs[i % copies]), None, None, 0, P(out), out.stride(0), M, Np, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr())
                        assert rc == 0, rc
                    us = timeit(run)
                    run(0); torch.cuda.synchronize(); got = out.clone()
                else:
                    x0 = torch.randn(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).bfloat16()
                    x = x0.clone(); nw = torch.ones(N, device=dev).bfloat16(); xo = torch.empty_like(x)
                    slabs = torch.empty(ks, M, N, device=dev, dtype=torch.float32)
                    def run(i):
                        rc = lib.isst_op_gemm_splitk_rmsnorm(P(A), LDA, P(packs[i % copies]), P(x), P(nw), P(xo), P(slabs), M, N, K, ks, 1e-5, E._stream_ptr())
                        assert rc == 0, rc
                    us = timeit(run)
                    x.copy_(x0); run(0); torch.cuda.synchronize(); got = x.clone()
                if ref is None: ref = got; tag = "old"
                elif (mode % 100) >= 10: tag = f"v{var}/dbg{(mode % 100) // 10}"
                else: tag = f"v{var}" + ("" if torch.equal(got, ref) else f"(DIFF {float((got.float() - ref.float()).abs().max()):.3g})")
                line += f"  {tag} {us:6.1f}"
            print(line, flush=True)
        del packs
lib.isst_op_set_gemm_tuning(900000 + 10, 0)
