"""gemm_dense.hip, round 6: the tile mix (row blocks of 256 and of 128 rows in one launch) and the K-slice count on the many-row shapes of a 64-stream prefill
and of the encoder at 64 streams; random operands, weights rotated over copies; every form compared bit for bit with the 256-row-tiles-only form of rounds 1-5.

    python profiles/dense_mix_probe.py [rows]
mix codes (isst_op_set_gemm_tuning(800000 + 2 + 10 * mix)): 0 = the launcher's model, 1 = 256-row tiles only, 2 = 128-row tiles only, 100 + h = h row blocks of 128.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from infinisst_amd import engine as E

lib = E.load_library()
dev = "cuda"
P = E._ptr


def timeit(fn, n=20):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


def mix(m):
    lib.isst_op_set_gemm_tuning(800000 + 2 + 10 * m, 0)


M = int(sys.argv[1]) if len(sys.argv) > 1 else 1408
copies = 3
nb = (M + 127) // 128
print(f"rows {M} = {nb} blocks of 128")
for name, N, K, epi in [("gate/up", 28672, 4096, "swiglu"), ("q/k/v", 6144, 4096, "none"), ("o_proj", 4096, 4096, "res"), ("down unsplit", 4096, 14336, "res")]:
    Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    A = torch.randn(M, K, device=dev).bfloat16()
    res = torch.randn(M, N, device=dev).bfloat16() if epi == "res" else None
    n_out = N // 2 if epi == "swiglu" else N
    ref = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)

    def run(i, o):
        rc = lib.isst_op_gemm(P(A), K, P(Wps[i % copies]), None, P(res), N if res is not None else 0, P(o), n_out, M, N, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr())
        assert rc == 0, rc

    mix(1)
    run(0, ref)
    t1 = timeit(lambda i: run(i, ref))
    run(0, ref)
    line = f"{name:13s} N={N:6d} K={K:6d}: 256-row tiles {t1:7.1f} us |"
    for m in [0, 2] + [100 + h for h in (1, 3, 5, 7, 9) if h <= nb]:
        mix(m)
        out.zero_()
        run(0, out)
        t = timeit(lambda i: run(i, out))
        run(0, out)
        torch.cuda.synchronize()
        line += f" mix {m}: {t:7.1f}{'' if torch.equal(out, ref) else ' DIFFERS(' + str(int((out != ref).sum())) + ')'} |"
    print(line, flush=True)
    del Wps

# split-K into slabs + the reducing RMSNorm (o_proj, down_proj as the engine runs them)
for name, N, K in [("o_proj", 4096, 4096), ("down", 4096, 14336)]:
    Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(copies)]
    A = torch.randn(M, K, device=dev).bfloat16()
    x0 = torch.randn(M, N, device=dev).bfloat16()
    nw = (1 + 0.1 * torch.randn(N, device=dev)).bfloat16()
    slabs = torch.empty(8 * M * N, device=dev, dtype=torch.float32)
    for ks in (1, 2, 3, 4, 5, 6, 7, 8):
        if (K // 64) // ks < 2:
            continue
        outs = {}
        line = f"{name:7s} split-K {ks} + reducing norm:"
        for m in (1, 0, 2):
            mix(m)
            x = x0.clone()
            o = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)

            def run(i):
                rc = lib.isst_op_gemm_splitk_rmsnorm(P(A), K, P(Wps[i % copies]), P(x), P(nw), P(o), P(slabs), M, N, K, ks, 1e-5, E._stream_ptr())
                assert rc == 0, rc
            run(0)
            torch.cuda.synchronize()
            outs[m] = (x.clone(), o.clone())
            t = timeit(run)
            same = torch.equal(outs[m][0], outs[1][0]) and torch.equal(outs[m][1], outs[1][1])
            line += f" mix {m}: {t:7.1f} us{'' if same else ' DIFFERS'} |"
        print(line, flush=True)
    del Wps
mix(0)
lib.isst_op_set_gemm_tuning(800001, 0)
