"""What each of the three streams of gemm_dense.hip's K loop costs alone: the product library beside the diagnostic builds of `make -C infinisst_amd/csrc ablate`
(DENSE_ABLATE 1 = no MFMAs, 2 = no DMAs in the loop, 4 = no fragment reads in the loop, 6 = MFMAs + barriers only), one process per library, on the gate/up and
o_proj shapes of a 64-stream prefill with 256-row tiles only (mix 1) and 128-row tiles only (mix 2).  Warm: 40 launches first, then the median of 5 x 10.

    python profiles/dense_ablate_probe.py            # all libraries (child processes)
    python profiles/dense_ablate_probe.py <lib.so>   # one library
"""
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) < 2:
    for tag in ("", "_ring0", "_ablate1", "_ablate2", "_ablate4", "_ablate6"):
        lib = os.path.join(ROOT, "infinisst_amd", f"libinfinisst_hip{tag}.so")
        if os.path.exists(lib):
            subprocess.run([sys.executable, os.path.abspath(__file__), lib], check=False)
    sys.exit(0)

import torch

from infinisst_amd import engine as E

lib = E.load_library(sys.argv[1])
P = E._ptr
dev = "cuda"
M = 1408
names = {"": "product", "_ring0": "two K-tile buffers (round 5)", "_ablate1": "no MFMA", "_ablate2": "no DMA in the loop", "_ablate4": "no fragment reads in the loop", "_ablate6": "MFMA + barriers only"}
tag = os.path.basename(sys.argv[1]).replace("libinfinisst_hip", "").replace(".so", "")
for name, N, K, epi, M in [("gate/up", 28672, 4096, "swiglu", 1408), ("o_proj", 4096, 4096, "none", 1408), ("enc qkv", 3072, 1024, "bias", 3072), ("enc fc1", 4096, 1024, "bias_gelu", 3072)]:
    Wps = [E.op_pack_weight((torch.randn(N, K, device=dev) * 0.02).bfloat16()) for _ in range(3)]
    A = torch.randn(M, K, device=dev).bfloat16()
    n_out = N // 2 if epi == "swiglu" else N
    out = torch.zeros(M, n_out, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(N, device=dev).bfloat16() if "bias" in epi else None

    def run(i):
        rc = lib.isst_op_gemm(P(A), K, P(Wps[i % 3]), P(bias), None, 0, P(out), n_out, M, N, K, n_out, E.EPI[epi], None, 0.0, E._stream_ptr())
        assert rc == 0, rc
    res = []
    for mix in (1, 2):
        lib.isst_op_set_gemm_tuning(800000 + 2 + 10 * mix, 0)
        for i in range(40):
            run(i)
        torch.cuda.synchronize()
        ts = []
        for r in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(10):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        res.append(statistics.median(ts))
    print(f"{names.get(tag, tag):30s} {name:8s} 256-row tiles {res[0]:7.1f} us | 128-row tiles {res[1]:7.1f} us", flush=True)
    del Wps
