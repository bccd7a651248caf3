"""Host micro-timing used to size bench.py's cpu_baseline sample (bf16 eager matmuls on the GPU box's CPU)."""
import sys
import time

import torch

for nt in (16, 64, 128):
    torch.set_num_threads(nt)
    for M in (22, 1):
        x = torch.randn(M, 4096).bfloat16()
        w = torch.randn(14336, 4096).bfloat16()
        for dt in (torch.bfloat16, torch.float32):
            xx, ww = x.to(dt), w.to(dt)
            torch.nn.functional.linear(xx, ww)
            t = time.time()
            for _ in range(3):
                torch.nn.functional.linear(xx, ww)
            print(f"threads {nt:4d} M {M:3d} {str(dt):16s} {(time.time() - t) / 3 * 1e3:9.2f} ms", flush=True)
