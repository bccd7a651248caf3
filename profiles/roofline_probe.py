"""Runs only bench.py's roofline probe (the gate/up GEMV on cold weights); used under rocprofv3 --pmc to read the
HBM traffic of the dominant kernel:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 profiles/roofline_probe.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from infinisst_amd.config import full_config
torch.cuda.set_device(0)
print(json.dumps(bench.gemm_roofline(full_config(), torch.device("cuda", 0), iters=40)))
