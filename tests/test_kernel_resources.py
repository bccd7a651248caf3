"""Register / scratch budget of the hand-scheduled kernels, read from the code objects inside the built library (no GPU needed).

gemm_wide.hip's inline-asm MFMA chains, the decode attention and the fused attention + o_proj launch are written against a fixed register budget: a build
that spills gives gemm_wide WRONG results (its waits are counted by hand, and scratch traffic counts in vmcnt) and costs the attention kernels their occupancy.
A toolchain or source change that pushes one of them into scratch must fail here, not show up as a slow or wrong kernel on the GPU box."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "..", "infinisst_amd", "libinfinisst_hip.so")

NO_SCRATCH = ("gemm_wide_kernel", "llm_attn_partial_kernel", "llm_attn_oproj_kernel", "gemm_skinny_kernel", "gemm_dense_kernel", "gemm_tiled_kernel",
              "enc_attention_kernel", "llm_attn_combine_kernel", "sample_fused_kernel")


def _kernels(tmp_path):
    lib = tmp_path / "lib.so"
    shutil.copy(LIB, lib)
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    out = {}
    for f in sorted(tmp_path.iterdir()):
        if not f.name.endswith("gfx950"):
            continue
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        for block in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            scratch = re.search(r"\.private_segment_fixed_size:\s+(\d+)", block)
            spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", block)
            vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
            if name and scratch and spill and vgpr:
                out[name.group(1)] = (int(scratch.group(1)), int(spill.group(1)), int(vgpr.group(1)))
    return out


@pytest.mark.skipif(not os.path.exists(f"{LLVM}/llvm-readelf"), reason="ROCm LLVM tools not installed")
def test_hand_scheduled_kernels_use_no_scratch(tmp_path, built_library):
    ks = _kernels(tmp_path)
    assert len(ks) > 100, f"only {len(ks)} kernels found in the library's gfx950 code objects"
    seen = {p: 0 for p in NO_SCRATCH}
    for name, (scratch, spill, vgpr) in ks.items():
        for p in NO_SCRATCH:
            if p in name:
                seen[p] += 1
                assert scratch == 0 and spill == 0, f"{name}: {scratch} bytes of scratch per lane, {spill} spilled VGPRs ({vgpr} VGPRs)"
    assert all(n > 0 for n in seen.values()), f"kernel families missing from the library: {[p for p, n in seen.items() if n == 0]}"
    wide = {n: v for n, v in ks.items() if "gemm_wide_kernel" in n}
    assert max(v[2] for v in wide.values()) <= 256  # two 8-wave workgroups per CU (launch_bounds(512, 2))
    fused = {n: v for n, v in ks.items() if "llm_attn_oproj_kernel" in n}
    assert max(v[2] for v in fused.values()) <= 256  # one 8-wave workgroup per CU
    # the many-stream encoder attention (48-row query blocks, QT = 3) runs two 8-wave workgroups per CU: 4 waves per SIMD = 128 registers (enc_attn.hip's launch bounds)
    enc3 = {n: v for n, v in ks.items() if "enc_attention_kernelILi3" in n}
    assert enc3 and max(v[2] for v in enc3.values()) <= 128
    # the prefill attention is written for two 8-wave workgroups per CU (128 registers); it is KNOWN to spill a few registers around its rotation at that budget
    # (24-28 since round 4) -- a jump means its loop changed shape
    pf = {n: v for n, v in ks.items() if "llm_attn_prefill_kernel" in n}
    assert pf and max(v[2] for v in pf.values()) <= 128 and max(v[1] for v in pf.values()) <= 32, {n: v for n, v in pf.items()}
