"""The reference-generated fixtures fed to the HIP kernels through their own C-ABI entry points (SURVEY 8(b)'s per-kernel list):
  splice.npz         <- SpeechLlamaModel.forward            (model/llm.py:86-113)            -> isst_op_splice_map + isst_op_embed_splice
  llm_attention.npz  <- llama_sdpa_attention_new_forward    (model/patches/patch_llm.py:231-336) -> isst_op_llm_attention
  encoder.npz        <- uni_mha_forward / uni_w2v2_forward  (patch_speech_encoder.py:228-933)    -> isst_op_enc_attention (layer 0)
The projections around the attention cores run on the CPU with the fixture's (seeded) weights, so that what is compared is exactly
the kernel under test: rotation, cache append, ring addressing, masking, softmax, P.V."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from infinisst_amd import engine as E
from infinisst_amd import rope, synth
from infinisst_amd.config import toy_config
from oracle import speech_encoder as oenc

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_splice_fixture_through_the_hip_kernel(golden_dir):
    g = np.load(os.path.join(golden_dir, "splice.npz"))
    user, assist, sh, _ = (int(x) for x in g["ids_cfg"])
    for case in range(3):
        ids = g[f"ids_{case}"]
        feats = torch.from_numpy(g[f"feats_{case}"]).bfloat16()
        table = torch.from_numpy(g[f"table_{case}"]).bfloat16()
        m = E.op_splice_map(ids, user, assist, sh, feats.shape[0])
        tok = torch.tensor([ids[t] if t >= 0 else 0 for t in m], dtype=torch.int32, device=DEV)
        srow = torch.tensor([-1 if t >= 0 else -1 - t for t in m], dtype=torch.int32, device=DEV)
        out = E.op_embed_splice(tok, srow, table.to(DEV), feats.to(DEV)).cpu()
        ref = torch.from_numpy(g[f"embeds_{case}"]).bfloat16()  # a pure row copy: the bf16 cast commutes with it
        assert torch.equal(out, ref), f"case {case}"


@pytest.mark.parametrize("rot_keys", [True, False])
@pytest.mark.parametrize("ring_start", [0, 100])
def test_llm_attention_fixture_through_the_hip_kernels(golden_dir, rot_keys, ring_start):
    """7 calls of the reference's patched attention on one layer: prefill (30), decode, decode, chunked prefill over a cache (22),
    decode, then eviction (keep first 5 + last 20: every surviving key re-indexes) followed by decode and another 22-row prefill.
    ring_start 100 with a 128-slot ring makes the live span wrap around the end of the ring."""
    g = np.load(os.path.join(golden_dir, "llm_attention.npz"))
    cfg = toy_config()
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.05, seed=4321)
    p = "model.layers.0.self_attn."
    H, KV = cfg.llm_heads, cfg.llm_kv_heads
    sys_cap, ring_cap, sys_len = 64, 128, 5
    slots = sys_cap + ring_cap
    kpool = torch.zeros((KV, slots, 128), dtype=torch.bfloat16, device=DEV)
    krpool, vpool = torch.zeros_like(kpool), torch.zeros_like(kpool)
    cos = torch.from_numpy(g["rope_cos"])[:, :64].bfloat16().contiguous()
    sin = torch.from_numpy(g["rope_sin"])[:, :64].bfloat16().contiguous()
    pad = torch.zeros((256 - cos.shape[0], 64), dtype=torch.bfloat16)
    cos_d, sin_d = torch.cat([cos, pad]).to(DEV), torch.cat([sin, pad]).to(DEV)
    total, start, worst = 0, ring_start, 0.0
    for j in range(int(g["n_calls"])):
        if j == int(g["evict_after"]):  # agents/infinisst.py:354-361 as a ring-start advance: keep the pinned 5 + the last 20
            drop = (total - sys_len) - 20
            start = (start + drop) % ring_cap
            total = sys_len + 20
        x = torch.from_numpy(g[f"bf16_x_{j}"])[0].bfloat16()
        qkv = torch.cat([F.linear(x, w[p + "q_proj.weight"]), F.linear(x, w[p + "k_proj.weight"]), F.linear(x, w[p + "v_proj.weight"])], dim=1)
        out = E.op_llm_attention(qkv.to(DEV).contiguous(), total, kpool, krpool, vpool, H, KV, sys_cap, ring_cap, sys_len, start, cos_d, sin_d, rot_keys)
        y = F.linear(out.cpu(), w[p + "o_proj.weight"]).float()
        ref = torch.from_numpy(g[f"bf16_y_{j}"])[0]
        d = float((y - ref).abs().max())
        worst = max(worst, d)
        print(f"call {j}: rows {x.shape[0]} cached {total} max |d| {d:.4f} (|ref| max {float(ref.abs().max()):.3f})")
        assert d <= 0.03 + 0.02 * float(ref.abs().max()), f"call {j}"
        total += x.shape[0]
    # the arena holds the reference's UNROTATED key cache, in logical order behind the ring mapping
    ref_k = torch.from_numpy(g["bf16_kcache"])[0]  # (KV, 48, 128)
    logical = [s if s < sys_len else sys_cap + (start + s - sys_len) % ring_cap for s in range(total)]
    got_k = kpool.cpu()[:, logical].float()
    assert got_k.shape == ref_k.shape
    assert float((got_k - ref_k).abs().max()) <= 0.016, "key cache differs from the reference's unrotated cache"  # CPU bf16 GEMM of the two runs: <= 1 ulp


def test_encoder_attention_fixture_layer0_through_the_hip_kernel(golden_dir):
    """encoder.npz (bf16 leg): 6 chunks of 16 frames, window 40 -- first chunk (training mask), growing window, saturated window with
    the ring trimmed by a start advance.  The front end up to layer 0's q/k/v projections runs on the CPU (oracle, pinned to the
    same fixture); the HIP kernel gets the qkv rows and its own K/V rings.  Checked: the attention output (through out_proj) against
    the oracle's uni_mha_forward restatement, and the ring's key content against the REFERENCE's layer-0 cache (`bf16_k0_<c>`)."""
    g = np.load(os.path.join(golden_dir, "encoder.npz"))
    cfg = toy_config().replace(block_size=int(g["block_size"]), max_cache_size=int(g["max_cache_size"]), enc_rope_mode="fp32")
    w = synth.random_weights(cfg, dtype=torch.bfloat16, std=0.08, norm_jitter=0.1, seed=1234)
    Hh, D, Q, C = cfg.enc_heads, cfg.enc_dim, cfg.block_size, cfg.max_cache_size
    cap = 64
    kring = torch.zeros((Hh, cap, 64), dtype=torch.bfloat16, device=DEV)
    vring = torch.zeros((Hh, 64, cap), dtype=torch.bfloat16, device=DEV)
    ec, es = rope.encoder_tables(cfg, cap)
    ec, es = ec.to(DEV), es.to(DEV)
    rope_o = oenc.make_rope(cfg)
    cache = oenc.new_cache(cfg)           # full oracle stream (front end + all layers), only its intermediates are used
    lc0 = oenc.LayerCache()               # layer 0's cache for the reference leg of the attention core
    audio = g["audio"]
    p = oenc.ENC + "encoder.layers.0."
    start, length, steps = 0, 0, 0
    for c in range(6):
        seg = torch.from_numpy(audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples])
        if c == 0:
            seg = torch.cat([torch.zeros(cfg.first_chunk_offset), seg])
        _, inter = oenc.w2v2_forward(w, cfg, seg.unsqueeze(0).bfloat16(), cache, cfg.block_size, rope_o, return_intermediates=True)
        x = inter["post_proj"][0]  # (Q, D) input of layer 0
        xn = F.layer_norm(x, (D,), w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], cfg.enc_ln_eps)
        qkv = torch.cat([F.linear(xn, w[p + f"self_attn.{n}.weight"], w[p + f"self_attn.{n}.bias"]) for n in ("q_proj", "k_proj", "v_proj")], dim=1)
        # reference leg: the oracle's uni_mha_forward on the same normalised rows and its own trimmed cache (patch_speech_encoder.py:516-520)
        if lc0.k is not None:
            lc0.k, lc0.v = lc0.k[:, -C:], lc0.v[:, -C:]
        mask = oenc.attn_mask_inference(Q, steps, C, cfg.block_size) if steps > 0 else oenc.attn_mask_training(Q, C, cfg.block_size)
        ref = oenc.mha_forward(w, cfg, p + "self_attn.", xn.unsqueeze(1), mask, lc0, rope_o)[:, 0].float()
        # HIP leg: trim = ring-start advance, then the kernel appends the chunk's keys itself
        if length > C:
            start = (start + length - C) % cap
            length = C
        out = E.op_enc_attention(qkv.to(DEV).contiguous(), kring, vring, start, steps, ec, es, False, Hh, C, cfg.block_size)
        y = F.linear(out.cpu(), w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"]).float()
        length += Q
        steps += Q
        d = float((y - ref).abs().max())
        print(f"chunk {c}: window {length} max |d| {d:.4f} (|ref| max {float(ref.abs().max()):.3f})")
        assert d <= 0.03 + 0.02 * float(ref.abs().max()), f"chunk {c}"
        ref_k = torch.from_numpy(g[f"bf16_k0_{c}"])  # (heads, length, 64): the reference's layer-0 key cache after this chunk
        slots_ = [(start + j) % cap for j in range(length)]
        got_k = kring.cpu()[:, slots_].float()
        assert got_k.shape == ref_k.shape
        assert float((got_k - ref_k).abs().max()) <= 0.016 + 0.008 * float(ref_k.abs().max()), f"chunk {c}: ring keys differ from the reference cache"
        got_v = vring.cpu()[:, :, slots_].transpose(1, 2).float()
        assert float((got_v - lc0.v.float()).abs().max()) <= 0.016 + 0.008 * float(lc0.v.float().abs().max()), f"chunk {c}: ring values"
