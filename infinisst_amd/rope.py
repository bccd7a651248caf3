"""Host-side rotary tables handed to the library (isst_set_rope_tables).

The kernels only look cos/sin up by position, so whichever third-party semantics has to be reproduced is
decided here, on the host:
  * encoder: rotary_embedding_torch.RotaryEmbedding(dim=64, use_xpos=False) as used at reference
    model/patches/patch_speech_encoder.py:631,:824 -- theta 10000, interleaved pairs.  After the reference casts
    the speech encoder to bf16 (agents/infinisst.py:173) the module's `freqs`, the `arange` positions, their
    product and cos/sin are all bf16 ("bf16" mode); "fp32" keeps everything fp32.
  * LLM: HF LlamaRotaryEmbedding(rope_type="llama3") [transformers 4.47.0] -- fp32 angles, cos/sin cast to bf16,
    emb = cat(freqs, freqs) so only the first 64 of 128 dims are stored.
"""
from __future__ import annotations

import math

import torch

from .config import ModelConfig


def encoder_tables(cfg: ModelConfig, rows: int, inv_freq=None):
    """fp32 cos, sin of shape (rows, head_dim // 2).  `inv_freq`: the `rotary_emb.freqs` parameter of a checkpoint
    (strict load_state_dict overwrites the module's initial value with it), default: the module's initial value."""
    hd = cfg.enc_head_dim
    if not cfg.enc_rope:  # --rope 0 (patch_speech_encoder.py:823): the kernel's rotation becomes the identity, exactly
        return torch.ones(rows, hd // 2), torch.zeros(rows, hd // 2)
    inv = 1.0 / (cfg.enc_rope_theta ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd))
    if inv_freq is not None:
        if tuple(inv_freq.shape) != tuple(inv.shape):
            raise ValueError(f"rotary freqs shape {tuple(inv_freq.shape)} != {tuple(inv.shape)}")
        inv = inv_freq.float()
    pos = torch.arange(rows, dtype=torch.float32)
    if cfg.enc_rope_mode == "bf16":
        ang = pos.bfloat16().unsqueeze(1) * inv.bfloat16().unsqueeze(0)
        return ang.cos().float().contiguous(), ang.sin().float().contiguous()
    if cfg.enc_rope_mode != "fp32":
        raise ValueError(f"enc_rope_mode {cfg.enc_rope_mode!r}")
    ang = pos.unsqueeze(1) * inv.unsqueeze(0)
    return ang.cos().contiguous(), ang.sin().contiguous()


ENC_POS_ROWS = 2305  # = ISST_ENC_POS_ROWS (include/infinisst_hip.h)


def encoder_position_values(rows: int = ENC_POS_ROWS) -> torch.Tensor:
    """The positions a bf16 `arange` can hold, in table order: 0..255, then the bf16 bit patterns from 0x4380 (= 256.0) on."""
    small = torch.arange(min(rows, 256), dtype=torch.float32)
    if rows <= 256:
        return small
    bits = (torch.arange(rows - 256, dtype=torch.int32) + 0x4380) << 16
    return torch.cat([small, bits.view(torch.float32)])


def encoder_position_table(cfg: ModelConfig, rows: int = ENC_POS_ROWS) -> torch.Tensor:
    """--rope 0: bf16 (rows, enc_dim) = [sin(p f_j) | cos(p f_j)] for every position p of `encoder_position_values`, with the arithmetic of the
    reference's sinusoidal_positional_embedding (model/patches/patch_speech_encoder.py:448-461): index, frequency, position, product and
    sin / cos all bf16.  handed to isst_set_enc_position_table; the kernel looks a frame's row up by the bf16 rounding of its position."""
    d = cfg.enc_dim
    if d % 2:
        raise ValueError("enc_dim must be even")
    half = d // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.bfloat16) * -step)
    pos = encoder_position_values(rows).bfloat16()  # exact: every value is a bf16 integer
    ang = pos.unsqueeze(1) * freq.unsqueeze(0)
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).contiguous()


def llama3_inv_freq(cfg: ModelConfig) -> torch.Tensor:
    dim = cfg.llm_head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    low_wl = cfg.rope_original_max_pos / cfg.rope_low_freq_factor
    high_wl = cfg.rope_original_max_pos / cfg.rope_high_freq_factor
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > low_wl, inv / cfg.rope_factor, inv)
    smooth = (cfg.rope_original_max_pos / wavelen - cfg.rope_low_freq_factor) / (
        cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
    mid = (1 - smooth) * scaled / cfg.rope_factor + smooth * scaled
    is_mid = ~(wavelen < high_wl) & ~(wavelen > low_wl)
    return torch.where(is_mid, mid, scaled)


def llm_tables(cfg: ModelConfig, rows: int):
    """bf16 cos, sin of shape (rows, head_dim // 2)."""
    freqs = torch.arange(rows, dtype=torch.float32).unsqueeze(1) * llama3_inv_freq(cfg).unsqueeze(0)
    return freqs.cos().bfloat16().contiguous(), freqs.sin().bfloat16().contiguous()
