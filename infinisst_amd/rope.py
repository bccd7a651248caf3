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
