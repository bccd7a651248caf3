"""Checkpoint loading for the hot path.

The reference loads `pytorch_model.bin` with `self.model.load_state_dict(torch.load(path))` after attaching the speech
encoder (agents/infinisst.py:176-180); the keys are the Lightning module's with the leading `model.` stripped
(train/prune_bin.py:5-11): `model.embed_tokens.*`, `model.layers.*`, `model.norm.*`, `lm_head.*`,
`model.speech_encoder.speech_encoder.*` (wav2vec2), `model.speech_encoder.length_shrink.*`, `model.speech_encoder.proj.*`.
Tensors that exist in real checkpoints but are not on the hot path are skipped: the convolutional positional embedding
(`...encoder.pos_conv.*`, never applied, patch_speech_encoder.py:488-498), wav2vec2 pre-training heads (quantizer,
project_q, final_proj, mask_emb, ...), and the rotary `freqs` parameters (consumed here to build the rotary table).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import synth
from .config import ModelConfig

ROTARY_SUFFIX = "self_attn.rotary_emb.freqs"


def split_state_dict(cfg: ModelConfig, state: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Optional[torch.Tensor], list]:
    """-> (hot-path tensors as bf16, encoder rotary inv_freq if the checkpoint carries one, skipped keys).
    Raises KeyError naming the first missing tensor and ValueError on a shape mismatch (strict, like the reference)."""
    want = synth.weight_shapes(cfg)
    out: Dict[str, torch.Tensor] = {}
    inv_freq = None
    skipped = []
    for k, v in state.items():
        if k in want:
            if tuple(v.shape) != tuple(want[k]):
                raise ValueError(f"{k}: checkpoint shape {tuple(v.shape)} != expected {tuple(want[k])}")
            out[k] = v.to(torch.bfloat16).contiguous()
        elif k.endswith(ROTARY_SUFFIX):
            if inv_freq is None:
                inv_freq = v.detach().float().clone()
            elif not torch.equal(inv_freq, v.detach().float()):
                raise ValueError("encoder layers carry different rotary freqs; one shared table is assumed")
            skipped.append(k)
        else:
            skipped.append(k)
    missing = [k for k in want if k not in out]
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} hot-path tensors, first: {missing[0]}")
    return out, inv_freq, skipped


def load_checkpoint(cfg: ModelConfig, path: str):
    """torch.load(path, map_location='cpu', weights_only=True) -> split_state_dict."""
    state = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(state, dict) and "state_dict" in state and not any(k.startswith("model.") for k in state):
        state = {k[len("model."):] if k.startswith("model.model.") or k.startswith("model.lm_head") else k: v
                 for k, v in state["state_dict"].items()}  # un-pruned Lightning checkpoint (train/prune_bin.py does this)
    return split_state_dict(cfg, state)
