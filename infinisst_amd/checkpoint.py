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

import dataclasses
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


def normalise_keys(state) -> Dict[str, torch.Tensor]:
    """Bring a checkpoint to the pruned layout the reference agent loads (train/prune_bin.py:5-11 strips ONE leading `model.`
    of the Lightning module): accepted inputs are the pruned `pytorch_model.bin`, an un-pruned Lightning checkpoint
    (`{"state_dict": {...}}`) and the flat un-pruned dict prune_bin.py itself consumes (every key prefixed `model.`)."""
    if isinstance(state, dict) and "state_dict" in state and isinstance(state["state_dict"], dict):
        state = state["state_dict"]
    keys = list(state.keys())
    unpruned = bool(keys) and all(k.startswith("model.") for k in keys) and any(
        k.startswith("model.lm_head.") or k.startswith("model.model.") for k in keys)
    if unpruned:
        state = {k[len("model."):]: v for k, v in state.items()}
    return state


def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """torch.load(path, map_location='cpu', weights_only=True) (reference agents/infinisst.py:179) -> pruned key layout."""
    return normalise_keys(torch.load(path, map_location="cpu", weights_only=True))


def load_checkpoint(cfg: ModelConfig, path: str):
    """`pytorch_model.bin` -> split_state_dict."""
    return split_state_dict(cfg, load_state_dict_file(path))


def infer_config(state: Dict[str, torch.Tensor], base: Optional[ModelConfig] = None, **overrides) -> ModelConfig:
    """Model geometry from the tensor SHAPES of a checkpoint in the reference key layout.

    The reference gets the Llama dimensions from `from_pretrained(args.model_name)`'s config.json and the wav2vec2 ones from the
    fairseq checkpoint at `--w2v2-path` (agents/infinisst.py:150-171); neither file is needed here because the state dict the
    agent loads anyway (`--state-dict-path`) carries every dimension: conv layers (dim, kernel) -- strides are wav2vec2's
    (kernel 10 -> 5, otherwise 2) --, encoder width / depth / FFN, shrink convs, Llama width / depth / heads / FFN / vocab.
    Head dims are fixed by the kernels (64 encoder, 128 Llama).  `overrides` (block_size, max_cache_size, ids, ...) win."""
    from .synth import ENC, SHR
    base = base or ModelConfig()

    def count(fmt: str) -> int:
        n = 0
        while fmt.format(n) in state:
            n += 1
        return n

    def need(key: str) -> torch.Tensor:
        if key not in state:
            raise KeyError(f"checkpoint lacks {key}")
        return state[key]

    n_conv = count(ENC + "feature_extractor.conv_layers.{}.0.weight")
    if n_conv == 0:
        raise KeyError(f"checkpoint lacks {ENC}feature_extractor.conv_layers.0.0.weight")
    conv = []
    for i in range(n_conv):
        c, _, k = state[f"{ENC}feature_extractor.conv_layers.{i}.0.weight"].shape
        conv.append((int(c), int(k), 5 if k == 10 else 2))
    enc_dim = int(need(ENC + "post_extract_proj.weight").shape[0])
    enc_layers = count(ENC + "encoder.layers.{}.fc1.weight")
    enc_ffn = int(need(ENC + "encoder.layers.0.fc1.weight").shape[0])
    n_shr = count(SHR + "conv_layers.{}.0.weight")
    shrink = []
    for i in range(n_shr):
        c, _, k = state[f"{SHR}conv_layers.{i}.0.weight"].shape
        shrink.append((int(c), int(k), int(k)))
    vocab, llm_dim = (int(x) for x in need("model.embed_tokens.weight").shape)
    llm_layers = count("model.layers.{}.mlp.down_proj.weight")
    hd = base.llm_head_dim
    q_rows = int(need("model.layers.0.self_attn.q_proj.weight").shape[0])
    k_rows = int(need("model.layers.0.self_attn.k_proj.weight").shape[0])
    if enc_dim % 64 or q_rows % hd or k_rows % hd:
        raise ValueError("head dims other than 64 (encoder) / 128 (Llama) are not supported by the kernels")
    cfg = dataclasses.replace(
        base, conv_layers=conv, conv_bias=(ENC + "feature_extractor.conv_layers.0.0.bias") in state, enc_dim=enc_dim,
        enc_layers=enc_layers, enc_heads=enc_dim // 64, enc_ffn=enc_ffn, shrink_layers=shrink, llm_dim=llm_dim, llm_layers=llm_layers,
        llm_heads=q_rows // hd, llm_kv_heads=k_rows // hd, llm_ffn=int(need("model.layers.0.mlp.gate_proj.weight").shape[0]), vocab=vocab)
    return dataclasses.replace(cfg, **overrides)
