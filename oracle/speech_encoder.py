"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference speech encoder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Plain PyTorch on CPU over an explicit weight dict (reference checkpoint key names) and an explicit
cache object; runs in fp32 or bf16 (`dtype` of the weights decides) with the reference's op order, so
every torch op rounds where the reference's does.  Each function cites the reference lines it follows.

Parity status: masks, encoder layer / MHA / cache handling, w2v2 streaming forward and the length-shrink
block are pinned against the reference's own functions (tests/golden/gen_golden.py drives them through
import shims; fixtures in tests/golden/).  The arithmetic of two un-vendored third-party packages is
restated from their published behaviour and is NOT pinned by any reference test ("parity unpinned"):
fairseq 0.12.2 ConvFeatureExtractionModel(mode=layer_norm, conv_bias) and rotary_embedding_torch
(RotaryEmbedding(dim=64, use_xpos=False).rotate_queries_with_cached_keys).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

NO_ROPE = "no-rope"  # the `rope` argument of the functions below under --rope 0 (cfg.enc_rope False)

ENC = "model.speech_encoder.speech_encoder."
SHR = "model.speech_encoder.length_shrink."
PRJ = "model.speech_encoder.proj."


# --------------------------------------------------------------------------------------------
# masks  (reference model/patches/patch_speech_encoder.py:30-50 and :52-77)
# --------------------------------------------------------------------------------------------
def attn_mask_training(seq_len: int, max_cache_size: Optional[int], blocksize: int) -> torch.Tensor:
    """Additive fp32 mask (seq_len, seq_len): query i sees keys j < end_of_block(i) and
    j >= i - max_cache_size  (patch_speech_encoder.py:30-50)."""
    i = torch.arange(seq_len).unsqueeze(1)
    j = torch.arange(seq_len).unsqueeze(0)
    block_end = torch.clamp((i // blocksize + 1) * blocksize, max=seq_len)
    allowed = j < block_end
    if max_cache_size is not None:
        allowed = allowed & (j >= i - max_cache_size)
    m = torch.zeros(seq_len, seq_len, dtype=torch.float32)
    m.masked_fill_(~allowed, float("-inf"))
    return m


def attn_mask_inference(seq_len: int, prefix_len: int, max_cache_size: int, blocksize: int) -> torch.Tensor:
    """Additive fp32 mask (seq_len, seq_len + min(prefix_len, max_cache_size))
    (patch_speech_encoder.py:52-77).  Column c is absolute frame c + max(0, prefix_len - max_cache_size)."""
    max_len = seq_len + min(prefix_len, max_cache_size)
    off = max(0, prefix_len - max_cache_size)
    total = seq_len + prefix_len
    i = torch.arange(seq_len).unsqueeze(1)
    c = torch.arange(max_len).unsqueeze(0)
    a = i + prefix_len  # absolute frame index of the query row
    block_end = torch.clamp((a // blocksize + 1) * blocksize, max=total)
    allowed = c < (block_end - off)
    lo = torch.clamp(i + prefix_len - max_cache_size, min=0) - off
    allowed = allowed & (c >= lo)
    m = torch.zeros(seq_len, max_len, dtype=torch.float32)
    m.masked_fill_(~allowed, float("-inf"))
    return m


# --------------------------------------------------------------------------------------------
# caches  (reference model/speech_encoder.py:80-97)
# --------------------------------------------------------------------------------------------
@dataclass
class LayerCache:
    k: Optional[torch.Tensor] = None  # (heads, T, head_dim), UNROTATED
    v: Optional[torch.Tensor] = None


@dataclass
class W2V2RoPECache:
    src: Optional[torch.Tensor] = None  # (1, n_samples) raw-sample cache
    src_len: int = 0  # number of frames already emitted for `src`
    n_steps: int = 0  # frames consumed by the transformer so far
    max_steps: int = 0  # sliding window (max_cache_size)
    layers: List[LayerCache] = field(default_factory=list)


# --------------------------------------------------------------------------------------------
# rotary embedding  [3P rotary_embedding_torch, restated; parity unpinned]
# --------------------------------------------------------------------------------------------
def enc_rope_tables(max_pos: int, head_dim: int, theta: float, mode: str, inv_freq: Optional[torch.Tensor] = None):
    """cos/sin tables (max_pos, head_dim//2) in fp32 for RotaryEmbedding(dim=head_dim, theta).

    mode "bf16": what the module computes after `speech_encoder.to(bf16)` (reference agents/infinisst.py:173):
    freqs Parameter rounded to bf16, positions `arange(..., dtype=bf16)`, angle product rounded to bf16,
    cos/sin rounded to bf16.  mode "fp32": everything fp32.
    """
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    if inv_freq is not None:  # the module's `freqs` Parameter as a checkpoint carries it (strict load_state_dict overwrites the initial value)
        inv = inv_freq.float()
    pos = torch.arange(max_pos, dtype=torch.float32)
    if mode == "bf16":
        inv = inv.bfloat16()
        posb = pos.bfloat16()
        ang = (posb.unsqueeze(1) * inv.unsqueeze(0))  # bf16 x bf16 -> bf16
        return ang.cos().float(), ang.sin().float()
    ang = pos.unsqueeze(1) * inv.unsqueeze(0)
    return ang.cos(), ang.sin()


def _rotate_interleaved(t: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, mode: str) -> torch.Tensor:
    """apply_rotary_emb with interleaved pairs (x0,x1),(x2,x3),...: out = t*cos + rotate_half(t)*sin,
    rotate_half((a,b)) = (-b,a).  t: (heads, T, hd); cos/sin: (T, hd/2) fp32."""
    cos2 = cos.repeat_interleave(2, dim=-1)
    sin2 = sin.repeat_interleave(2, dim=-1)
    t2 = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-t2[..., 1], t2[..., 0]), dim=-1).reshape(t.shape)
    if mode == "bf16" and t.dtype == torch.bfloat16:
        c = cos2.to(t.dtype)
        s = sin2.to(t.dtype)
        return (t * c) + (rot * s)  # each op rounds to bf16
    return (t.float() * cos2 + rot.float() * sin2).to(t.dtype)


def rotate_queries_with_cached_keys(q, k, cos_tab, sin_tab, mode: str):
    """q at offsets K-Q..K-1, all k at 0..K-1 (patch_speech_encoder.py:824, [3P] semantics)."""
    Q, K = q.shape[1], k.shape[1]
    q = _rotate_interleaved(q, cos_tab[K - Q: K], sin_tab[K - Q: K], mode)
    k = _rotate_interleaved(k, cos_tab[:K], sin_tab[:K], mode)
    return q, k


# --------------------------------------------------------------------------------------------
# absolute positions (--rope 0)  (patch_speech_encoder.py:448-461, :488-493)
# --------------------------------------------------------------------------------------------
def sinusoidal_positional_embedding(offset: int, length: int, d_model: int) -> torch.Tensor:
    """(length, d_model) bf16, rows offset .. offset+length-1: [sin(p * f_j) | cos(p * f_j)], f_j = exp(-j ln(1e4) / (d/2 - 1)).
    Everything is bf16 in the reference, whatever the model dtype: the index j, the frequencies, the positions themselves
    (integers above 256 round to the bf16 grid: 257 -> 256, 1001 -> 1000, ...), their product and sin / cos."""
    half = d_model // 2
    step = math.log(10000) / (half - 1)
    freq = torch.exp(torch.arange(half, dtype=torch.bfloat16) * -step)
    pos = torch.arange(offset, offset + length, dtype=torch.bfloat16)
    ang = pos.unsqueeze(1) * freq.unsqueeze(0)
    emb = torch.cat([torch.sin(ang), torch.cos(ang)], dim=1).view(length, -1)
    if d_model % 2 == 1:
        emb = torch.cat([emb, torch.zeros(length, 1)], dim=1)
    return emb


# --------------------------------------------------------------------------------------------
# conv feature extractor  [3P fairseq ConvFeatureExtractionModel(mode="layer_norm"), restated]
# call sites: patch_speech_encoder.py:245-251
# --------------------------------------------------------------------------------------------
def conv_feature_extractor(w: Dict[str, torch.Tensor], cfg, source: torch.Tensor,
                           return_layers: bool = False):
    """source (B, n_samples) -> (B, C, T).  Per layer: Conv1d(no padding) -> LayerNorm over channels
    (computed in fp32, Fp32LayerNorm) -> GELU."""
    x = source.unsqueeze(1)
    outs = []
    for i, (c, k, s) in enumerate(cfg.conv_layers):
        p = f"{ENC}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, w[p + "0.weight"], w.get(p + "0.bias"), stride=s)
        x = x.transpose(1, 2)
        x = F.layer_norm(x.float(), (c,), w[p + "2.1.weight"].float(), w[p + "2.1.bias"].float(), 1e-5).type_as(x)
        x = x.transpose(1, 2)
        x = F.gelu(x)
        if return_layers:
            outs.append(x)
    return (x, outs) if return_layers else x


def shrink_block(w: Dict[str, torch.Tensor], cfg, feature: torch.Tensor) -> torch.Tensor:
    """length_shrink (reference model/speech_encoder.py:18-78, :233): per layer Conv1d(no bias, no pad)
    -> LayerNorm over channels -> GELU.  feature (B, T, C) -> (B, T/4, C)."""
    x = feature.transpose(1, 2)
    for i, (c, k, s) in enumerate(cfg.shrink_layers):
        p = f"{SHR}conv_layers.{i}."
        x = F.conv1d(x, w[p + "0.weight"], None, stride=s)
        x = x.transpose(1, 2)
        x = F.layer_norm(x, (c,), w[p + "2.1.weight"], w[p + "2.1.bias"], 1e-5)
        x = x.transpose(1, 2)
        x = F.gelu(x)
    return x.transpose(1, 2)


# --------------------------------------------------------------------------------------------
# transformer encoder  (patch_speech_encoder.py:464-596, :692-933)
# --------------------------------------------------------------------------------------------
def mha_forward(w, cfg, prefix: str, x: torch.Tensor, attn_mask: torch.Tensor, cache: LayerCache,
                rope) -> torch.Tensor:
    """uni_mha_forward (patch_speech_encoder.py:692-933) for bsz == 1.  x: (T, 1, C)."""
    T, bsz, C = x.shape
    assert bsz == 1
    H, hd = cfg.enc_heads, cfg.enc_head_dim
    q = F.linear(x, w[prefix + "q_proj.weight"], w[prefix + "q_proj.bias"])
    k = F.linear(x, w[prefix + "k_proj.weight"], w[prefix + "k_proj.bias"])
    v = F.linear(x, w[prefix + "v_proj.weight"], w[prefix + "v_proj.bias"])
    q = q * (hd ** -0.5)  # :768, before RoPE
    q = q.contiguous().view(T, H, hd).transpose(0, 1)
    k = k.contiguous().view(T, H, hd).transpose(0, 1)
    v = v.contiguous().view(T, H, hd).transpose(0, 1)
    if cache.k is not None:  # :797-821  append UNROTATED k
        cache.k = torch.cat([cache.k.to(q), k], dim=1)
        cache.v = torch.cat([cache.v.to(q), v], dim=1)
        k, v = cache.k, cache.v
    else:
        cache.k, cache.v = k, v
    if rope is not NO_ROPE:  # :823
        cos_tab, sin_tab, mode = rope
        q, k = rotate_queries_with_cached_keys(q, k, cos_tab, sin_tab, mode)  # :824
    attn = torch.bmm(q, k.transpose(1, 2))  # :853, output in x.dtype
    attn += attn_mask.unsqueeze(0)  # :858-862 (in place, keeps dtype)
    attn_f = F.softmax(attn.float(), dim=-1)  # :887-889 utils.softmax -> fp32
    attn = attn_f.type_as(attn)  # :890
    out = torch.bmm(attn, v)  # :915
    out = out.transpose(0, 1).contiguous().view(T, bsz, C)  # :922
    return F.linear(out, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])  # :923


def encoder_layer(w, cfg, i: int, x, attn_mask, cache: LayerCache, rope):
    """uni_self_attn_forward, pre-LN (patch_speech_encoder.py:556-596)."""
    p = f"{ENC}encoder.layers.{i}."
    d = (cfg.enc_dim,)
    residual = x
    x = F.layer_norm(x, d, w[p + "self_attn_layer_norm.weight"], w[p + "self_attn_layer_norm.bias"], cfg.enc_ln_eps)
    x = mha_forward(w, cfg, p + "self_attn.", x, attn_mask, cache, rope)
    x = residual + x
    residual = x
    x = F.layer_norm(x, d, w[p + "final_layer_norm.weight"], w[p + "final_layer_norm.bias"], cfg.enc_ln_eps)
    x = F.gelu(F.linear(x, w[p + "fc1.weight"], w[p + "fc1.bias"]).float()).type_as(x)  # fairseq gelu: fp32 inside
    x = F.linear(x, w[p + "fc2.weight"], w[p + "fc2.bias"])
    return residual + x


def transformer_encoder(w, cfg, x: torch.Tensor, cache: W2V2RoPECache, blocksize: int, rope,
                        return_layers: bool = False):
    """uni_transformer_encoder_extract_features + _forward (patch_speech_encoder.py:464-554, :440-446).
    x: (1, T, C).  No positional conv; `rope is NO_ROPE` (--rope 0) adds the absolute sinusoid of the frames' stream positions
    instead of rotating q / k (:488-493)."""
    if rope is NO_ROPE:
        x = x + sinusoidal_positional_embedding(cache.n_steps, x.size(1), x.size(2))
    x = x.transpose(0, 1)  # T x B x C (:501)
    prefix_len, seq_len = cache.n_steps, x.size(0)
    if prefix_len > 0:  # :506-509
        mask = attn_mask_inference(seq_len, prefix_len, cache.max_steps, blocksize)
    else:
        mask = attn_mask_training(seq_len, cache.max_steps, blocksize)
    per_layer = []
    for i in range(cfg.enc_layers):
        lc = cache.layers[i]
        if lc.k is not None:  # :516-520 trim to the last max_steps BEFORE the layer call
            lc.k = lc.k[:, -cache.max_steps:]
            lc.v = lc.v[:, -cache.max_steps:]
        x = encoder_layer(w, cfg, i, x, mask, lc, rope)
        if return_layers:
            per_layer.append(x.transpose(0, 1))
    cache.n_steps += seq_len  # :533
    x = x.transpose(0, 1)
    x = F.layer_norm(x, (cfg.enc_dim,), w[ENC + "encoder.layer_norm.weight"], w[ENC + "encoder.layer_norm.bias"],
                     cfg.enc_ln_eps)  # :443-444
    return (x, per_layer) if return_layers else x


def w2v2_forward(w, cfg, source: torch.Tensor, cache: W2V2RoPECache, blocksize: int, rope,
                 return_intermediates: bool = False):
    """uni_w2v2_forward, features_only path (patch_speech_encoder.py:241-353).  source (1, n) in w dtype.

    The reference re-runs the conv stack over (cached samples + new samples) and keeps the new frames;
    this restatement does the same (the HIP path computes only the new frames -- identical values)."""
    if cache.src is not None:  # :241-243
        source = torch.cat([cache.src, source], dim=1)
    cache.src = source
    features = conv_feature_extractor(w, cfg, source)  # (1, C, T)  :245-251
    if cache.src_len > 0:  # :254-262
        new_src_len = features.size(-1)
        features = features[..., cache.src_len:]
        cache.src_len = new_src_len
        max_src = cfg.first_chunk_offset + cfg.samples_per_frame * blocksize  # 79 + 320 + 320*blocksize
        if cache.src.size(1) > max_src:
            cache.src = cache.src[:, -max_src:]
            cache.src_len = blocksize
    else:
        cache.src_len = features.size(-1)  # :264
    conv_out = features
    features = features.transpose(1, 2)  # :268
    features = F.layer_norm(features, (cfg.conv_dim,), w[ENC + "layer_norm.weight"], w[ENC + "layer_norm.bias"], 1e-5)
    x = F.linear(features, w[ENC + "post_extract_proj.weight"], w[ENC + "post_extract_proj.bias"])  # :301
    if return_intermediates:
        y, per_layer = transformer_encoder(w, cfg, x, cache, blocksize, rope, return_layers=True)
        return y, {"conv": conv_out.transpose(1, 2), "post_proj": x, "layers": per_layer}
    return transformer_encoder(w, cfg, x, cache, blocksize, rope)  # :340


def new_cache(cfg) -> W2V2RoPECache:
    """reference model/speech_encoder.py:220-224."""
    return W2V2RoPECache(max_steps=cfg.max_cache_size, layers=[LayerCache() for _ in range(cfg.enc_layers)])


def make_rope(cfg, max_pos: Optional[int] = None, inv_freq: Optional[torch.Tensor] = None):
    if not getattr(cfg, "enc_rope", True):
        return NO_ROPE
    n = max_pos or (cfg.max_cache_size + 8 * cfg.block_size * 4)
    cos, sin = enc_rope_tables(n, cfg.enc_head_dim, cfg.enc_rope_theta, cfg.enc_rope_mode, inv_freq)
    return cos, sin, cfg.enc_rope_mode


def encode_speech(w, cfg, src_tokens: torch.Tensor, cache: Optional[W2V2RoPECache], multiplier: int = 1,
                  rope=None, return_intermediates: bool = False):
    """SpeechEncoderW2V2RoPE.encode_speech with set_blocksize(multiplier)
    (reference model/speech_encoder.py:143-145, :219-236).  Returns ((1, S, llm_dim), cache)."""
    if cache is None:
        cache = new_cache(cfg)
    if rope is None:
        rope = make_rope(cfg)
    blocksize = cfg.block_size * multiplier
    if return_intermediates:
        feat, inter = w2v2_forward(w, cfg, src_tokens, cache, blocksize, rope, True)
    else:
        feat = w2v2_forward(w, cfg, src_tokens, cache, blocksize, rope)
    shr = shrink_block(w, cfg, feat)  # :233
    out = F.linear(shr, w[PRJ + "weight"], w[PRJ + "bias"])  # :234
    if return_intermediates:
        inter["enc_out"] = feat
        inter["shrink"] = shr
        return out, cache, inter
    return out, cache


def feat_extract_output_lengths(cfg, n_samples: int) -> int:
    """_get_feat_extract_output_lengths (reference model/speech_encoder.py:202-217)."""
    n = n_samples
    for _, k, s in list(cfg.conv_layers) + list(cfg.shrink_layers):
        n = (n - k) // s + 1
    return n
