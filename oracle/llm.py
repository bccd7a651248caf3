"""ORACLE (test infrastructure, not product code) -- CPU restatement of the reference Speech-Llama forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows reference model/llm.py:51-126,192-295 (speech splice, lm_head over all positions) and
model/patches/patch_llm.py:231-336 (attention that caches UNROTATED K and re-applies RoPE to the whole K
cache with positions 0..T-1 on every call).  The surrounding Llama blocks (RMSNorm, llama3 rotary table,
apply_rotary_pos_emb, repeat_kv, SwiGLU MLP, causal mask for attention_mask=None) live in the un-vendored
transformers==4.47.0 and are restated from that release's published code ("parity unpinned" for those;
tests cross-check them against the transformers build present in this image as a secondary reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# ---- [3P transformers 4.47.0] LlamaRMSNorm (pinned bit-exactly to the image's transformers 5.15.0: tests/golden/llama_blocks.npz) ----
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    dt = x.dtype
    h = x.float()
    var = h.pow(2).mean(-1, keepdim=True)
    h = h * torch.rsqrt(var + eps)
    return weight * h.to(dt)  # cast to input dtype BEFORE the weight multiply


# ---- [3P transformers 4.47.0] LlamaRotaryEmbedding(rope_type="llama3") ----------------------
def llama3_inv_freq(cfg) -> torch.Tensor:
    dim = cfg.llm_head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).float() / dim))
    factor, lo, hi, old = cfg.rope_factor, cfg.rope_low_freq_factor, cfg.rope_high_freq_factor, cfg.rope_original_max_pos
    low_wl, high_wl = old / lo, old / hi
    wavelen = 2 * math.pi / inv
    inv_l = torch.where(wavelen > low_wl, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
    return torch.where(medium, smoothed, inv_l)


def llm_rope_tables(cfg, max_pos: int, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin (max_pos, head_dim): fp32 angles, emb = cat(freqs, freqs), cast to the activation dtype."""
    inv = llama3_inv_freq(cfg)
    freqs = torch.arange(max_pos, dtype=torch.float32).unsqueeze(1) * inv.unsqueeze(0)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x (1, heads, T, hd); cos/sin (T, hd) in x.dtype: (x*cos) + (rotate_half(x)*sin), each op in x.dtype."""
    return (x * cos.unsqueeze(0).unsqueeze(0)) + (rotate_half(x) * sin.unsqueeze(0).unsqueeze(0))


# ---- KV cache: list over layers of [K, V], each (1, kv_heads, T, hd), K UNROTATED -------------
def new_kv(cfg) -> List[List[Optional[torch.Tensor]]]:
    return [[None, None] for _ in range(cfg.llm_layers)]


def kv_len(kv) -> int:
    return 0 if kv[0][0] is None else kv[0][0].size(2)


def attention(w, cfg, i: int, h: torch.Tensor, kv, rope) -> torch.Tensor:
    """llama_sdpa_attention_new_forward (reference model/patches/patch_llm.py:231-336), bsz == 1.
    h (1, q_len, D)."""
    p = f"model.layers.{i}.self_attn."
    bsz, q_len, _ = h.shape
    H, KV, hd = cfg.llm_heads, cfg.llm_kv_heads, cfg.llm_head_dim
    q = F.linear(h, w[p + "q_proj.weight"]).view(bsz, q_len, H, hd).transpose(1, 2)  # :260-267
    k = F.linear(h, w[p + "k_proj.weight"]).view(bsz, q_len, KV, hd).transpose(1, 2)
    v = F.linear(h, w[p + "v_proj.weight"]).view(bsz, q_len, KV, hd).transpose(1, 2)
    if kv[i][0] is not None:  # :280-284 cache.update(unrotated K, V)
        k = torch.cat([kv[i][0], k], dim=2)
        v = torch.cat([kv[i][1], v], dim=2)
    kv[i][0], kv[i][1] = k, v
    total = k.size(2)
    past = total - q_len
    cos, sin = rope
    q = apply_rope(q, cos[past:total], sin[past:total])  # :291,:295,:298  positions past..total-1
    k = apply_rope(k, cos[:total], sin[:total])  # :290,:294,:299  ALL keys at 0..total-1
    k = k.repeat_interleave(H // KV, dim=1)  # repeat_kv :304-305
    vv = v.repeat_interleave(H // KV, dim=1)
    # [3P] HF builds no mask when q_len == 1, is_causal when past == 0, else a lower-right aligned
    # causal mask -- all three are "query at absolute position p sees keys <= p".
    scores = torch.matmul(q.float(), k.float().transpose(2, 3)) / math.sqrt(hd)
    qi = torch.arange(past, total).unsqueeze(1)
    kj = torch.arange(total).unsqueeze(0)
    scores = scores.masked_fill(~(kj <= qi), float("-inf"))
    probs = F.softmax(scores, dim=-1).to(h.dtype)
    out = torch.matmul(probs.float(), vv.float()).to(h.dtype)  # fp32 accumulate, one rounding
    out = out.transpose(1, 2).contiguous().view(bsz, q_len, H * hd)  # :331-332
    return F.linear(out, w[p + "o_proj.weight"])  # :334


def mlp(w, i: int, x: torch.Tensor) -> torch.Tensor:
    """[3P] LlamaMLP: down(silu(gate(x)) * up(x))."""
    p = f"model.layers.{i}.mlp."
    g = F.linear(x, w[p + "gate_proj.weight"])
    u = F.linear(x, w[p + "up_proj.weight"])
    return F.linear(F.silu(g) * u, w[p + "down_proj.weight"])


def decoder_stack(w, cfg, x: torch.Tensor, kv, rope, return_layers: bool = False):
    """[3P] LlamaModel layers + final norm.  x (1, T, D)."""
    per_layer = []
    for i in range(cfg.llm_layers):
        p = f"model.layers.{i}."
        r = x
        h = rmsnorm(x, w[p + "input_layernorm.weight"], cfg.rms_eps)
        x = r + attention(w, cfg, i, h, kv, rope)
        r = x
        h = rmsnorm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_eps)
        x = r + mlp(w, i, h)
        if return_layers:
            per_layer.append(x)
    x = rmsnorm(x, w["model.norm.weight"], cfg.rms_eps)
    return (x, per_layer) if return_layers else x


def splice_speech(cfg, input_ids: torch.Tensor, embeds: torch.Tensor, speech: torch.Tensor) -> torch.Tensor:
    """Overwrite prompt embeddings with speech features (reference model/llm.py:86-113), batch 1.

    For each (`user`, `assistant`) header pair (token preceded by <|start_header_id|>):
    embeds[u+3 : a-2] <- speech[index : index + (a-u-5)].
    input_ids (T,), embeds (T, D), speech (S, D)."""
    ids = input_ids.tolist()
    users = [j for j, t in enumerate(ids) if t == cfg.user_id and j > 0 and ids[j - 1] == cfg.start_header_id]
    assists = [j for j, t in enumerate(ids) if t == cfg.assistant_id and j > 0 and ids[j - 1] == cfg.start_header_id]
    out = embeds
    index = 0
    for u, a in zip(users, assists):
        n = a - u - 5
        out = torch.cat([out[: u + 3], speech[index: index + n], out[a - 2:]], dim=0)
        index += n
    return out


def model_forward(w, cfg, input_ids: torch.Tensor, kv, rope, speech: Optional[torch.Tensor] = None,
                  all_logits: bool = False, return_layers: bool = False):
    """SpeechLlamaForCausalLM.forward at batch 1 (reference model/llm.py:51-126, :192-270).

    Step 0 of a chunk passes `speech` ((S, D) encoder output) and the whole prompt; later steps pass only the
    last token (model/llm.py:114-115).  Returns logits of the last position (1-D, model dtype) -- the reference
    computes lm_head on every position (model/llm.py:237) and `_sample` then keeps `[:, -1]`; `all_logits`
    returns all rows."""
    emb = F.embedding(input_ids, w["model.embed_tokens.weight"])
    if speech is not None:
        emb = splice_speech(cfg, input_ids, emb, speech)
    x = emb.unsqueeze(0)
    if return_layers:
        x, per = decoder_stack(w, cfg, x, kv, rope, True)
    else:
        x = decoder_stack(w, cfg, x, kv, rope)
    if all_logits:
        logits = F.linear(x[0], w["lm_head.weight"])
    else:
        logits = F.linear(x[0, -1:], w["lm_head.weight"])[0]
    return (logits, per) if return_layers else logits
