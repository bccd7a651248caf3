"""Synthetic weights / audio / prompts for benchmarks and parity tests.

Key names follow the reference checkpoint layout (agents/infinisst.py:176-180, train/prune_bin.py:5-11,
model/speech_encoder.py:111-121): `model.*` / `lm_head.*` for the Llama part and
`model.speech_encoder.*` for the wav2vec2 encoder, length-shrink convs and projector.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from .config import ModelConfig

SEED = 998244353  # reference agents/infinisst.py:74

ENC = "model.speech_encoder.speech_encoder."
SHR = "model.speech_encoder.length_shrink."
PRJ = "model.speech_encoder.proj."


def weight_shapes(cfg: ModelConfig) -> Dict[str, tuple]:
    """name -> shape for every tensor the hot path reads."""
    s: Dict[str, tuple] = {}
    cin = 1
    for i, (c, k, _) in enumerate(cfg.conv_layers):
        s[f"{ENC}feature_extractor.conv_layers.{i}.0.weight"] = (c, cin, k)
        if cfg.conv_bias:
            s[f"{ENC}feature_extractor.conv_layers.{i}.0.bias"] = (c,)
        s[f"{ENC}feature_extractor.conv_layers.{i}.2.1.weight"] = (c,)
        s[f"{ENC}feature_extractor.conv_layers.{i}.2.1.bias"] = (c,)
        cin = c
    s[f"{ENC}layer_norm.weight"] = (cin,)
    s[f"{ENC}layer_norm.bias"] = (cin,)
    d = cfg.enc_dim
    s[f"{ENC}post_extract_proj.weight"] = (d, cin)
    s[f"{ENC}post_extract_proj.bias"] = (d,)
    for i in range(cfg.enc_layers):
        p = f"{ENC}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (d, d)
            s[p + f"self_attn.{n}.bias"] = (d,)
        s[p + "self_attn_layer_norm.weight"] = (d,)
        s[p + "self_attn_layer_norm.bias"] = (d,)
        s[p + "fc1.weight"] = (cfg.enc_ffn, d)
        s[p + "fc1.bias"] = (cfg.enc_ffn,)
        s[p + "fc2.weight"] = (d, cfg.enc_ffn)
        s[p + "fc2.bias"] = (d,)
        s[p + "final_layer_norm.weight"] = (d,)
        s[p + "final_layer_norm.bias"] = (d,)
    s[f"{ENC}encoder.layer_norm.weight"] = (d,)
    s[f"{ENC}encoder.layer_norm.bias"] = (d,)
    cin = d
    for i, (c, k, _) in enumerate(cfg.shrink_layers):
        s[f"{SHR}conv_layers.{i}.0.weight"] = (c, cin, k)
        s[f"{SHR}conv_layers.{i}.2.1.weight"] = (c,)
        s[f"{SHR}conv_layers.{i}.2.1.bias"] = (c,)
        cin = c
    s[f"{PRJ}weight"] = (cfg.llm_dim, cin)
    s[f"{PRJ}bias"] = (cfg.llm_dim,)
    D = cfg.llm_dim
    s["model.embed_tokens.weight"] = (cfg.vocab, D)
    for i in range(cfg.llm_layers):
        p = f"model.layers.{i}."
        s[p + "input_layernorm.weight"] = (D,)
        s[p + "self_attn.q_proj.weight"] = (cfg.llm_heads * cfg.llm_head_dim, D)
        s[p + "self_attn.k_proj.weight"] = (cfg.llm_kv_heads * cfg.llm_head_dim, D)
        s[p + "self_attn.v_proj.weight"] = (cfg.llm_kv_heads * cfg.llm_head_dim, D)
        s[p + "self_attn.o_proj.weight"] = (D, cfg.llm_heads * cfg.llm_head_dim)
        s[p + "post_attention_layernorm.weight"] = (D,)
        s[p + "mlp.gate_proj.weight"] = (cfg.llm_ffn, D)
        s[p + "mlp.up_proj.weight"] = (cfg.llm_ffn, D)
        s[p + "mlp.down_proj.weight"] = (D, cfg.llm_ffn)
    s["model.norm.weight"] = (D,)
    s["lm_head.weight"] = (cfg.vocab, D)
    return s


def _is_norm_weight(name: str) -> bool:
    return name.endswith("norm.weight") or name.endswith(".2.1.weight")


PEAKED_EMBED_SCALE = 16.0   # "peaked" recipe: embedding rows scaled by this
PEAKED_TOP_LOGIT = 40.0     # target logit of the first structured continuation
PEAKED_LEVELS = 8           # structured continuations per token ...
PEAKED_LEVEL_RATIO = 0.8    # ... with geometrically decreasing scores (40, 32, 25.6, ...)
PEAKED_BRANCH_SHARE = 0.5   # |sum of all residual-branch outputs| aimed at this fraction of the embedding's norm
PEAKED_BULK_SCALE = 0.5     # the random (unstructured) part of lm_head is scaled by this


def peaked_permutations(cfg: ModelConfig):
    """PEAKED_LEVELS fixed permutations of the vocabulary (numpy int64)."""
    rng = np.random.default_rng(SEED + 29)
    return [rng.permutation(cfg.vocab) for _ in range(PEAKED_LEVELS)]


def peaked_successors(cfg: ModelConfig, token: int, perms=None):
    """The structured continuations of `token` under the peaked recipe, best first: pi_l^-1(token)."""
    perms = perms if perms is not None else peaked_permutations(cfg)
    return [int(np.nonzero(p == token)[0][0]) for p in perms]


def apply_recipe(cfg: ModelConfig, w: Dict[str, torch.Tensor], recipe: str) -> None:
    """In-place variants of the N(0, std^2) init (`plain`: SURVEY 8(d), what the benchmark runs).

    `peaked` (parity tests of greedy TOKEN IDS).  With the plain init the logits are 128 k nearly exchangeable Gaussians: the
    top-2 margin is ~0.2 sigma while bf16 arithmetic alone moves a logit by 2-5 % of sigma (measured, profiles/peaked_recipe_probe.py:
    the same with 1/sqrt(2L)-scaled residual branches, so it is rounding, not chaos); argmax is a coin flip on a quarter of the steps
    and id parity cannot be asserted.  ANY contest between Gaussian scores has that problem (a loud sub-vocabulary of 25 tokens: 79 %
    decisive steps), so the peak has to be structural, as in a trained model whose output embedding is aligned with what the residual
    stream carries:
      * embedding rows x PEAKED_EMBED_SCALE; the residual-branch output projections (o_proj, down_proj) scaled so that all the
        branches together add about PEAKED_BRANCH_SHARE of the embedding's norm (with the plain init ONE MLP adds a vector several
        times longer than the embedding and the stream forgets its token; the factor follows from E[silu(g)^2] of the init);
      * lm_head[v] = PEAKED_BULK_SCALE * random + gamma * sum_l PEAKED_LEVEL_RATIO^l * embed[pi_l(v)] for PEAKED_LEVELS fixed
        permutations, gamma set for a top logit of ~PEAKED_TOP_LOGIT.
    The hidden state of a position is (its token's embedding + what the layers add), so the tokens pi_l^-1(last token) score
    ~40, 32, 25.6, ... and the rest stays a Gaussian bulk with a maximum of ~6-8: greedy decoding walks a pseudo-random chain whose
    every step is decided by a margin tens of times the bf16 noise; when the no-repeat-n-gram processors ban a continuation (every
    chunk restarts from the same prompt tail, so step 5 of a chunk finds its 5-gram in the previous chunks' ids) the next level
    takes over, and only when all levels are banned does a step fall back to the bulk (a non-decisive step).  Everything else -- the
    projections, norms, the speech encoder -- is the plain recipe; the logits still depend on the whole network (bulk values and the
    exact heights of the peaks), which the teacher-forced logit tests measure."""
    if recipe == "plain":
        return
    if recipe != "peaked":
        raise ValueError(f"unknown weight recipe {recipe!r}")
    D, F_, L = cfg.llm_dim, cfg.llm_ffn, cfg.llm_layers
    std = float(w["model.layers.0.mlp.gate_proj.weight"].float().std())
    gs = torch.Generator().manual_seed(SEED)
    gsmp = torch.randn(200000, generator=gs) * (std * np.sqrt(D))
    act_rms = float(torch.sqrt((torch.nn.functional.silu(gsmp) ** 2).mean())) * std * float(np.sqrt(D))  # rms of silu(g) * u
    mlp_branch = std * act_rms * float(np.sqrt(F_)) * float(np.sqrt(D))   # norm one plain MLP adds to the residual stream
    e_norm = PEAKED_EMBED_SCALE * std * float(np.sqrt(D))
    scale = PEAKED_BRANCH_SHARE * e_norm / (float(np.sqrt(L)) * mlp_branch)
    for i in range(L):
        for n in ("self_attn.o_proj.weight", "mlp.down_proj.weight"):
            k = f"model.layers.{i}.{n}"
            w[k] = (w[k].float() * scale).to(w[k].dtype)
    emb = w["model.embed_tokens.weight"]
    w["model.embed_tokens.weight"] = (emb.float() * PEAKED_EMBED_SCALE).to(emb.dtype)
    e = w["model.embed_tokens.weight"].float()  # the tied part is built from the ROUNDED embedding the model will read
    x_norm = e_norm * float(np.sqrt(1.0 + PEAKED_BRANCH_SHARE ** 2))
    gamma = PEAKED_TOP_LOGIT * x_norm / (e_norm ** 2 * float(np.sqrt(D)))
    lm = w["lm_head.weight"]
    out = lm.float() * PEAKED_BULK_SCALE
    for lvl, p in enumerate(peaked_permutations(cfg)):
        out += (gamma * PEAKED_LEVEL_RATIO ** lvl) * e[torch.from_numpy(p).to(lm.device)]
    w["lm_head.weight"] = out.to(lm.dtype)


def random_weights(cfg: ModelConfig, dtype=torch.bfloat16, device="cpu", seed: int = SEED,
                   std: float = 0.02, norm_jitter: float = 0.0, recipe: str = "plain") -> Dict[str, torch.Tensor]:
    """Linear/Conv/Embedding ~ N(0, std^2), norm weights 1, biases 0.

    `norm_jitter` > 0 perturbs norm weights and all biases (parity tests use it so that a kernel that
    ignores a bias or a norm weight cannot pass).  Tensors are generated one by one from a single
    generator, so the values depend only on (cfg, seed, std, norm_jitter) and not on the device.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        if _is_norm_weight(name):
            t = torch.ones(shape)
            if norm_jitter:
                t = t + norm_jitter * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = torch.zeros(shape)
            if norm_jitter:
                t = norm_jitter * torch.randn(shape, generator=g)
        else:
            t = std * torch.randn(shape, generator=g)
        out[name] = t.to(dtype).to(device)
    apply_recipe(cfg, out, recipe)
    return out


def random_weights_device(cfg: ModelConfig, device, dtype=torch.bfloat16, seed: int = SEED,
                          std: float = 0.02, recipe: str = "plain") -> Dict[str, torch.Tensor]:
    """Same distribution as `random_weights` but drawn on the device (8B parameters in seconds).
    Values differ from the CPU generator's; use `random_weights` when an oracle must see the same tensors (or copy these to the host).
    `recipe`: see `apply_recipe`."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for name, shape in weight_shapes(cfg).items():
        if _is_norm_weight(name):
            t = torch.ones(shape, device=device, dtype=dtype)
        elif name.endswith(".bias"):
            t = torch.zeros(shape, device=device, dtype=dtype)
        else:
            t = torch.empty(shape, device=device, dtype=torch.float32).normal_(0.0, std, generator=g).to(dtype)
        out[name] = t
    apply_recipe(cfg, out, recipe)
    return out


def synthetic_audio(n_samples: int, stream_id: int = 0) -> np.ndarray:
    """16 kHz mono fp32, clip(0.1*N(0,1), -1, 1) with default_rng(SEED + stream_id)."""
    rng = np.random.default_rng(SEED + stream_id)
    return np.clip(0.1 * rng.standard_normal(n_samples), -1.0, 1.0).astype(np.float32)


def system_prompt_ids(cfg: ModelConfig, multiplier: int = 1, n_filler: int = 38) -> List[int]:
    """Stand-in for the Llama-3.1 chat-template system message (no tokenizer files in this image).

    Layout follows reference agents/infinisst.py:228-241 / SURVEY.md 8(a2):
    [BOS, SH, system, EH, \\n\\n, <filler: date block + instruction + latency token>, EOT].
    """
    rng = np.random.default_rng(SEED + 17)
    lim = min(cfg.sp_patch_id, cfg.start_header_id, cfg.bos_id) - 1
    filler = [int(x) for x in rng.integers(3, lim, size=n_filler)]
    filler = [t for t in filler if t not in (cfg.user_id, cfg.assistant_id)]
    latency_token = cfg.sp_patch_id + 2 + multiplier  # <latency_m>
    return [cfg.bos_id, cfg.start_header_id, cfg.system_id, cfg.end_header_id, cfg.nl2_id] + filler + [
        latency_token, cfg.eot_id]


def chunk_prompt_ids(cfg: ModelConfig, multiplier: int = 1, first: bool = False) -> List[int]:
    """Token ids of one chunk's prompt (reference agents/infinisst.py:242-264).

    later chunks: [EOT, SH, user, EH, \\n\\n, S x <sp_patch>, EOT, SH, assistant, EH, \\n\\n]
    (the leading EOT closes the previous assistant turn; 22 tokens at multiplier 1);
    first chunk: system prompt + the same without the leading EOT.
    """
    n_sp = cfg.block_size // 4 * multiplier
    turn = [cfg.start_header_id, cfg.user_id, cfg.end_header_id, cfg.nl2_id] + [cfg.sp_patch_id] * n_sp + [
        cfg.eot_id, cfg.start_header_id, cfg.assistant_id, cfg.end_header_id, cfg.nl2_id]
    if first:
        return system_prompt_ids(cfg, multiplier) + turn
    return [cfg.eot_id] + turn
