"""Model / runtime configuration for the InfiniSST hot path.

The dimensions are not recorded in the reference repo itself; they come from the
public wav2vec2-large (LV-60) and Llama-3.1-8B configs and from the production
inference script (reference scripts/infer/infinisst.sh:42-87).
The flag names mirror reference agents/options.py and agents/infinisst.py:185-198.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class ModelConfig:
    # --- wav2vec2 conv feature extractor (fairseq ConvFeatureExtractionModel, mode=layer_norm)
    conv_layers: List[Tuple[int, int, int]] = field(
        default_factory=lambda: [(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2
    )
    conv_bias: bool = True
    # --- streaming transformer encoder (reference model/patches/patch_speech_encoder.py)
    enc_dim: int = 1024
    enc_layers: int = 24
    enc_heads: int = 16  # head_dim must be 64
    enc_ffn: int = 4096
    enc_ln_eps: float = 1e-5
    block_size: int = 48  # --block-size (frames of 20 ms per block at multiplier 1)
    max_cache_size: int = 576  # --max-cache-size (sliding window of encoder KV, frames)
    enc_rope_theta: float = 10000.0
    # semantic of the un-vendored rotary_embedding_torch after `.to(bf16)`
    # (reference agents/infinisst.py:173): "bf16" = positions, inv_freq, angles, cos/sin and
    # every product rounded to bf16; "fp32" = fp32 tables, one rounding at the end.
    enc_rope_mode: str = "bf16"
    # --rope (reference agents/options.py:37-41, patch_speech_encoder.py:488-493, :823): False = no rotation of q / k, the bf16
    # sinusoid of each frame's stream position is added to the encoder input instead
    enc_rope: bool = True
    # --- length shrink + projector (reference model/speech_encoder.py:117-121)
    shrink_layers: List[Tuple[int, int, int]] = field(default_factory=lambda: [(1024, 2, 2)] * 2)
    # --- Llama decoder
    llm_dim: int = 4096
    llm_layers: int = 32
    llm_heads: int = 32
    llm_kv_heads: int = 8
    llm_head_dim: int = 128
    llm_ffn: int = 14336
    vocab: int = 128263  # 128256 + <sp_patch>,<sp_start>,<sp_end> + <latency_1..4>
    rms_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 8.0
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_pos: int = 8192
    # --- special token ids the splice logic needs (reference model/llm.py:181-183)
    sp_patch_id: int = 128256
    start_header_id: int = 128006
    end_header_id: int = 128007
    eot_id: int = 128009
    bos_id: int = 128000
    user_id: int = 882
    assistant_id: int = 78191
    system_id: int = 9125
    nl2_id: int = 271
    eos_ids: Tuple[int, ...] = (128001, 128008, 128009)
    pad_id: int = 128004  # <|finetune_right_pad_id|>

    @property
    def enc_head_dim(self) -> int:
        return self.enc_dim // self.enc_heads

    @property
    def conv_dim(self) -> int:
        return self.conv_layers[-1][0]

    @property
    def samples_per_frame(self) -> int:
        s = 1
        for _, _, st in self.conv_layers:
            s *= st
        return s  # 320

    @property
    def receptive_field(self) -> int:
        # samples spanned by one output frame (400 for wav2vec2)
        r = 1
        for _, k, st in reversed(self.conv_layers):
            r = (r - 1) * st + k
        return r

    @property
    def shrink_factor(self) -> int:
        s = 1
        for _, _, st in self.shrink_layers:
            s *= st
        return s  # 4

    @property
    def chunk_samples(self) -> int:
        # reference agents/infinisst.py:201  int(block_size // 4 * 0.08 * 16000)
        return int(self.block_size // 4 * 0.08 * 16000)

    @property
    def first_chunk_offset(self) -> int:
        # reference agents/infinisst.py:217   79 + 320 zeros in front of the first chunk
        return self.receptive_field - 1  # 399

    def replace(self, **kw) -> "ModelConfig":
        return dataclasses.replace(self, **kw)


def full_config() -> ModelConfig:
    """wav2vec2-large + Llama-3.1-8B (BASELINE.json configs[1])."""
    return ModelConfig()


def toy_config() -> ModelConfig:
    """Small instance with the same structure, for parity tests the CPU oracle finishes in seconds.

    head dims stay 64 (encoder) / 128 (LLM) because the HIP attention kernels are specialised on them.
    """
    return ModelConfig(
        conv_layers=[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2,
        enc_dim=128,
        enc_layers=2,
        enc_heads=2,
        enc_ffn=256,
        shrink_layers=[(128, 2, 2)] * 2,
        llm_dim=256,
        llm_layers=2,
        llm_heads=4,
        llm_kv_heads=2,
        llm_head_dim=128,
        llm_ffn=512,
        vocab=1031,
        sp_patch_id=1024,
        start_header_id=1006,
        end_header_id=1007,
        eot_id=1009,
        bos_id=1000,
        user_id=882,
        assistant_id=781,
        system_id=912,
        nl2_id=271,
        eos_ids=(1001, 1008, 1009),
        pad_id=1004,
    )


@dataclass
class GenConfig:
    """Generation / cache flags (reference agents/options.py:43-108, agents/infinisst.py:185-198;
    production values from scripts/infer/infinisst.sh:42-87)."""

    latency_multiplier: int = 1
    max_new_tokens: int = 10
    beam: int = 1  # greedy; the reference asserts beam > 1 (agents/infinisst.py:86), see DESIGN.md
    no_repeat_ngram_size: int = 5
    no_repeat_ngram_lookback: int = 100
    repetition_penalty: float = 1.2
    max_llm_cache_size: int = 1000
    always_cache_system_prompt: bool = True
    suppress_tokens: Tuple[int, ...] = ()
    # the sample branch (agents/options.py:43-108 --do-sample / --top-p / --top-k / --epsilon-cutoff / --temperature; greedy when do_sample is off)
    do_sample: bool = False
    temperature: float = 1.0
    top_k: int = 0
    top_p: float = 1.0
    epsilon_cutoff: float = 0.0
    seed: int = 998244353  # reference agents/infinisst.py:74
