"""ctypes binding of libinfinisst_hip.so (include/infinisst_hip.h) -- the only compute path of this package.

There is no CPU fallback: if the HIP library is missing or no GPU is visible, construction fails loudly.
PyTorch is used here for plumbing only (device tensors handed over as raw pointers, the current HIP stream).
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import time
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import rope
from .config import GenConfig, ModelConfig

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libinfinisst_hip.so")
_lib = None

ISST_MAX_CONV, ISST_MAX_SHRINK, ISST_MAX_EOS = 8, 4, 8

EPI = {"none": 0, "bias": 1, "bias_gelu": 2, "res": 3, "bias_res": 4, "swiglu": 5, "f32": 6, "swiglu8": 8}


class IsstError(RuntimeError):
    pass


class _Config(C.Structure):
    _fields_ = [
        ("n_conv", C.c_int), ("conv_dim", C.c_int * ISST_MAX_CONV), ("conv_k", C.c_int * ISST_MAX_CONV),
        ("conv_stride", C.c_int * ISST_MAX_CONV), ("conv_bias", C.c_int),
        ("enc_dim", C.c_int), ("enc_layers", C.c_int), ("enc_heads", C.c_int), ("enc_ffn", C.c_int),
        ("enc_ln_eps", C.c_float), ("block_size", C.c_int), ("max_cache_size", C.c_int),
        ("enc_rope_round_each", C.c_int),
        ("n_shrink", C.c_int), ("shrink_dim", C.c_int * ISST_MAX_SHRINK), ("shrink_k", C.c_int * ISST_MAX_SHRINK),
        ("shrink_stride", C.c_int * ISST_MAX_SHRINK),
        ("llm_dim", C.c_int), ("llm_layers", C.c_int), ("llm_heads", C.c_int), ("llm_kv_heads", C.c_int),
        ("llm_ffn", C.c_int), ("vocab", C.c_int), ("rms_eps", C.c_float),
        ("user_id", C.c_int), ("assistant_id", C.c_int), ("start_header_id", C.c_int),
        ("n_eos", C.c_int), ("eos_ids", C.c_int * ISST_MAX_EOS),
        ("max_streams", C.c_int), ("max_multiplier", C.c_int), ("max_prompt_len", C.c_int),
        ("max_new_tokens", C.c_int), ("max_llm_cache_size", C.c_int), ("max_system_prompt", C.c_int),
        ("debug_taps", C.c_int), ("max_beams", C.c_int), ("enc_abs_pos", C.c_int),
    ]


class _GenParams(C.Structure):
    _fields_ = [
        ("multiplier", C.c_int), ("max_new_tokens", C.c_int), ("no_repeat_ngram_size", C.c_int),
        ("encoder_no_repeat_ngram_size", C.c_int), ("repetition_penalty", C.c_float),
        ("suppress_tokens", C.POINTER(C.c_int)), ("n_suppress", C.c_int), ("system_prompt_size", C.c_int),
        ("num_beams", C.c_int), ("length_penalty", C.c_float), ("pcm_on_device", C.c_int),
        ("do_sample", C.c_int), ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_float), ("epsilon_cutoff", C.c_float),
        ("seed", C.c_ulonglong),
    ]


class _StreamInfo(C.Structure):
    _fields_ = [("llm_cache_len", C.c_int), ("llm_sys_len", C.c_int), ("enc_n_steps", C.c_int),
                ("enc_cache_len", C.c_int), ("chunks", C.c_int)]


# every symbol include/infinisst_hip.h declares
EXPORTS = [
    "isst_create", "isst_destroy", "isst_last_error", "isst_load_weight", "isst_set_rope_tables", "isst_set_enc_position_table",
    "isst_finalize_weights", "isst_stream_open", "isst_stream_reset", "isst_stream_close", "isst_stream_info_get",
    "isst_stream_import_llm_kv", "isst_stream_import_enc_kv", "isst_stream_import_audio_history",
    "isst_debug_beam_trace_begin", "isst_debug_beam_trace_step", "isst_debug_beam_trace_end",
    "isst_op_attn_combine", "isst_op_gemm_attn_merge", "isst_op_splice_map", "isst_op_embed_splice", "isst_op_enc_attention", "isst_op_llm_attention",
    "isst_generate", "isst_kv_evict", "isst_encode_speech", "isst_debug_tap", "isst_debug_read_kv", "isst_profile_begin", "isst_profile_begin_rows", "isst_profile_end", "isst_op_pack_weight", "isst_op_pack_gateup8",
    "isst_op_packed_elems", "isst_op_gemm", "isst_op_gemm_splitk_rmsnorm", "isst_op_gemm_splitk_fused", "isst_op_gemm_splitk_plain", "isst_op_gemm_norm_ssq", "isst_op_gemm_splitk_layernorm", "isst_op_set_gemm_tuning", "isst_op_set_attn_tuning", "isst_op_set_reduce_tuning", "isst_op_topk_rows", "isst_op_layernorm", "isst_op_rmsnorm", "isst_op_conv0", "isst_op_sample", "isst_op_warp", "isst_op_warp_sample", "isst_op_sample_uniform", "isst_op_multinomial_wor",
]


def load_library(path: Optional[str] = None):
    """dlopen the HIP library; raises if it has not been built (python __graft_entry__.py / make -C csrc)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _LIB_PATH
    if not os.path.exists(p):
        raise IsstError(f"{p} not found: build it with `make -C infinisst_amd/csrc` (no CPU fallback exists)")
    lib = C.CDLL(p)
    lib.isst_last_error.restype = C.c_char_p
    lib.isst_last_error.argtypes = [C.c_void_p]
    lib.isst_destroy.restype = None
    lib.isst_destroy.argtypes = [C.c_void_p]
    lib.isst_op_packed_elems.restype = C.c_int64
    lib.isst_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
    lib.isst_load_weight.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    lib.isst_set_rope_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.isst_set_enc_position_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.isst_finalize_weights.argtypes = [C.c_void_p]
    lib.isst_stream_open.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.isst_stream_reset.argtypes = [C.c_void_p, C.c_int]
    lib.isst_stream_close.argtypes = [C.c_void_p, C.c_int]
    lib.isst_stream_info_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(_StreamInfo)]
    lib.isst_kv_evict.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.isst_stream_import_llm_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.isst_stream_import_enc_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.isst_stream_import_audio_history.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.isst_debug_beam_trace_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    lib.isst_debug_beam_trace_step.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.isst_debug_beam_trace_end.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.isst_op_attn_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_gemm_attn_merge.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_splice_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    lib.isst_op_embed_splice.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_enc_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_llm_attention.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.isst_generate.argtypes = [C.c_void_p, C.POINTER(_GenParams), C.c_int, C.POINTER(C.c_int),
                                  C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.POINTER(C.c_int),
                                  C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_void_p, C.c_void_p]
    lib.isst_encode_speech.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                       C.POINTER(C.c_int), C.c_void_p]
    lib.isst_debug_tap.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.isst_debug_read_kv.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.isst_profile_begin.argtypes = [C.c_void_p]
    lib.isst_profile_begin_rows.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.isst_profile_end.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.isst_op_pack_weight.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_packed_elems.argtypes = [C.c_int, C.c_int]
    lib.isst_op_pack_gateup8.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_gemm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p]
    lib.isst_op_gemm_splitk_rmsnorm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.isst_op_gemm_splitk_layernorm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.isst_op_set_gemm_tuning.argtypes = [C.c_int, C.c_int]
    lib.isst_op_set_attn_tuning.argtypes = [C.c_int]
    lib.isst_op_set_reduce_tuning.argtypes = [C.c_int, C.c_int]
    lib.isst_op_layernorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                      C.c_int, C.c_void_p]
    lib.isst_op_rmsnorm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]
    lib.isst_op_conv0.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.isst_op_warp_sample.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_double, C.POINTER(C.c_int)]
    lib.isst_op_topk_rows.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.isst_op_sample_uniform.argtypes = [C.c_ulonglong, C.c_int, C.c_int, C.c_int]
    lib.isst_op_sample_uniform.restype = C.c_double
    lib.isst_op_sample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                   C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    if path is None:
        _lib = lib
    return lib


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def make_c_config(cfg: ModelConfig, max_streams: int, max_multiplier: int, max_prompt_len: int, max_new_tokens: int,
                  max_llm_cache_size: int, max_system_prompt: int, debug_taps: bool, max_beams: int = 1) -> _Config:
    c = _Config()
    c.n_conv = len(cfg.conv_layers)
    for i, (d, k, s) in enumerate(cfg.conv_layers):
        c.conv_dim[i], c.conv_k[i], c.conv_stride[i] = d, k, s
    c.conv_bias = int(cfg.conv_bias)
    c.enc_dim, c.enc_layers, c.enc_heads, c.enc_ffn = cfg.enc_dim, cfg.enc_layers, cfg.enc_heads, cfg.enc_ffn
    c.enc_ln_eps, c.block_size, c.max_cache_size = cfg.enc_ln_eps, cfg.block_size, cfg.max_cache_size
    c.enc_rope_round_each = int(cfg.enc_rope_mode == "bf16")
    c.n_shrink = len(cfg.shrink_layers)
    for i, (d, k, s) in enumerate(cfg.shrink_layers):
        c.shrink_dim[i], c.shrink_k[i], c.shrink_stride[i] = d, k, s
    if cfg.llm_head_dim != 128:
        raise IsstError("llm_head_dim must be 128")
    c.llm_dim, c.llm_layers, c.llm_heads, c.llm_kv_heads = cfg.llm_dim, cfg.llm_layers, cfg.llm_heads, cfg.llm_kv_heads
    c.llm_ffn, c.vocab, c.rms_eps = cfg.llm_ffn, cfg.vocab, cfg.rms_eps
    c.user_id, c.assistant_id, c.start_header_id = cfg.user_id, cfg.assistant_id, cfg.start_header_id
    c.n_eos = len(cfg.eos_ids)
    for i, e in enumerate(cfg.eos_ids):
        c.eos_ids[i] = e
    c.max_streams, c.max_multiplier, c.max_prompt_len = max_streams, max_multiplier, max_prompt_len
    c.max_new_tokens, c.max_llm_cache_size, c.max_system_prompt = max_new_tokens, max_llm_cache_size, max_system_prompt
    c.debug_taps = int(debug_taps)
    c.max_beams = max_beams
    c.enc_abs_pos = int(not cfg.enc_rope)
    return c


def _pack_int32(seqs, n: int):
    """n int32 sequences (lists, arrays or None) -> (owner of the memory, c_void_p[n] pointing into it or NULL, c_int[n] lengths)."""
    if len(seqs) != n:
        raise IsstError(f"{len(seqs)} sequences for {n} streams")
    lens = [0 if s is None else len(s) for s in seqs]
    total = sum(lens)
    if total == 0:
        return None, (C.c_void_p * n)(), (C.c_int * n)(*lens)
    live = [s for s, l in zip(seqs, lens) if l]
    if all(isinstance(s, np.ndarray) for s in live):
        flat = np.concatenate(live).astype(np.int32, copy=False)
    else:
        flat = np.fromiter(itertools.chain.from_iterable(live), dtype=np.int32, count=total)
    flat = np.ascontiguousarray(flat)
    base = flat.ctypes.data
    ptrs, off = [], 0
    for l in lens:
        ptrs.append(base + 4 * off if l else None)
        off += l
    return flat, (C.c_void_p * n)(*ptrs), (C.c_int * n)(*lens)


class Engine:
    """Owns one library handle: device weights + per-stream KV state on the current CUDA(HIP) device."""

    def __init__(self, cfg: ModelConfig, max_streams: int = 1, max_multiplier: int = 1, max_prompt_len: int = 128,
                 max_new_tokens: int = 40, max_llm_cache_size: int = 1000, max_system_prompt: int = 128,
                 debug_taps: bool = False, max_beams: int = 1):
        if not torch.cuda.is_available():
            raise IsstError("no GPU visible: the InfiniSST hot path has no CPU implementation in this package")
        self.lib = load_library()
        self.cfg = cfg
        self.c_cfg = make_c_config(cfg, max_streams, max_multiplier, max_prompt_len, max_new_tokens,
                                   max_llm_cache_size, max_system_prompt, debug_taps, max_beams)
        self.max_new_tokens = max_new_tokens
        self.last_call_seconds = 0.0
        r64 = lambda x: (x + 63) // 64 * 64
        # rows the library wants from isst_set_rope_tables (engine_core.hip isst_create: sys_cap + ring_cap, enc_cap)
        self._llm_rope_rows = r64(max_system_prompt) + r64(max_llm_cache_size + max_prompt_len + max_new_tokens + 8)
        self._enc_rope_rows = r64(cfg.max_cache_size + cfg.block_size * max_multiplier)
        self.h = C.c_void_p()
        rc = self.lib.isst_create(C.byref(self.c_cfg), C.byref(self.h))
        if rc != 0:
            raise IsstError(f"isst_create failed ({rc}): {self.lib.isst_last_error(None).decode()}")

    # ---------------------------------------------------------------- errors / lifetime
    def _check(self, rc: int, what: str):
        if rc != 0:
            raise IsstError(f"{what} failed ({rc}): {self.lib.isst_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.isst_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- weights
    def load_checkpoint(self, path: str):
        """Reference `pytorch_model.bin` (agents/infinisst.py:179-180) -> device weights; returns the skipped keys."""
        from .checkpoint import load_checkpoint
        weights, inv_freq, skipped = load_checkpoint(self.cfg, path)
        self.load_weights(weights, enc_inv_freq=inv_freq)
        return skipped

    def load_weights(self, weights: Dict[str, torch.Tensor], strict: bool = True, enc_inv_freq=None):
        """`weights`: reference checkpoint keys -> bf16 tensors (CPU or GPU).  Unknown keys raise if strict."""
        for name, t in weights.items():
            if t.dtype != torch.bfloat16:
                raise IsstError(f"{name}: expected bfloat16, got {t.dtype}")
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            rc = self.lib.isst_load_weight(self.h, name.encode(), C.c_void_p(t.data_ptr()), t.dim(), shape, int(t.is_cuda))
            if rc == -5 and not strict:
                continue
            self._check(rc, f"isst_load_weight({name})")
        rows_e, rows_l = max(1024, self._enc_rope_rows), max(1 << 14, self._llm_rope_rows)
        ec, es = rope.encoder_tables(self.cfg, rows_e, enc_inv_freq)
        lc, ls = rope.llm_tables(self.cfg, rows_l)
        self._check(self.lib.isst_set_rope_tables(self.h, C.c_void_p(ec.data_ptr()), C.c_void_p(es.data_ptr()), rows_e,
                                                  C.c_void_p(lc.data_ptr()), C.c_void_p(ls.data_ptr()), rows_l),
                    "isst_set_rope_tables")
        if not self.cfg.enc_rope:
            table = rope.encoder_position_table(self.cfg)
            self._check(self.lib.isst_set_enc_position_table(self.h, C.c_void_p(table.data_ptr()), table.shape[0]), "isst_set_enc_position_table")
        self._check(self.lib.isst_finalize_weights(self.h), "isst_finalize_weights")
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- streams
    def open_stream(self) -> int:
        sid = C.c_int(-1)
        self._check(self.lib.isst_stream_open(self.h, C.byref(sid)), "isst_stream_open")
        return sid.value

    def reset_stream(self, sid: int):
        self._check(self.lib.isst_stream_reset(self.h, sid), "isst_stream_reset")

    def close_stream(self, sid: int):
        self._check(self.lib.isst_stream_close(self.h, sid), "isst_stream_close")

    def stream_info(self, sid: int) -> dict:
        info = _StreamInfo()
        self._check(self.lib.isst_stream_info_get(self.h, sid, C.byref(info)), "isst_stream_info_get")
        return {k: getattr(info, k) for k, _ in _StreamInfo._fields_}

    def stream_cache_lens(self, sids: Sequence[int]) -> list:
        """llm_cache_len of several streams (what the agent reads as past_key_values[0][0].size(2), agents/infinisst.py:337)."""
        info = _StreamInfo()
        ref = C.byref(info)
        out = []
        for sid in sids:
            self._check(self.lib.isst_stream_info_get(self.h, sid, ref), "isst_stream_info_get")
            out.append(info.llm_cache_len)
        return out

    def kv_evict(self, sid: int, new_cache_size: int, keep_prefix: int):
        self._check(self.lib.isst_kv_evict(self.h, sid, new_cache_size, keep_prefix), "isst_kv_evict")

    # ---------------------------------------------------------------- state import (resume / steady-state set-up)
    def import_llm_kv(self, sid: int, kv, sys_len: int, ring_start: int = 0):
        """`kv`: past_key_values as a list over layers of (K, V), each (1, kv_heads, T, 128) bf16 with UNROTATED keys
        (the reference's patched cache, model/patches/patch_llm.py:280-284); the first `sys_len` entries become the pinned prefix."""
        if len(kv) != self.cfg.llm_layers:
            raise IsstError(f"{len(kv)} layers given, the model has {self.cfg.llm_layers}")
        for layer, (k, v) in enumerate(kv):
            k = k.detach().to("cpu", torch.bfloat16).reshape(self.cfg.llm_kv_heads, -1, 128).contiguous()
            v = v.detach().to("cpu", torch.bfloat16).reshape(self.cfg.llm_kv_heads, -1, 128).contiguous()
            self._check(self.lib.isst_stream_import_llm_kv(self.h, sid, layer, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), k.shape[1],
                                                           sys_len, ring_start), "isst_stream_import_llm_kv")

    def import_speech_cache(self, sid: int, layers, n_steps: int, audio_tail=None, ring_start: int = 0):
        """`layers`: per encoder layer (K, V), each (heads, T, 64) bf16 with unrotated keys (= speech_cache.layers[i].k / .v, reference
        model/speech_encoder.py:80-97); `n_steps` = speech_cache.n_steps; `audio_tail`: the last 399 samples of speech_cache.src."""
        if len(layers) != self.cfg.enc_layers:
            raise IsstError(f"{len(layers)} layers given, the encoder has {self.cfg.enc_layers}")
        for layer, (k, v) in enumerate(layers):
            k = k.detach().to("cpu", torch.bfloat16).reshape(self.cfg.enc_heads, -1, 64).contiguous()
            v = v.detach().to("cpu", torch.bfloat16).reshape(self.cfg.enc_heads, -1, 64).contiguous()
            self._check(self.lib.isst_stream_import_enc_kv(self.h, sid, layer, C.c_void_p(k.data_ptr()), C.c_void_p(v.data_ptr()), k.shape[1],
                                                           n_steps, ring_start), "isst_stream_import_enc_kv")
        if audio_tail is not None:
            a = torch.as_tensor(audio_tail).detach().to("cpu", torch.bfloat16).reshape(-1).contiguous()
            self._check(self.lib.isst_stream_import_audio_history(self.h, sid, C.c_void_p(a.data_ptr()), a.numel()), "isst_stream_import_audio_history")

    # ---------------------------------------------------------------- beam-search test aid
    def beam_trace_begin(self, num_beams: int, forced_tokens=None, forced_parents=None):
        ft = np.ascontiguousarray(np.asarray(forced_tokens if forced_tokens is not None else [], dtype=np.int32).reshape(-1))
        fp = np.ascontiguousarray(np.asarray(forced_parents if forced_parents is not None else [], dtype=np.int32).reshape(-1))
        steps = ft.size // num_beams
        self._check(self.lib.isst_debug_beam_trace_begin(self.h, num_beams, ft.ctypes.data if steps else None, fp.ctypes.data if steps else None, steps),
                    "isst_debug_beam_trace_begin")

    def beam_trace_end(self):
        """-> list over steps of (top_val [rows][n_keep], top_idx [rows][n_keep], beam_scores [rows])."""
        n = C.c_int(0)
        out = []
        step = 0
        while True:
            rows, keep = C.c_int(0), C.c_int(0)
            if self.lib.isst_debug_beam_trace_step(self.h, step, C.byref(rows), C.byref(keep), None, None, None, 0) != 0:
                break
            val = np.zeros((rows.value, keep.value), dtype=np.float32)
            idx = np.zeros((rows.value, keep.value), dtype=np.int32)
            sc = np.zeros(rows.value, dtype=np.float32)
            self._check(self.lib.isst_debug_beam_trace_step(self.h, step, C.byref(rows), C.byref(keep), val.ctypes.data, idx.ctypes.data,
                                                            sc.ctypes.data, val.size), "isst_debug_beam_trace_step")
            out.append((val, idx, sc))
            step += 1
        self._check(self.lib.isst_debug_beam_trace_end(self.h, C.byref(n)), "isst_debug_beam_trace_end")
        return out

    # ---------------------------------------------------------------- hot path
    def generate(self, gen: GenConfig, stream_ids: Sequence[int], pcm: Sequence[np.ndarray],
                 prompt_ids: Sequence[Sequence[int]], prev_target_ids: Sequence[Sequence[int]],
                 system_prompt_size: int = 0, forced_tokens: Optional[Sequence[Optional[Sequence[int]]]] = None,
                 return_logits: bool = False):
        """model.generate(...) of reference agents/infinisst.py:307-332 for len(stream_ids) streams.
        `pcm`: one fp32 waveform segment per stream -- numpy arrays (host; the library uploads them, reference :222) or, all of them,
        contiguous fp32 CUDA tensors (audio already resident in HBM: isst_gen_params.pcm_on_device).
        Returns (list of generated id lists, logits or None)."""
        n = len(stream_ids)
        on_device = len(pcm) > 0 and all(isinstance(x, torch.Tensor) and x.is_cuda for x in pcm)
        if on_device:
            if any(x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 1 for x in pcm):
                raise IsstError("device audio must be contiguous 1-D fp32 tensors")
            pcm = list(pcm)
        else:
            pcm = [np.ascontiguousarray(x, dtype=np.float32) for x in pcm]
        n_samples = int(pcm[0].shape[0])
        if any(tuple(x.shape) != (n_samples,) for x in pcm):
            raise IsstError("all streams of one call must bring the same number of samples")
        p = _GenParams()
        p.pcm_on_device = int(on_device)
        p.multiplier, p.max_new_tokens = gen.latency_multiplier, gen.max_new_tokens
        p.no_repeat_ngram_size = p.encoder_no_repeat_ngram_size = gen.no_repeat_ngram_size
        p.repetition_penalty = gen.repetition_penalty
        sup = np.asarray(gen.suppress_tokens, dtype=np.int32)
        p.suppress_tokens = sup.ctypes.data_as(C.POINTER(C.c_int)) if sup.size else None
        p.n_suppress = int(sup.size)
        p.system_prompt_size = system_prompt_size
        p.num_beams = gen.beam
        p.length_penalty = 1.0
        p.do_sample, p.temperature, p.top_k = int(gen.do_sample), float(gen.temperature), int(gen.top_k)
        p.top_p, p.epsilon_cutoff, p.seed = float(gen.top_p), float(gen.epsilon_cutoff), int(gen.seed)
        sid_arr = (C.c_int * n)(*stream_ids)
        # every int32 sequence of the call lives in ONE buffer per argument, the per-stream pointers are offsets into it (64 streams: the
        # marshalling is host time during which the GPU idles between two steps)
        keep_p, prompt_ptrs, prompt_lens = _pack_int32(prompt_ids, n)
        keep_v, prev_ptrs, prev_lens = _pack_int32(prev_target_ids, n)
        keep_f, forced_ptrs, forced_lens = _pack_int32([None] * n if forced_tokens is None else forced_tokens, n)
        outs = np.zeros((n, max(1, gen.max_new_tokens)), dtype=np.int32)
        out_base, out_stride = outs.ctypes.data, outs.strides[0]
        out_ptrs = (C.c_void_p * n)(*[out_base + i * out_stride for i in range(n)])
        out_lens = (C.c_int * n)()
        logits = np.zeros((n, gen.max_new_tokens, self.cfg.vocab), dtype=np.float32) if return_logits else None
        if on_device:
            pcm_ptrs = (C.c_void_p * n)(*[x.data_ptr() for x in pcm])
        else:
            pcm_ptrs = (C.c_void_p * n)(*[x.ctypes.data for x in pcm])
        args = (self.h, C.byref(p), n, sid_arr, pcm_ptrs, n_samples, prompt_ptrs, prompt_lens, prev_ptrs, prev_lens, forced_ptrs, forced_lens,
                out_ptrs, out_lens, None if logits is None else logits.ctypes.data, _stream_ptr())
        t0 = time.perf_counter()
        rc = self.lib.isst_generate(*args)
        self.last_call_seconds = time.perf_counter() - t0  # wall time inside the library (it returns once the last token id is on the host)
        self._check(rc, "isst_generate")
        del keep_p, keep_v, keep_f
        rows = outs.tolist()
        return [rows[i][: out_lens[i]] for i in range(n)], logits

    def encode_speech(self, sid: int, pcm: np.ndarray, multiplier: int = 1) -> torch.Tensor:
        """speech_encoder.encode_speech for one stream -> (S, llm_dim) bf16 CPU tensor (test aid)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        S_max = pcm.shape[0] // (self.cfg.samples_per_frame * self.cfg.shrink_factor)
        out = torch.empty((S_max, self.cfg.llm_dim), dtype=torch.bfloat16)
        rows = C.c_int(0)
        self._check(self.lib.isst_encode_speech(self.h, sid, pcm.ctypes.data, pcm.shape[0], multiplier, out.data_ptr(),
                                                C.byref(rows), _stream_ptr()), "isst_encode_speech")
        return out[: rows.value]

    def read_kv(self, sid: int, pos: int, layer: int = 0, kv_head: int = 0, beam: int = 0):
        """(K, V) rows (128 bf16 each, unrotated) of one cached position (test aid)."""
        k = torch.empty(128, dtype=torch.bfloat16)
        v = torch.empty(128, dtype=torch.bfloat16)
        self._check(self.lib.isst_debug_read_kv(self.h, sid, beam, layer, kv_head, pos, k.data_ptr(), v.data_ptr()), "isst_debug_read_kv")
        return k, v

    def profile_begin(self, rows_lo: int = 1, rows_hi: int = 1):
        """Bracket the gate/up launch of every Llama pass of rows_lo..rows_hi rows (default: the one-token decode GEMV)."""
        self._check(self.lib.isst_profile_begin_rows(self.h, rows_lo, rows_hi), "isst_profile_begin_rows")

    def profile_end(self):
        """(average event bracket in us around the profiled gate/up launches, launches timed) since profile_begin."""
        avg, n = C.c_double(0.0), C.c_int64(0)
        self._check(self.lib.isst_profile_end(self.h, _stream_ptr(), C.byref(avg), C.byref(n)), "isst_profile_end")
        return avg.value, n.value

    def debug_tap(self, name: str) -> torch.Tensor:
        got = C.c_int64(0)
        self._check(self.lib.isst_debug_tap(self.h, name.encode(), None, 0, C.byref(got)), f"isst_debug_tap({name})")
        out = torch.empty(got.value, dtype=torch.bfloat16)
        self._check(self.lib.isst_debug_tap(self.h, name.encode(), out.data_ptr(), got.value, C.byref(got)),
                    "isst_debug_tap")
        return out


# -------------------------------------------------------------------- per-kernel wrappers (device tensors)
def op_pack_weight(w: torch.Tensor, conv_k: int = 0) -> torch.Tensor:
    lib = load_library()
    n_rows = w.shape[0]
    K = w.numel() // n_rows
    out = torch.zeros(lib.isst_op_packed_elems(n_rows, K), dtype=torch.bfloat16, device=w.device)
    rc = lib.isst_op_pack_weight(_ptr(w.contiguous()), _ptr(out), n_rows, K, conv_k, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_pack_weight -> {rc}")
    return out


def op_pack_gateup8(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """gate_proj / up_proj (ffn, K) as self-paired tiles for op_gemm(..., N=2 * ffn, epi="swiglu8")."""
    lib = load_library()
    ffn, K = gate.shape
    out = torch.zeros(2 * ffn * K, dtype=torch.bfloat16, device=gate.device)
    rc = lib.isst_op_pack_gateup8(_ptr(gate.contiguous()), _ptr(up.contiguous()), _ptr(out), ffn, K, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_pack_gateup8 -> {rc}")
    return out


def op_gemm(A: torch.Tensor, packed: torch.Tensor, N: int, epi: str = "none", bias=None, res=None, n_valid=None,
            lda: Optional[int] = None, M: Optional[int] = None, K: Optional[int] = None, norm_w=None,
            norm_eps: float = 1e-5) -> torch.Tensor:
    lib = load_library()
    M = A.shape[0] if M is None else M
    K = A.shape[1] if K is None else K
    lda = A.stride(0) if lda is None else lda
    n_out = (N // 2 if epi in ("swiglu", "swiglu8") else N) if n_valid is None else n_valid
    out = torch.zeros((M, n_out), dtype=torch.float32 if epi == "f32" else torch.bfloat16, device=A.device)
    rc = lib.isst_op_gemm(_ptr(A), lda, _ptr(packed), _ptr(bias), _ptr(res), 0 if res is None else res.stride(0),
                          _ptr(out), out.stride(0), M, N, K, n_out, EPI[epi], _ptr(norm_w), norm_eps, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm -> {rc}")
    return out


def dense_kernel_name() -> str:
    """The kernel isst_op_gemm dispatches to above 64 rows (bench.py's `mfma` block names it)."""
    return "gemm_dense_kernel<EPI> (256 x 256 tile, 8-wave ping-pong, LDS-DMA; gemm_dense.hip)"


def op_gemm_splitk_rmsnorm(A: torch.Tensor, packed: torch.Tensor, x: torch.Tensor, ksplit: int, norm_w=None, norm_eps: float = 1e-5):
    """x <- bf16(x + bf16(A @ W^T)) through `ksplit` fp32 K-slabs; returns (x_new, RMSNorm(x_new) or None).  17..64 rows."""
    lib = load_library()
    M, K = A.shape
    N = x.shape[1]
    x = x.clone()
    out = torch.empty_like(x) if norm_w is not None else None
    slabs = torch.empty((ksplit, M, N), dtype=torch.float32, device=A.device)
    rc = lib.isst_op_gemm_splitk_rmsnorm(_ptr(A), A.stride(0), _ptr(packed), _ptr(x), _ptr(norm_w), _ptr(out), _ptr(slabs), M, N, K,
                                         ksplit, norm_eps, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm_splitk_rmsnorm -> {rc}")
    return x, out


def op_gemm_splitk_fused(A: torch.Tensor, packed: torch.Tensor, x: torch.Tensor, ksplit: int, tickets: Optional[torch.Tensor] = None):
    """The launch-free form of op_gemm_splitk_rmsnorm's first half: returns (x_new, ssq [M][N/32] fp32, tickets) -- 13..64 rows."""
    lib = load_library()
    M, K = A.shape
    N = x.shape[1]
    x = x.clone()
    slabs = torch.empty((ksplit, M, N), dtype=torch.float32, device=A.device)
    ssq = torch.zeros((M, N // 32), dtype=torch.float32, device=A.device)
    if tickets is None:
        tickets = torch.zeros(N // 16, dtype=torch.int32, device=A.device)
    lib.isst_op_gemm_splitk_fused.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    rc = lib.isst_op_gemm_splitk_fused(_ptr(A), A.stride(0), _ptr(packed), _ptr(x), _ptr(slabs), _ptr(ssq), _ptr(tickets), M, N, K, ksplit, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm_splitk_fused -> {rc}")
    return x, ssq, tickets


def op_gemm_splitk_plain(A: torch.Tensor, packed: torch.Tensor, N: int, ksplit: int, norm_w=None, ssq=None, norm_eps: float = 1e-5):
    """bf16(A @ W^T) through `ksplit` K slices reduced inside the launch (no residual); norm_w + ssq: A rows RMS-normalised while staged."""
    lib = load_library()
    M, K = A.shape
    out = torch.zeros((M, N), dtype=torch.bfloat16, device=A.device)
    slabs = torch.empty((ksplit, M, N), dtype=torch.float32, device=A.device)
    tickets = torch.zeros(N // 16, dtype=torch.int32, device=A.device)
    lib.isst_op_gemm_splitk_plain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    rc = lib.isst_op_gemm_splitk_plain(_ptr(A), A.stride(0), _ptr(packed), _ptr(out), out.stride(0), _ptr(slabs), _ptr(tickets), M, N, K, ksplit,
                                       _ptr(norm_w), norm_eps, _ptr(ssq), _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm_splitk_plain -> {rc}")
    assert int(tickets.abs().sum()) == 0, "arrival counters not re-armed"
    return out


def op_gemm_norm_ssq(x: torch.Tensor, packed: torch.Tensor, N: int, norm_w: torch.Tensor, ssq: torch.Tensor, epi: str = "none", norm_eps: float = 1e-5):
    """epi(RMSNorm(x) @ W^T) with the rows normalised while they are staged (1/rms from the producer's sums of squares) -- 13..64 rows."""
    lib = load_library()
    M, K = x.shape
    n_out = N // 2 if epi == "swiglu" else N
    out = torch.zeros((M, n_out), dtype=torch.float32 if epi == "f32" else torch.bfloat16, device=x.device)
    lib.isst_op_gemm_norm_ssq.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
    rc = lib.isst_op_gemm_norm_ssq(_ptr(x), x.stride(0), _ptr(packed), _ptr(out), out.stride(0), M, N, K, n_out, EPI[epi], _ptr(norm_w), norm_eps, _ptr(ssq), _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm_norm_ssq -> {rc}")
    return out


def op_gemm_splitk_layernorm(A, packed, bias, x, ksplit, ln_w, ln_b, eps=1e-5):
    """x <- bf16(x + bf16(A @ W^T + bias)) through `ksplit` fp32 K-slabs; returns (x_new, LayerNorm(x_new))."""
    lib = load_library()
    M, K = A.shape
    N = x.shape[1]
    x = x.clone()
    out = torch.empty_like(x)
    slabs = torch.empty((ksplit, M, N), dtype=torch.float32, device=A.device)
    rc = lib.isst_op_gemm_splitk_layernorm(_ptr(A), A.stride(0), _ptr(packed), _ptr(bias), _ptr(x), _ptr(ln_w), _ptr(ln_b), _ptr(out), _ptr(slabs),
                                           M, N, K, ksplit, eps, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_gemm_splitk_layernorm -> {rc}")
    return x, out


def op_layernorm(x, w, b, eps=1e-5, gelu=False):
    lib = load_library()
    out = torch.empty_like(x)
    rc = lib.isst_op_layernorm(_ptr(x), _ptr(w), _ptr(b), _ptr(out), x.shape[0], x.shape[1], eps, int(gelu), _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_layernorm -> {rc}")
    return out


def op_rmsnorm(x, w, eps=1e-5):
    lib = load_library()
    out = torch.empty_like(x)
    rc = lib.isst_op_rmsnorm(_ptr(x), _ptr(w), _ptr(out), x.shape[0], x.shape[1], eps, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_rmsnorm -> {rc}")
    return out


def op_conv0(audio, w, bias, ln_w, ln_b, k, stride):
    lib = load_library()
    Cc = w.shape[0]
    T = (audio.shape[0] - k) // stride + 1
    out = torch.empty((T, Cc), dtype=torch.bfloat16, device=audio.device)
    rc = lib.isst_op_conv0(_ptr(audio), _ptr(w.contiguous()), _ptr(bias), _ptr(ln_w), _ptr(ln_b), _ptr(out), T, Cc, k,
                           stride, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_conv0 -> {rc}")
    return out


def op_sample(logits: torch.Tensor, ids, enc_ids, suppress, penalty, ngram, enc_ngram) -> int:
    lib = load_library()
    dev = logits.device

    def mk(x):
        return torch.tensor(list(x) if len(x) else [0], dtype=torch.int32, device=dev)

    ids_t, enc_t, sup_t = mk(ids), mk(enc_ids), mk(suppress)
    out = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = lib.isst_op_sample(_ptr(logits), logits.numel(), _ptr(ids_t), len(ids), _ptr(enc_t), len(enc_ids), _ptr(sup_t),
                            len(suppress), penalty, ngram, enc_ngram, _ptr(out), _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_sample -> {rc}")
    return int(out.item())


def op_warp_sample(scores: np.ndarray, temperature: float, top_k: int, top_p: float, epsilon_cutoff: float, u: float):
    """HF's warpers Temperature -> TopK -> TopP -> Epsilon on one row of processed scores + the inverse-CDF draw (host code: no GPU needed).
    Returns (warped scores, token)."""
    lib = load_library()
    s = np.ascontiguousarray(scores, dtype=np.float32).copy()
    tok = C.c_int(-1)
    rc = lib.isst_op_warp_sample(s.ctypes.data, s.size, temperature, top_k, top_p, epsilon_cutoff, u, C.byref(tok))
    if rc:
        raise IsstError(f"isst_op_warp_sample -> {rc}")
    return s, tok.value


def op_warp(scores: np.ndarray, temperature: float, top_k: int, top_p: float, epsilon_cutoff: float, min_tokens_to_keep: int = 1) -> np.ndarray:
    """HF's warpers on one row of processed scores (host code: csrc/warp.hip); returns the warped copy."""
    lib = load_library()
    s = np.ascontiguousarray(scores, dtype=np.float32).copy()
    lib.isst_op_warp.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_float, C.c_float, C.c_int]
    rc = lib.isst_op_warp(s.ctypes.data, s.size, temperature, top_k, top_p, epsilon_cutoff, min_tokens_to_keep)
    if rc:
        raise IsstError(f"isst_op_warp -> {rc}")
    return s


def op_multinomial_wor(scores: np.ndarray, k: int, uniforms: Sequence[float]) -> list:
    """`torch.multinomial(softmax(scores), k)` without replacement as k sequential inverse-CDF draws at the given uniforms (host code: csrc/warp.hip)."""
    lib = load_library()
    s = np.ascontiguousarray(scores, dtype=np.float32)
    u = np.ascontiguousarray(uniforms, dtype=np.float64)
    out = np.zeros(k, dtype=np.int64)
    lib.isst_op_multinomial_wor.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_void_p]
    rc = lib.isst_op_multinomial_wor(s.ctypes.data, s.size, k, u.ctypes.data, out.ctypes.data)
    if rc:
        raise IsstError(f"isst_op_multinomial_wor -> {rc}")
    return out.tolist()


def op_sample_uniform(seed: int, stream: int, chunk: int, step: int) -> float:
    return float(load_library().isst_op_sample_uniform(seed, stream, chunk, step))


def op_splice_map(ids: Sequence[int], user_id: int, assistant_id: int, start_header_id: int, n_features: int) -> list:
    """Row map of the speech splice (host logic of the library): entry >= 0 = prompt token index, < 0 = speech feature -1 - entry."""
    lib = load_library()
    a = np.ascontiguousarray(np.asarray(ids, dtype=np.int32))
    out = np.zeros(a.size, dtype=np.int32)
    n = C.c_int(0)
    rc = lib.isst_op_splice_map(a.ctypes.data, a.size, user_id, assistant_id, start_header_id, n_features, out.ctypes.data, C.byref(n))
    if rc:
        raise IsstError(f"isst_op_splice_map -> {rc}")
    return out[: n.value].tolist()


def op_embed_splice(ids: torch.Tensor, speech_row: Optional[torch.Tensor], table: torch.Tensor, speech: Optional[torch.Tensor]) -> torch.Tensor:
    """out[r] = speech[speech_row[r]] if speech_row[r] >= 0 else table[ids[r]] (int32 device tensors, bf16 tables)."""
    lib = load_library()
    rows, D = ids.numel(), table.shape[1]
    out = torch.empty((rows, D), dtype=torch.bfloat16, device=table.device)
    rc = lib.isst_op_embed_splice(_ptr(ids), _ptr(speech_row), _ptr(table), _ptr(speech), _ptr(out), rows, D, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_embed_splice -> {rc}")
    return out


def op_enc_attention(qkv: torch.Tensor, kring: torch.Tensor, vring: torch.Tensor, ring_start: int, prefix: int, rope_cos: torch.Tensor,
                     rope_sin: torch.Tensor, round_each: bool, heads: int, max_cache: int, blocksize: int) -> torch.Tensor:
    """qkv (Q, 3*heads*64) bf16; kring (heads, cap, 64), vring (heads, 64, cap) are updated in place; -> (Q, heads*64)."""
    lib = load_library()
    Q, cap = qkv.shape[0], kring.shape[1]
    out = torch.empty((Q, heads * 64), dtype=torch.bfloat16, device=qkv.device)
    rc = lib.isst_op_enc_attention(_ptr(qkv), _ptr(kring), _ptr(vring), ring_start, prefix, _ptr(rope_cos), _ptr(rope_sin), int(round_each), _ptr(out),
                                   Q, heads, cap, max_cache, blocksize, _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_enc_attention -> {rc}")
    return out


def op_llm_attention(qkv: torch.Tensor, pos0: int, kpool: torch.Tensor, krpool: torch.Tensor, vpool: torch.Tensor, heads: int, kv_heads: int,
                     sys_cap: int, ring_cap: int, sys_len: int, ring_start: int, rope_cos: torch.Tensor, rope_sin: torch.Tensor,
                     rot_keys: bool = True) -> torch.Tensor:
    """qkv (rows, (heads + 2 kv_heads) * 128) bf16 at positions pos0..; pools (kv_heads, sys_cap + ring_cap, 128) updated in place;
    rope tables bf16 (>= pos0 + rows, 64); -> (rows, heads * 128)."""
    lib = load_library()
    rows = qkv.shape[0]
    out = torch.empty((rows, heads * 128), dtype=torch.bfloat16, device=qkv.device)
    rc = lib.isst_op_llm_attention(_ptr(qkv), rows, pos0, _ptr(kpool), _ptr(krpool), _ptr(vpool), heads, kv_heads, sys_cap, ring_cap, sys_len, ring_start,
                                   _ptr(rope_cos), _ptr(rope_sin), int(rot_keys), _ptr(out), _stream_ptr())
    if rc:
        raise IsstError(f"isst_op_llm_attention -> {rc}")
    return out
