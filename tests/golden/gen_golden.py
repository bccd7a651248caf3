"""Generate golden input/output vectors by running the REFERENCE's own functions (this container only).

    python tests/golden/gen_golden.py        # needs /root/reference ; writes tests/golden/*.npz

The reference cannot be imported as a whole here (fairseq, simuleval, lightning, rotary_embedding_torch,
wandb are absent; transformers is 5.15 instead of the pinned 4.47) -- SURVEY.md section 8(c) / Appendix A.
This script registers stub modules for those packages in `sys.modules`, imports the reference modules from
/root/reference unchanged, binds the reference's functions onto toy modules and records what they compute.
Nothing from the reference is copied: the fixtures hold only tensors (inputs, weights, outputs) and scalars.

Third-party behaviour that has to be *restated* by a stub (and is therefore NOT pinned by these fixtures):
rotary_embedding_torch.RotaryEmbedding (stub below), fairseq's ConvFeatureExtractionModel (toy stand-in).
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..", "..")))


def _mod(name, **attrs):
    import importlib.machinery
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


# ------------------------------------------------------------------ stubs: fairseq + rotary -----
class _TransposeLast(nn.Module):
    def forward(self, x):
        return x.transpose(-2, -1)


class _StubRotary(nn.Module):
    """Stand-in for rotary_embedding_torch.RotaryEmbedding(dim, use_xpos=False): fp32 tables, interleaved pairs,
    queries at offset K-Q.  (Restated third-party behaviour: NOT a pin of that package.)"""

    def __init__(self, dim, use_xpos=False, theta=10000):
        super().__init__()
        self.dim = dim
        self.register_buffer("freqs", 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)),
                             persistent=False)

    def _rot(self, t, offset):
        n = t.shape[-2]
        pos = torch.arange(n, dtype=torch.float32) + offset
        ang = (pos.unsqueeze(1) * self.freqs.float().unsqueeze(0)).repeat_interleave(2, dim=-1)
        t2 = t.reshape(*t.shape[:-1], -1, 2)
        rot = torch.stack((-t2[..., 1], t2[..., 0]), dim=-1).reshape(t.shape)
        return (t.float() * ang.cos() + rot.float() * ang.sin()).to(t.dtype)

    def rotate_queries_with_cached_keys(self, q, k):
        return self._rot(q, k.shape[-2] - q.shape[-2]), self._rot(k, 0)


def install_fairseq_stubs():
    def pad_to_multiple(x, multiple, dim=-1, value=0):
        if x is None:
            return None, 0
        tsz = x.size(dim)
        m = tsz / multiple
        remainder = int(np.ceil(m)) * multiple - tsz
        if m == int(m):
            return x, 0
        pad_offset = (0,) * (-1 - dim) * 2
        return F.pad(x, (*pad_offset, 0, remainder), value=value), remainder

    class MultiheadAttention(nn.Module):
        @staticmethod
        def _append_prev_key_padding_mask(**kw):
            return None

        def reset_parameters(self):
            pass

        def apply_sparse_mask(self, attn_weights, tgt_len, src_len, bsz):
            return attn_weights

    class _Empty(nn.Module):
        pass

    utils = _mod("fairseq.utils", index_put=lambda t, m, v: t.masked_fill(m, v), is_xla_tensor=lambda t: False,
                 softmax=lambda x, dim, onnx_trace=False: F.softmax(x.float(), dim=dim),
                 eval_str_dict=lambda x, type=dict: None if x is None else eval(x))
    fs = _mod("fairseq", utils=utils)
    W2V = type("Wav2Vec2Model", (nn.Module,), {})
    _mod("fairseq.models")
    _mod("fairseq.models.wav2vec", TransformerEncoder=type("TransformerEncoder", (nn.Module,), {}),
         TransformerSentenceEncoderLayer=type("TransformerSentenceEncoderLayer", (nn.Module,), {}),
         Wav2Vec2Model=W2V, Wav2VecEncoder=type("Wav2VecEncoder", (nn.Module,), {}))
    _mod("fairseq.models.wav2vec.wav2vec2", Wav2Vec2Model=W2V)
    _mod("fairseq.models.wav2vec.utils", pad_to_multiple=pad_to_multiple)
    _mod("fairseq.models.hubert")
    _mod("fairseq.models.hubert.hubert", HubertModel=type("HubertModel", (nn.Module,), {}))
    _mod("fairseq.models.speech_to_text", lengths_to_padding_mask=lambda l: None)
    _mod("fairseq.modules", GradMultiply=None, TransposeLast=_TransposeLast)
    _mod("fairseq.modules.multihead_attention", MultiheadAttention=MultiheadAttention)
    _mod("fairseq.modules.fairseq_dropout", FairseqDropout=lambda p, module_name=None: nn.Identity())
    _mod("fairseq.modules.quant_noise", quant_noise=lambda m, p, b: m)
    _mod("rotary_embedding_torch", RotaryEmbedding=_StubRotary)
    return fs


def install_misc_stubs():
    _mod("lightning", LightningModule=nn.Module)
    _mod("wandb")
    _mod("jieba")
    tr = _mod("train")
    tr.__path__ = []
    # string constants of reference train/dataset.py:47-57 that the hot path reads (data values, supplied here
    # because train/dataset.py itself needs fairseq's data stack to import)
    _mod("train.dataset", SpeechSampler=object, DEFAULT_SPEECH_PATCH_TOKEN="<sp_patch>",
         DEFAULT_SPEECH_START_TOKEN="<sp_start>", DEFAULT_SPEECH_END_TOKEN="<sp_end>",
         DEFAULT_LATENCY_TOKEN="<latency_{}>", IGNORE_INDEX=-100)

    class AgentStates:
        def __init__(self):
            self.reset()

        def reset(self):
            self.source, self.target = [], []
            self.source_finished, self.target_finished = False, False
            self.source_sample_rate = 0

    class SpeechToTextAgent:
        def __init__(self, args=None):
            self.args = args

    class WriteAction:
        def __init__(self, content, finished):
            self.content, self.finished = content, finished

    class ReadAction:
        pass

    _mod("simuleval")
    _mod("simuleval.agents", SpeechToTextAgent=SpeechToTextAgent)
    _mod("simuleval.agents.states", AgentStates=AgentStates)
    _mod("simuleval.agents.actions", WriteAction=WriteAction, ReadAction=ReadAction)
    _mod("simuleval.utils", entrypoint=lambda c: c)
    _mod("simuleval.data")
    _mod("simuleval.data.segments", SpeechSegment=object)
    _mod("model.patches.patch_hf", patch_hf=lambda: None)  # needs transformers.generation.beam_search (4.47 only)


# ------------------------------------------------------------------ 1. masks ---------------------
def gen_masks(pse):
    out = {}
    cases_t = [(48, 576, 48), (48, None, 48), (96, 576, 96), (50, 20, 16), (7, 3, 2), (33, 10, 12), (144, 100, 48)]
    for n, (s, c, b) in enumerate(cases_t):
        out[f"train_{n}_args"] = np.array([s, -1 if c is None else c, b])
        out[f"train_{n}"] = pse.get_attn_mask_training(s, c, b, device="cpu").numpy()
    cases_i = [(48, 48, 576, 48), (48, 576, 576, 48), (48, 624, 576, 48), (48, 1200, 576, 48), (96, 960, 576, 96),
               (16, 40, 20, 16), (10, 7, 5, 4), (5, 33, 12, 3), (48, 100, 576, 48), (96, 48, 576, 48),
               (24, 30, 20, 16)]
    for n, (s, p, c, b) in enumerate(cases_i):
        out[f"inf_{n}_args"] = np.array([s, p, c, b])
        out[f"inf_{n}"] = pse.get_attn_mask_inference(s, p, c, b, device="cpu").numpy()
    out["n_train"], out["n_inf"] = np.array(len(cases_t)), np.array(len(cases_i))
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **out)
    print("masks.npz", len(out))


# ------------------------------------------------------------------ 2. encoder stack -------------
def build_ref_encoder(pse, cfg, w, dtype):
    """Toy fairseq-shaped modules with the reference's patched methods bound (Appendix A)."""
    from fairseq.models.wav2vec import TransformerEncoder, TransformerSentenceEncoderLayer, Wav2Vec2Model
    from fairseq.modules.multihead_attention import MultiheadAttention
    from oracle.speech_encoder import ENC

    def lin(name, bias=True):
        W = w[name + ".weight"]
        m = nn.Linear(W.shape[1], W.shape[0], bias=bias)
        m.weight.data = W.clone()
        if bias:
            m.bias.data = w[name + ".bias"].clone()
        return m

    def ln(name, dim):
        m = nn.LayerNorm(dim)
        m.weight.data = w[name + ".weight"].clone()
        m.bias.data = w[name + ".bias"].clone()
        return m

    layers = []
    for i in range(cfg.enc_layers):
        p = f"{ENC}encoder.layers.{i}."
        layer = TransformerSentenceEncoderLayer()
        attn = MultiheadAttention(cfg.enc_dim, cfg.enc_heads, self_attention=True)  # patched __init__
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            setattr(attn, n, lin(p + "self_attn." + n))
        layer.self_attn = attn
        layer.self_attn_layer_norm = ln(p + "self_attn_layer_norm", cfg.enc_dim)
        layer.final_layer_norm = ln(p + "final_layer_norm", cfg.enc_dim)
        layer.fc1, layer.fc2 = lin(p + "fc1"), lin(p + "fc2")
        layer.activation_fn = lambda x: F.gelu(x.float()).type_as(x)  # fairseq utils.get_activation_fn("gelu")
        layer.dropout1 = layer.dropout2 = layer.dropout3 = nn.Identity()
        layer.layer_norm_first = True
        layers.append(layer)
    enc = TransformerEncoder()
    enc.layers = nn.ModuleList(layers)
    enc.layer_norm = ln(ENC + "encoder.layer_norm", cfg.enc_dim)
    enc.layer_norm_first, enc.required_seq_len_multiple, enc.dropout, enc.layerdrop = True, 1, 0.0, 0.0
    enc.blocksize = cfg.block_size

    class ToyExtractor(nn.Module):  # stand-in for fairseq ConvFeatureExtractionModel(mode=layer_norm)
        def forward(self_, x):
            from oracle.speech_encoder import conv_feature_extractor
            return conv_feature_extractor(w, cfg, x)

    m = Wav2Vec2Model()
    m.feature_grad_mult, m.feature_extractor = 0.0, ToyExtractor()
    m.layer_norm = ln(ENC + "layer_norm", cfg.conv_dim)
    m.post_extract_proj = lin(ENC + "post_extract_proj")
    m.dropout_input = m.dropout_features = nn.Identity()
    m.input_quantizer, m.crop_seq_to_multiple, m.encoder, m.blocksize = None, 1, enc, cfg.block_size
    m.to(dtype).eval()
    return m


def gen_encoder(pse, msp):
    from infinisst_amd.config import toy_config
    from infinisst_amd import synth
    cfg = toy_config().replace(block_size=16, max_cache_size=40, enc_rope_mode="fp32")
    pse.patch_w2v2(0, 1)
    out = {"block_size": np.array(cfg.block_size), "max_cache_size": np.array(cfg.max_cache_size)}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        w = synth.random_weights(cfg, dtype=dtype, std=0.08, norm_jitter=0.1, seed=1234)
        model = build_ref_encoder(pse, cfg, w, dtype)
        cache = msp.W2V2RoPECache(max_steps=cfg.max_cache_size, layers=[msp.LayerCache() for _ in range(cfg.enc_layers)])
        n_chunks = 6  # 16 frames each: window (40) saturates from chunk 3 on
        audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=3)
        for c in range(n_chunks):
            seg = torch.from_numpy(audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples])
            if c == 0:
                seg = torch.cat([torch.zeros(cfg.first_chunk_offset), seg])
            with torch.no_grad():
                res = model.forward(seg.unsqueeze(0).to(dtype), padding_mask=None, mask=False, features_only=True,
                                    cache=cache)
            out[f"{tag}_x_{c}"] = res["x"].float().numpy()
            out[f"{tag}_k0_{c}"] = cache.layers[0].k.float().numpy()
            out[f"{tag}_state_{c}"] = np.array([cache.src.size(1), cache.src_len, cache.n_steps])
        # one-shot over the same audio with the training mask (streaming == one-shot invariant, fp32 only)
        if tag == "fp32":
            cache1 = msp.W2V2RoPECache(max_steps=cfg.max_cache_size,
                                       layers=[msp.LayerCache() for _ in range(cfg.enc_layers)])
            full = torch.cat([torch.zeros(cfg.first_chunk_offset), torch.from_numpy(audio)])
            with torch.no_grad():
                res = model.forward(full.unsqueeze(0), padding_mask=None, mask=False, features_only=True, cache=cache1)
            out["fp32_oneshot"] = res["x"].numpy()
    out["audio"] = audio
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)
    print("encoder.npz", len(out))


# ------------------------------------------------------------------ 2b. encoder stack, --rope 0 --
def gen_encoder_abs_pos(pse, msp):
    """The reference's own absolute-position branch (patch_w2v2(xpos, rope=0): patch_speech_encoder.py:448-461, :488-493, :823):
    the sinusoid function at offsets on both sides of the bf16 integer grid, and the streaming encoder with it."""
    from infinisst_amd.config import toy_config
    from infinisst_amd import synth
    cfg = toy_config().replace(block_size=16, max_cache_size=40, enc_rope=False)
    out = {"block_size": np.array(cfg.block_size), "max_cache_size": np.array(cfg.max_cache_size)}
    cases = [(0, 16, 64), (16, 16, 64), (250, 16, 64), (1000, 48, 64), (22491, 48, 1024), (65000, 7, 33)]
    for n, (off, length, d) in enumerate(cases):
        out[f"pos_{n}_args"] = np.array([off, length, d])
        out[f"pos_{n}"] = pse.sinusoidal_positional_embedding(off, length, d, "cpu").float().numpy()
    out["n_pos"] = np.array(len(cases))
    pse.patch_w2v2(1, 0)
    try:
        for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
            w = synth.random_weights(cfg, dtype=dtype, std=0.08, norm_jitter=0.1, seed=4321)
            model = build_ref_encoder(pse, cfg, w, dtype)
            cache = msp.W2V2RoPECache(max_steps=cfg.max_cache_size, layers=[msp.LayerCache() for _ in range(cfg.enc_layers)])
            n_chunks = 20  # 16 frames each: positions cross 256, where bf16 stops holding every integer
            audio = synth.synthetic_audio(cfg.chunk_samples * n_chunks, stream_id=5)
            for c in range(n_chunks):
                seg = torch.from_numpy(audio[c * cfg.chunk_samples:(c + 1) * cfg.chunk_samples])
                if c == 0:
                    seg = torch.cat([torch.zeros(cfg.first_chunk_offset), seg])
                with torch.no_grad():
                    res = model.forward(seg.unsqueeze(0).to(dtype), padding_mask=None, mask=False, features_only=True, cache=cache)
                if c < 3 or c >= n_chunks - 4:
                    out[f"{tag}_x_{c}"] = res["x"].float().numpy()
                    out[f"{tag}_k0_{c}"] = cache.layers[0].k.float().numpy()
                    out[f"{tag}_state_{c}"] = np.array([cache.src.size(1), cache.src_len, cache.n_steps])
        out["audio"] = audio
        out["n_chunks"] = np.array(n_chunks)
    finally:
        pse.patch_w2v2(0, 1)
    np.savez_compressed(os.path.join(OUT, "encoder_abs_pos.npz"), **out)
    print("encoder_abs_pos.npz", len(out))


# ------------------------------------------------------------------ 3. length shrink -------------
def gen_shrink(msp):
    torch.manual_seed(7)
    m = msp.ConvFeatureExtractionModel([(32, 2, 2)] * 2, in_d=32)
    for p in m.parameters():
        p.data = p.data + 0.05 * torch.randn_like(p)
    x = torch.randn(1, 32, 48)
    with torch.no_grad():
        y = m(x)
    out = {"x": x.numpy(), "y": y.numpy()}
    for i in range(2):
        out[f"conv{i}"] = m.conv_layers[i][0].weight.data.numpy()
        out[f"ln{i}_w"] = m.conv_layers[i][2][1].weight.data.numpy()
        out[f"ln{i}_b"] = m.conv_layers[i][2][1].bias.data.numpy()
    np.savez_compressed(os.path.join(OUT, "shrink.npz"), **out)
    print("shrink.npz")


# ------------------------------------------------------------------ 4. llama attention -----------
def gen_llm_attention():
    import transformers.models.llama.modeling_llama as ml
    ml.LlamaFlashAttention2 = type("LlamaFlashAttention2", (), {})
    ml.LlamaSdpaAttention = type("LlamaSdpaAttention", (nn.Module,), {})
    import importlib
    pl = importlib.import_module("model.patches.patch_llm")
    from transformers import LlamaConfig
    from infinisst_amd.config import toy_config
    from infinisst_amd import synth
    cfg = toy_config()
    hcfg = LlamaConfig(hidden_size=cfg.llm_dim, num_attention_heads=cfg.llm_heads, num_key_value_heads=cfg.llm_kv_heads,
                       head_dim=cfg.llm_head_dim, max_position_embeddings=131072,
                       rope_parameters={"rope_type": "llama3", "rope_theta": cfg.rope_theta, "factor": cfg.rope_factor,
                                        "low_freq_factor": cfg.rope_low_freq_factor,
                                        "high_freq_factor": cfg.rope_high_freq_factor,
                                        "original_max_position_embeddings": cfg.rope_original_max_pos})
    rot = ml.LlamaRotaryEmbedding(hcfg)  # transformers 5.15 build: secondary stand-in for 4.47's class

    class CatCache:
        def __init__(self):
            self.k, self.v = {}, {}

        def update(self, k, v, layer_idx, kwargs=None):
            if layer_idx in self.k:
                self.k[layer_idx] = torch.cat([self.k[layer_idx], k], dim=-2)
                self.v[layer_idx] = torch.cat([self.v[layer_idx], v], dim=-2)
            else:
                self.k[layer_idx], self.v[layer_idx] = k, v
            return self.k[layer_idx], self.v[layer_idx]

    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        w = synth.random_weights(cfg, dtype=dtype, std=0.05, seed=4321)
        p = "model.layers.0.self_attn."

        class Shim(nn.Module):
            pass
        shim = Shim()
        for n in ("q_proj", "k_proj", "v_proj", "o_proj"):
            W = w[p + n + ".weight"]
            lin = nn.Linear(W.shape[1], W.shape[0], bias=False)
            lin.weight.data = W.clone()
            setattr(shim, n, lin)
        shim.head_dim, shim.num_heads, shim.num_key_value_groups = cfg.llm_head_dim, cfg.llm_heads, cfg.llm_heads // cfg.llm_kv_heads
        shim.attention_dropout, shim.layer_idx = 0.0, 0
        shim.rotary_emb = lambda x, position_ids: rot(x, position_ids)
        shim.eval()

        def causal(q_len, total):  # HF 4-D additive mask, lower-right aligned (attention_mask=None, cache present)
            past = total - q_len
            m = torch.full((q_len, total), float("-inf"))
            m = m.masked_fill(torch.arange(total).unsqueeze(0) <= (torch.arange(q_len).unsqueeze(1) + past), 0.0)
            return m.to(dtype).unsqueeze(0).unsqueeze(0)

        g = torch.Generator().manual_seed(99)
        cache = CatCache()
        seqs = [30, 1, 1, 22, 1]  # prefill, decode, decode, chunked prefill (q>1, past>0), decode
        xs, ys = [], []
        total = 0
        for q_len in seqs:
            x = (0.5 * torch.randn(1, q_len, cfg.llm_dim, generator=g)).to(dtype)
            total += q_len
            pos = torch.arange(total - q_len, total).unsqueeze(0)
            pe = rot(x, pos)
            mask = None if q_len == 1 else causal(q_len, total)
            with torch.no_grad():
                y, _, _ = pl.llama_sdpa_attention_new_forward(shim, hidden_states=x, attention_mask=mask,
                                                              past_key_value=cache, position_embeddings=pe)
            xs.append(x.float().numpy())
            ys.append(y.float().numpy())
        # eviction: keep first 5 (system) + last 20, then decode again -> positions re-index to 0..T-1
        keep = torch.cat([torch.arange(5), torch.arange(total - 20, total)])
        cache.k[0], cache.v[0] = cache.k[0][:, :, keep], cache.v[0][:, :, keep]
        total = 25
        for q_len in (1, 22):
            x = (0.5 * torch.randn(1, q_len, cfg.llm_dim, generator=g)).to(dtype)
            total += q_len
            pe = rot(x, torch.arange(total - q_len, total).unsqueeze(0))
            mask = None if q_len == 1 else causal(q_len, total)
            with torch.no_grad():
                y, _, _ = pl.llama_sdpa_attention_new_forward(shim, hidden_states=x, attention_mask=mask,
                                                              past_key_value=cache, position_embeddings=pe)
            xs.append(x.float().numpy())
            ys.append(y.float().numpy())
        for j, (x, y) in enumerate(zip(xs, ys)):
            out[f"{tag}_x_{j}"], out[f"{tag}_y_{j}"] = x, y
        out[f"{tag}_kcache"] = cache.k[0].float().numpy()  # UNROTATED keys
        out["n_calls"] = np.array(len(xs))
        out["evict_after"] = np.array(len(seqs))
        out["evict_keep"] = keep.numpy()
    cos, sin = rot(torch.zeros(1, 1, 1), torch.arange(64).unsqueeze(0))
    out["rope_cos"], out["rope_sin"] = cos[0].numpy(), sin[0].numpy()
    np.savez_compressed(os.path.join(OUT, "llm_attention.npz"), **out)
    print("llm_attention.npz", len(out))


# ------------------------------------------------------------------ 5. agent policy ---------------
def gen_agent():
    import importlib
    ag = importlib.import_module("agents.infinisst")

    class FakeCache:  # stands in for transformers DynamicCache (key_cache/value_cache lists + iteration)
        def __init__(self, n_layers, T):
            self.key_cache = [torch.arange(T, dtype=torch.float32).view(1, 1, T, 1).repeat(1, 2, 1, 4) for _ in range(n_layers)]
            self.value_cache = [k + 0.5 for k in self.key_cache]

        def __iter__(self):
            return iter(zip(self.key_cache, self.value_cache))

        def __getitem__(self, i):
            return self.key_cache[i], self.value_cache[i]

    class FakeTok:
        pad_token_id = 0

        def decode(self, ids, skip_special_tokens=True):
            return " ".join(f"t{i}" for i in ids)

    SYS, CH = 40, 22
    rng = np.random.default_rng(5)

    class FakeModel:
        device, dtype = torch.device("cpu"), torch.bfloat16
        model = types.SimpleNamespace(speech_features_extracted=False)

        def __init__(self):
            self.calls = []

        def generate(self, **kw):
            st = kw["states"]
            past = kw["past_key_values"]
            T_past = 0 if past is None else past[0][0].size(2)
            n_gen = int(rng.integers(2, 11))  # tokens sampled; the last one is not cached
            ids = kw["input_ids"]
            gen = torch.tensor(rng.integers(10, 900, size=(1, n_gen)))
            self.calls.append(dict(speech=kw["speech_batch"].float().numpy().copy(), enc_ids=kw["encoder_input_ids"].numpy().copy(),
                                   T_past=T_past, n_gen=n_gen, prompt_len=ids.size(1)))
            st.speech_cache = object()
            T_new = T_past + ids.size(1) + n_gen - 1
            cache = FakeCache(2, T_new)
            if past is not None:  # carry over the identity of surviving entries so eviction can be traced
                for i in range(2):
                    cache.key_cache[i][:, :, :T_past] = past.key_cache[i]
                    cache.key_cache[i][:, :, T_past:] = 1000 * len(self.calls) + torch.arange(T_new - T_past).view(1, 1, -1, 1)
                    cache.value_cache[i] = cache.key_cache[i] + 0.5
            else:
                for i in range(2):
                    cache.key_cache[i] = (1000 * len(self.calls) + torch.arange(T_new).view(1, 1, -1, 1)).float().repeat(1, 2, 1, 4)
                    cache.value_cache[i] = cache.key_cache[i] + 0.5
            return types.SimpleNamespace(sequences=torch.cat([ids, gen], dim=1), past_key_values=[cache])

    out = {}
    for variant, (keep_sys, max_cache) in enumerate([(True, 150), (False, 150), (True, 90)]):
        agent = object.__new__(ag.InfiniSST)
        agent.args = types.SimpleNamespace(block_size=48)
        agent.min_start_sec, agent.latency_multiplier, agent.beam = 0.0, 1, 4
        agent.no_repeat_ngram_lookback, agent.no_repeat_ngram_size, agent.repetition_penalty = 100, 5, 1.2
        agent.max_new_tokens, agent.do_sample, agent.top_p, agent.top_k, agent.epsilon_cutoff, agent.temperature = 10, False, 1.0, 0, 0.0, 1.0
        agent.pseudo_batch_size, agent.max_llm_cache_size, agent.always_cache_system_prompt = 1, max_cache, keep_sys
        agent.cache_checkpoints, agent.dpo_sampling, agent.bad_words_ids = [], False, []
        agent.target_lang, agent.tokenizer, agent.model = "German", FakeTok(), FakeModel()
        agent.system_prompt_size = SYS

        def prep_inputs(states, _a=agent):
            n = CH + (SYS if states.speech_cache is None else 0)
            return torch.arange(n).unsqueeze(0)
        agent._prepare_inputs = prep_inputs
        st = ag.S2TAgentStates(src_len=0, speech_cache=None, past_key_values=None, target_ids=[], segment_idx=0,
                               translations_list=[])
        st.reset()
        st.source_sample_rate = 16000
        seg_lens = [15360, 15360, 15360, 15360, 15361, 15359, 30720, 1, 15360, 7000, 15360, 15360]
        audio = (0.1 * rng.standard_normal(sum(seg_lens))).astype(np.float32)
        pos, recs = 0, []
        for c, n in enumerate(seg_lens):
            st.source.extend(audio[pos:pos + n].tolist())
            pos += n
            st.source_finished = c == len(seg_lens) - 1
            act = agent.policy(st)
            kv0 = st.past_key_values.key_cache[0][0, 0, :, 0].numpy().copy()
            out[f"v{variant}_kvtrace_{c}"] = kv0
            out[f"v{variant}_ckpt_{c}"] = np.array(agent.cache_checkpoints)
            out[f"v{variant}_speech_{c}"] = agent.model.calls[-1]["speech"]
            out[f"v{variant}_encids_{c}"] = agent.model.calls[-1]["enc_ids"]
            recs.append([agent.model.calls[-1]["n_gen"], agent.model.calls[-1]["prompt_len"], len(st.target_ids),
                         int(isinstance(act, ag.WriteAction)), int(getattr(act, "finished", False))])
        out[f"v{variant}_recs"] = np.array(recs)
        out[f"v{variant}_target_ids"] = np.array(st.target_ids)
        out[f"v{variant}_cfg"] = np.array([int(keep_sys), max_cache, SYS, CH])
        out[f"v{variant}_seg_lens"] = np.array(seg_lens)
        out[f"v{variant}_audio"] = audio
    np.savez_compressed(os.path.join(OUT, "agent.npz"), **out)
    print("agent.npz", len(out))


# ------------------------------------------------------------------ 6. speech splice --------------
def gen_splice():
    import importlib
    mllm = importlib.import_module("model.llm")
    from transformers import LlamaModel
    captured = {}

    def fake_super_forward(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, **kw):
        captured["embeds"] = inputs_embeds
        return None
    orig = LlamaModel.forward
    LlamaModel.forward = fake_super_forward
    try:
        D, V = 16, 1100
        cfg = mllm.SpeechLlamaConfig(hidden_size=D, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2,
                                     num_key_value_heads=1, vocab_size=V)
        cfg.user_token_id, cfg.assist_token_id, cfg.start_header_id = 882, 781, 1006
        model = mllm.SpeechLlamaModel(cfg)
        model.inference = True
        out = {"ids_cfg": np.array([882, 781, 1006, 1024])}
        g = torch.Generator().manual_seed(3)
        for case, n_sp in enumerate([12, 24, 12]):
            feats = torch.randn(1, n_sp + (6 if case == 2 else 0), D, generator=g)  # case 2: surplus features dropped

            class Enc:
                def set_blocksize(self, m):
                    pass

                def encode_speech(self, sb, sl, cache=None):
                    return feats, "CACHE"
            model.speech_encoder = Enc()
            model.speech_features_extracted = False
            turn = [1006, 882, 1007, 271] + [1024] * n_sp + [1009, 1006, 781, 1007, 271]
            ids = ([1000, 1006, 912, 1007, 271, 5, 6, 7, 882, 781, 1009] if case != 1 else [1009]) + turn
            states = types.SimpleNamespace(speech_cache=None)
            with torch.no_grad():
                model.forward(input_ids=torch.tensor([ids]), speech_batch=torch.zeros(1, 10), states=states, multiplier=1)
            out[f"ids_{case}"] = np.array(ids)
            out[f"feats_{case}"] = feats[0].numpy()
            out[f"table_{case}"] = model.embed_tokens.weight.data.numpy()
            out[f"embeds_{case}"] = captured["embeds"][0].numpy()
        np.savez_compressed(os.path.join(OUT, "splice.npz"), **out)
        print("splice.npz")
    finally:
        LlamaModel.forward = orig



# --------------------------------------------------------------------------------------------
# beam scorer: the reference's own beam_search_process / beam_search_finalize / beam_hypotheses_add
# (model/patches/patch_hf.py:43-302), executed from their source text on a stand-in scorer object
# --------------------------------------------------------------------------------------------
def gen_beam_scorer():
    """patch_hf.py cannot be imported (transformers 5.15 has no generation.beam_search); its three scorer functions only
    need torch, UserDict and an object with the BeamSearchScorer / BeamHypotheses attributes they touch, so their definitions
    are compiled straight from the reference file.  The one third-party piece, BeamHypotheses.is_done (transformers 4.47,
    early_stopping=False), is restated below and is therefore NOT pinned by this fixture."""
    import ast
    from collections import UserDict
    from typing import Dict, List, Optional, Tuple, Union
    path = os.path.join(REF, "model", "patches", "patch_hf.py")
    src = open(path).read()
    tree = ast.parse(src)
    want = {"beam_search_process", "beam_search_finalize", "beam_hypotheses_add"}

    class KVStandIn:  # the subset of DynamicCache the functions use: DynamicCache(n).update(k, v, layer)
        def __init__(self, n=0):
            self.layers = {}

        def update(self, k, v, i):
            self.layers[i] = (k, v)

    ns = {"torch": torch, "UserDict": UserDict, "Optional": Optional, "Union": Union, "List": List, "Dict": Dict, "Tuple": Tuple,
          "DynamicCache": KVStandIn}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            exec(compile(ast.Module([node], []), path, "exec"), ns)

    class Hyps:
        def __init__(self, num_beams, length_penalty):
            self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, False
            self.beams, self.worst_score = [], 1e9

        def __len__(self):
            return len(self.beams)

        add = ns["beam_hypotheses_add"]

        def is_done(self, best_sum_logprobs, cur_len, decoder_prompt_len=0):  # [3P transformers 4.47, restated]
            if len(self) < self.num_beams:
                return False
            highest_attainable = best_sum_logprobs / (cur_len - decoder_prompt_len) ** self.length_penalty
            return self.worst_score >= highest_attainable

    class Scorer:
        process = ns["beam_search_process"]
        finalize = ns["beam_search_finalize"]

        def __init__(self, B, length_penalty):
            self.num_beams = self.group_size = B
            self.num_beam_groups, self.num_beam_hyps_to_keep = 1, 1
            self.device = torch.device("cpu")
            self._beam_hyps = [Hyps(B, length_penalty)]
            self._done = torch.tensor([False])

    out = {}
    cases = [(4, 1.0, 0.25, 11, 7), (4, 1.0, 0.0, 12, 6), (2, 1.0, 0.4, 13, 9), (4, 0.6, 0.3, 14, 8), (3, 1.0, 0.6, 15, 10)]
    eos = [901, 908, 909]
    for ci, (B, lp, p_eos, seed, steps) in enumerate(cases):
        g = torch.Generator().manual_seed(seed)
        sc = Scorer(B, lp)
        prompt_len = 5
        max_length = prompt_len + steps
        n_keep = max(2, 1 + len(eos)) * B
        input_ids = torch.randint(10, 800, (1, prompt_len), generator=g).repeat(B, 1)
        marker = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)  # one-layer "KV": which beam's cache travels where
        beam_scores = torch.tensor([0.0] + [-1e9] * (B - 1))
        n_steps = 0
        for st in range(steps):
            # candidates: every beam proposes tokens with scores below its own running score, merged and sorted like topk
            cand = []
            for b in range(B):
                for _ in range(n_keep):
                    tok = int(torch.randint(10, 800, (1,), generator=g))
                    if float(torch.rand(1, generator=g)) < p_eos:
                        tok = eos[int(torch.randint(0, len(eos), (1,), generator=g))]
                    cand.append((float(beam_scores[b]) - float(torch.rand(1, generator=g)) * 3.0, tok, b))
            cand.sort(key=lambda x: -x[0])
            cand = cand[:n_keep]
            ns_ = torch.tensor([[c[0] for c in cand]], dtype=torch.float32)
            nt_ = torch.tensor([[c[1] for c in cand]])
            ni_ = torch.tensor([[c[2] for c in cand]])
            kv = [(marker + 100.0 * st, marker + 100.0 * st + 0.5)]
            try:
                res = sc.process(input_ids, ns_, nt_, ni_, pad_token_id=904, eos_token_id=eos, beam_indices=None,
                                 decoder_prompt_len=prompt_len, past_key_values=kv)
            except ValueError:
                break  # fewer than B non-EOS candidates: the reference raises; stop the case here
            out[f"c{ci}_s{st}_in_ids"] = input_ids.numpy().copy()
            out[f"c{ci}_s{st}_scores"] = ns_.numpy()[0]
            out[f"c{ci}_s{st}_tokens"] = nt_.numpy()[0]
            out[f"c{ci}_s{st}_beams"] = ni_.numpy()[0]
            out[f"c{ci}_s{st}_next_scores"] = res["next_beam_scores"].numpy()
            out[f"c{ci}_s{st}_next_tokens"] = res["next_beam_tokens"].numpy()
            out[f"c{ci}_s{st}_next_beams"] = res["next_beam_indices"].numpy()
            out[f"c{ci}_s{st}_done"] = np.array(bool(sc._done[0]))
            out[f"c{ci}_s{st}_n_hyps"] = np.array(len(sc._beam_hyps[0]))
            out[f"c{ci}_s{st}_worst"] = np.array(sc._beam_hyps[0].worst_score, dtype=np.float64)
            beam_scores = res["next_beam_scores"]
            idx = res["next_beam_indices"]
            input_ids = torch.cat([input_ids[idx], res["next_beam_tokens"].unsqueeze(-1)], dim=-1)  # patch_hf.py beam loop
            marker = marker[idx]
            n_steps = st + 1
            if bool(sc._done[0]) or input_ids.shape[-1] >= max_length:
                break
        kv = [(marker + 100.0 * n_steps, marker + 100.0 * n_steps + 0.5)]
        fin = sc.finalize(input_ids, beam_scores, None, None, max_length=max_length, pad_token_id=904, eos_token_id=eos,
                          beam_indices=None, decoder_prompt_len=prompt_len, past_key_values=kv)
        out[f"c{ci}_cfg"] = np.array([B, prompt_len, max_length, n_steps], dtype=np.int64)
        out[f"c{ci}_lp"] = np.array(lp)
        out[f"c{ci}_final_ids"] = input_ids.numpy().copy()
        out[f"c{ci}_final_scores"] = beam_scores.numpy().copy()
        out[f"c{ci}_sequence"] = fin["sequences"][0].numpy()
        out[f"c{ci}_sequence_score"] = fin["sequence_scores"].numpy()
        out[f"c{ci}_kv_marker"] = fin["past_key_values"][0].layers[0][0].reshape(-1).numpy()  # which (step, beam) cache won
    out["n_cases"] = np.array(len(cases))
    out["eos"] = np.array(eos)
    np.savez_compressed(os.path.join(OUT, "beam_scorer.npz"), **out)
    print("beam_scorer.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------
# beam-search LOOP: the reference's own generation_mixin_beam_search (+ _expand_inputs_for_generation and the three scorer
# functions), model/patches/patch_hf.py:43-342,687-967, executed from their source text on a toy model
# --------------------------------------------------------------------------------------------
def gen_beam_loop():
    """patch_hf.py cannot be imported under transformers 5.15, so the function DEFINITIONS are compiled from the reference file's
    own AST (as gen_beam_scorer does) into a namespace that supplies the names they use.  `self` is a stand-in model: toy forward
    (toy_beam_forward), `prepare_inputs_for_generation` always handing over the full input_ids with the model embedding the whole
    prompt on the first call and the last token afterwards (model/llm.py:69,114-115,272-295), a 4.47-style DynamicCache restated
    below ([3P]: key_cache / value_cache lists, iteration yields (k, v), update() concatenates on dim -2, reorder_cache =
    index_select on dim 0 -- `_temporary_reorder_cache`), stopping criteria and logits processors from the image's transformers
    5.15 built by ITS `_get_logits_processor` from the agent's generate kwargs (agents/infinisst.py:307-332) -- whose class ORDER
    is stored as a secondary pin of oracle/generate.py::process_logits (patch_hf.py:586-600; 4.47 itself is absent)."""
    import ast
    from collections import UserDict
    from typing import Any, Callable, Dict, List, Optional, Tuple, Union
    import transformers
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM
    from transformers.generation.logits_process import LogitsProcessorList
    from transformers.generation.stopping_criteria import MaxLengthCriteria, StoppingCriteriaList
    from transformers.generation.utils import (GenerateBeamDecoderOnlyOutput, GenerateBeamEncoderDecoderOutput, GenerateBeamOutput)
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..")))
    from toy_beam_model import toy_beam_forward
    path = os.path.join(REF, "model", "patches", "patch_hf.py")
    tree = ast.parse(open(path).read())
    want = {"beam_search_process", "beam_search_finalize", "beam_hypotheses_add", "generation_mixin_beam_search",
            "generation_mixin_expand_inputs_for_generation"}

    class Cache447:  # [3P] transformers 4.47 DynamicCache, the subset the reference touches
        def __init__(self, n=None):
            self.key_cache, self.value_cache = [], []

        def __len__(self):
            return len(self.key_cache)

        def __iter__(self):
            for i in range(len(self)):
                yield self.key_cache[i], self.value_cache[i]

        def __getitem__(self, i):
            return self.key_cache[i], self.value_cache[i]

        def update(self, k, v, layer_idx, cache_kwargs=None):
            if len(self.key_cache) <= layer_idx:
                self.key_cache.append(k)
                self.value_cache.append(v)
            else:
                self.key_cache[layer_idx] = torch.cat([self.key_cache[layer_idx], k], dim=-2)
                self.value_cache[layer_idx] = torch.cat([self.value_cache[layer_idx], v], dim=-2)
            return self.key_cache[layer_idx], self.value_cache[layer_idx]

        def reorder_cache(self, beam_idx):
            for i in range(len(self)):
                self.key_cache[i] = self.key_cache[i].index_select(0, beam_idx)
                self.value_cache[i] = self.value_cache[i].index_select(0, beam_idx)

    ns = {"torch": torch, "nn": nn, "UserDict": UserDict, "Optional": Optional, "Union": Union, "List": List, "Dict": Dict,
          "Tuple": Tuple, "Any": Any, "Callable": Callable, "DynamicCache": Cache447, "BeamScorer": object,
          "LogitsProcessorList": LogitsProcessorList, "StoppingCriteriaList": StoppingCriteriaList, "GenerationConfig": GenerationConfig,
          "GenerateBeamOutput": GenerateBeamOutput, "GenerateBeamDecoderOnlyOutput": GenerateBeamDecoderOnlyOutput,
          "GenerateBeamEncoderDecoderOutput": GenerateBeamEncoderDecoderOutput}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want:
            node.decorator_list = []  # @staticmethod / @torch.no_grad: bound below
            exec(compile(ast.Module([node], []), path, "exec"), ns)

    class Hyps:
        def __init__(self, num_beams, length_penalty):
            self.num_beams, self.length_penalty, self.early_stopping = num_beams, length_penalty, False
            self.beams, self.worst_score = [], 1e9

        def __len__(self):
            return len(self.beams)

        add = ns["beam_hypotheses_add"]

        def is_done(self, best_sum_logprobs, cur_len, decoder_prompt_len=0):  # [3P transformers 4.47, restated]
            if len(self) < self.num_beams:
                return False
            highest_attainable = best_sum_logprobs / (cur_len - decoder_prompt_len) ** self.length_penalty
            return self.worst_score >= highest_attainable

    trace = {}

    class Scorer:
        finalize = ns["beam_search_finalize"]

        def __init__(self, B, length_penalty):
            self.num_beams = self.group_size = B
            self.num_beam_groups, self.num_beam_hyps_to_keep = 1, 1
            self.device = torch.device("cpu")
            self._beam_hyps = [Hyps(B, length_penalty)]
            self._done = torch.tensor([False])

        @property
        def is_done(self):  # [3P] BeamSearchScorer.is_done
            return bool(self._done.all())

        def process(self, input_ids, next_scores, next_tokens, next_indices, **kw):
            res = ns["beam_search_process"](self, input_ids, next_scores, next_tokens, next_indices, **kw)
            st = trace["n"]
            pre = trace["pre"]
            out[pre + f"s{st}_in_ids"] = input_ids.numpy().copy()
            out[pre + f"s{st}_cand_scores"] = next_scores[0].numpy().copy()
            out[pre + f"s{st}_cand_tokens"] = next_tokens[0].numpy().copy()
            out[pre + f"s{st}_cand_beams"] = next_indices[0].numpy().copy()
            out[pre + f"s{st}_next_scores"] = res["next_beam_scores"].numpy().copy()
            out[pre + f"s{st}_next_tokens"] = res["next_beam_tokens"].numpy().copy()
            out[pre + f"s{st}_next_beams"] = res["next_beam_indices"].numpy().copy()
            out[pre + f"s{st}_done"] = np.array(bool(self._done[0]))
            trace["n"] = st + 1
            return res

    # the processors of the agent's generate() call, built by transformers 5.15's own _get_logits_processor
    V, D = 61, 12
    tiny = LlamaForCausalLM(LlamaConfig(hidden_size=16, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2,
                                        num_key_value_heads=1, vocab_size=V))
    out = {"transformers_version": np.array(transformers.__version__)}
    cases = [(4, 1.0, 9, 10, 0.0, 101), (4, 1.0, 14, 10, 2.5, 102), (2, 1.0, 7, 12, 1.5, 103), (3, 0.6, 11, 9, 2.0, 104),
             (4, 1.0, 22, 20, 1.0, 105), (4, 1.0, 8, 6, 4.0, 106), (4, 1.0, 9, 8, 1.0, 107), (2, 1.0, 12, 10, 2.0, 108)]
    # the last two run the loop's do_sample branch (:871-875: beam sample): (temperature, top_k, top_p, epsilon_cutoff).  torch.multinomial's random stream cannot be
    # restated, so the compiled function sees a `torch` whose multinomial is the oracle's counter-based sequential draw (oracle/generate.py
    # draw_without_replacement at sample_uniform(seed, 0, 0, 64 step + j)): everything else of the branch -- softmax over all beams, gather, sort, scorer -- is the reference's
    samples = {6: (0.8, 0, 1.0, 0.0), 7: (1.6, 40, 0.995, 0.0)}  # (every step must leave >= max(2, 1 + n_eos) * beams tokens with non-zero probability: torch.multinomial raises otherwise)
    from oracle import generate as ogen

    class TorchWithCountedDraw:
        def __init__(self, seed):
            self.seed = seed

        def __getattr__(self, name):
            return getattr(torch, name)

        def multinomial(self, probs, num_samples, replacement=False):
            assert not replacement and probs.shape[0] == 1
            us = [ogen.sample_uniform(self.seed, 0, 0, 64 * trace["n"] + j) for j in range(num_samples)]
            return torch.tensor([ogen.draw_without_replacement(probs[0], num_samples, us)], dtype=torch.long)
    eos = [57, 58, 59]
    for ci, (B, lp, prompt_len, max_new, eos_bias, seed) in enumerate(cases):
        g = torch.Generator().manual_seed(seed)
        E = torch.randn(V, D, generator=g)
        O = torch.randn(D, V, generator=g) * 1.5
        bias = torch.randn(V, generator=g) * 0.3
        bias[eos] += eos_bias - 2.0  # EOS candidates show up among the top ranks in the biased cases
        decay = 0.8
        fwd = toy_beam_forward(E, O, bias, decay)
        small = 9 if ci in (1, 4) else V - 4  # a small alphabet makes repeated n-grams (banned tokens) likely
        prompt = torch.randint(3, small, (1, prompt_len), generator=g)
        enc_ids = torch.randint(3, small, (1, 30), generator=g)
        suppress = [5, 40]
        ngram = 3 if ci in (1, 4) else 5
        smp = samples.get(ci)
        skw = {} if smp is None else dict(do_sample=True, temperature=smp[0], top_k=smp[1], top_p=smp[2], epsilon_cutoff=smp[3])
        gc = GenerationConfig(repetition_penalty=1.2, no_repeat_ngram_size=ngram, encoder_no_repeat_ngram_size=ngram,
                              suppress_tokens=suppress, num_beams=B, max_new_tokens=max_new, pad_token_id=60, eos_token_id=eos, **skw)
        if smp is not None:
            gc._eos_token_tensor = torch.tensor(eos)  # (generate() sets it in _prepare_special_tokens; the warpers' min_tokens_to_keep reads it)
        procs = tiny._get_logits_processor(generation_config=gc, input_ids_seq_length=prompt_len, encoder_input_ids=enc_ids,
                                           prefix_allowed_tokens_fn=None, logits_processor=LogitsProcessorList(), device="cpu",
                                           model_kwargs={})
        out[f"c{ci}_processor_order"] = np.array([type(p).__name__ for p in procs])

        class Model:
            config = types.SimpleNamespace(is_encoder_decoder=False)

            def __init__(self):
                self.first = True  # model.model.speech_features_extracted = False (agents/infinisst.py:306)

            def _get_initial_cache_position(self, input_ids, model_kwargs):
                return model_kwargs

            def _has_unfinished_sequences(self, this_peer_finished, synced_gpus, device=None):
                return not this_peer_finished

            def prepare_inputs_for_generation(self, input_ids, past_key_values=None, **kw):  # model/llm.py:272-295
                return {"input_ids": input_ids, "past_key_values": past_key_values}

            def __call__(self, input_ids=None, past_key_values=None, return_dict=True):
                toks = input_ids if self.first else input_ids[:, -1:]  # model/llm.py:69 vs :114-115
                self.first = False
                rows = []
                k_new = E[toks]  # (beams, T, D): the toy cache holds the embedding rows of every consumed token
                k_all, _ = past_key_values.update(k_new.unsqueeze(1), k_new.unsqueeze(1) + 0.5, 0)
                for b in range(input_ids.shape[0]):
                    kv = [k_all[b, 0, i] for i in range(k_all.shape[2])]
                    n = len(kv)
                    wts = decay ** torch.arange(n - 1, -1, -1, dtype=torch.float32)
                    h = (torch.stack(kv) * wts[:, None]).sum(0)
                    rows.append(torch.tanh(h) @ O + bias)
                logits = torch.stack(rows).unsqueeze(1)  # (beams, 1, V): only [:, -1] is read (:833)
                return types.SimpleNamespace(logits=logits, past_key_values=past_key_values)

            def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False):
                model_kwargs["past_key_values"] = outputs.past_key_values
                return model_kwargs

            def _temporary_reorder_cache(self, past_key_values, beam_idx):  # [3P] 4.47: DynamicCache.reorder_cache
                past_key_values.reorder_cache(beam_idx)
                return past_key_values

        # the stream's cache before this chunk: `n_past` tokens already consumed (batch 1), expanded by the reference's own function
        n_past = [0, 6, 3, 0, 17, 5, 4, 0][ci]
        past_tokens = torch.randint(3, V - 4, (n_past,), generator=g)
        past = Cache447()
        if n_past:
            past.update(E[past_tokens][None, None], E[past_tokens][None, None] + 0.5, 0)
        else:
            past.update(torch.zeros(1, 1, 0, D), torch.zeros(1, 1, 0, D), 0)
        ids_x, kw_x = ns["generation_mixin_expand_inputs_for_generation"](expand_size=B, is_encoder_decoder=False, input_ids=prompt,
                                                                          past_key_values=past)
        gcfg = types.SimpleNamespace(_pad_token_tensor=torch.tensor(60), _eos_token_tensor=torch.tensor(eos), output_attentions=False,
                                     output_hidden_states=False, output_scores=False, output_logits=False,
                                     return_dict_in_generate=True, low_memory=False, do_sample=smp is not None)
        trace.update(n=0, pre=f"c{ci}_")
        scorer = Scorer(B, lp)
        ns["generation_mixin_beam_search"].__globals__["torch"] = TorchWithCountedDraw(seed) if smp is not None else torch
        out[f"c{ci}_sample"] = np.array([0.0, 0.0, 0.0, 0.0] if smp is None else list(smp), dtype=np.float64)
        out[f"c{ci}_seed"] = np.array(seed)
        res = ns["generation_mixin_beam_search"](Model(), ids_x, scorer, logits_processor=procs,
                                                 stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(prompt_len + max_new)]),
                                                 generation_config=gcfg, synced_gpus=False, **kw_x)
        out[f"c{ci}_cfg"] = np.array([B, prompt_len, max_new, trace["n"], ngram], dtype=np.int64)
        out[f"c{ci}_lp"] = np.array(lp)
        out[f"c{ci}_E"], out[f"c{ci}_O"], out[f"c{ci}_bias"], out[f"c{ci}_decay"] = E.numpy(), O.numpy(), bias.numpy(), np.array(decay)
        out[f"c{ci}_prompt"] = prompt[0].numpy()
        out[f"c{ci}_enc_ids"] = enc_ids[0].numpy()
        out[f"c{ci}_past_tokens"] = past_tokens.numpy()
        out[f"c{ci}_suppress"] = np.array(suppress)
        out[f"c{ci}_sequence"] = res.sequences[0].numpy()
        out[f"c{ci}_hyp_scores"] = np.array(sorted(h[0] for h in scorer._beam_hyps[0].beams), dtype=np.float64)  # :937-939 drops sequences_scores
        out[f"c{ci}_winner_kv"] = res.past_key_values[0].key_cache[0][0, 0].numpy()  # (T, D): the cache that travels with the winner
        print(f"  beam loop case {ci}: {trace['n']} steps, sequence {res.sequences[0].tolist()[prompt_len:]}, order {list(out[f'c{ci}_processor_order'])}")
    out["n_cases"] = np.array(len(cases))
    out["eos"] = np.array(eos)
    np.savez_compressed(os.path.join(OUT, "beam_loop.npz"), **out)
    print("beam_loop.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------
# logits processors: HF transformers' own classes (the image has 5.15.0; the reference pins 4.47.0 -- the four processors
# used by agents/infinisst.py:307-332 have had the same semantics since 4.2x)
# --------------------------------------------------------------------------------------------
def gen_logits_processors():
    import transformers
    from transformers.generation.logits_process import (EncoderNoRepeatNGramLogitsProcessor, NoRepeatNGramLogitsProcessor,
                                                        RepetitionPenaltyLogitsProcessor, SuppressTokensLogitsProcessor)
    out = {"transformers_version": np.array(transformers.__version__)}
    V = 300
    cases = []
    g = torch.Generator().manual_seed(77)
    for ci in range(12):
        n_ids = [1, 3, 4, 9, 40, 120][ci % 6]
        n_enc = [0, 2, 5, 30, 100, 100][(ci * 5) % 6]
        ngram = [5, 3, 2, 5, 4, 1][ci % 6]
        penalty = [1.2, 1.0, 1.5, 1.2, 1.2, 2.0][ci % 6]
        small = 12 if ci % 2 else V  # a small alphabet makes repeated n-grams likely
        ids = torch.randint(0, small, (1, n_ids), generator=g)
        enc = torch.randint(0, small, (1, max(n_enc, 1)), generator=g)[:, :n_enc]
        if n_enc >= ngram and n_ids >= ngram:  # make sure an encoder n-gram match exists in some cases
            enc[0, :ngram - 1] = ids[0, -(ngram - 1):] if ngram > 1 else enc[0, :0]
        suppress = sorted(set(int(t) for t in torch.randint(0, V, (ci % 4,), generator=g)))
        scores = torch.randn(1, V, generator=g) * 3
        s = scores.clone()
        s = RepetitionPenaltyLogitsProcessor(penalty=penalty)(ids, s) if penalty != 1.0 else s
        s = NoRepeatNGramLogitsProcessor(ngram)(ids, s)
        if n_enc > 0:
            s = EncoderNoRepeatNGramLogitsProcessor(ngram, enc)(ids, s)
        if suppress:
            s = SuppressTokensLogitsProcessor(suppress, device="cpu")(ids, s)
        out[f"c{ci}_ids"] = ids[0].numpy()
        out[f"c{ci}_enc"] = enc[0].numpy()
        out[f"c{ci}_suppress"] = np.array(suppress, dtype=np.int64)
        out[f"c{ci}_cfg"] = np.array([ngram, penalty], dtype=np.float64)
        out[f"c{ci}_scores"] = scores[0].numpy()
        out[f"c{ci}_out"] = s[0].numpy()
        cases.append(ci)
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "logits_processors.npz"), **out)
    print("logits_processors.npz:", len(out), "arrays, transformers", transformers.__version__)


# --------------------------------------------------------------------------------------------
# the sample branch's warpers: transformers' own classes, built by ITS _get_logits_processor from the agent's kwargs with do_sample=True
# (agents/infinisst.py:311-315 -> patch_hf.py:586-600; 4.47 is absent, the image has 5.15)
# --------------------------------------------------------------------------------------------
def gen_sampling_warpers():
    import transformers
    from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM
    from transformers.generation.logits_process import LogitsProcessorList
    V = 400
    tiny = LlamaForCausalLM(LlamaConfig(hidden_size=16, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1, vocab_size=V))
    out = {"transformers_version": np.array(transformers.__version__)}
    g = torch.Generator().manual_seed(99)
    cases = [(0.7, 50, 0.9, 0.0), (1.0, 0, 0.8, 0.0), (1.3, 10, 1.0, 0.0), (1.0, 0, 1.0, 0.004), (0.5, 40, 0.95, 0.002), (1.0, 1, 1.0, 0.0), (2.0, 0, 0.3, 0.0),
             (0.8, 400, 0.99, 0.0005),
             # under beam search the same function builds the warpers with min_tokens_to_keep = len(eos ids) + 1 (5th entry: the eos id count)
             (1.0, 2, 1.0, 0.0, 3), (1.0, 0, 0.05, 0.0, 3), (1.0, 0, 1.0, 0.2, 3), (0.9, 3, 0.1, 0.05, 1)]
    for ci, case in enumerate(cases):
        temp, top_k, top_p, eps = case[:4]
        n_eos = case[4] if len(case) > 4 else None
        kw = dict(do_sample=True, temperature=temp, top_k=top_k, top_p=top_p, pad_token_id=0)
        if eps > 0:
            kw["epsilon_cutoff"] = eps
        if n_eos is not None:
            kw.update(num_beams=4, eos_token_id=list(range(V - n_eos, V)))
        gc = GenerationConfig(**kw)
        if n_eos is not None:
            gc._eos_token_tensor = torch.tensor(list(range(V - n_eos, V)))
        out[f"c{ci}_min_keep"] = np.array(1 if n_eos is None else n_eos + 1)
        procs = tiny._get_logits_processor(generation_config=gc, input_ids_seq_length=3, encoder_input_ids=None, prefix_allowed_tokens_fn=None,
                                           logits_processor=LogitsProcessorList(), device="cpu", model_kwargs={})
        scores = torch.randn(1, V, generator=g) * (3.0 if ci % 2 else 1.5)
        scores[0, torch.randint(0, V, (7,), generator=g)] = float("-inf")  # tokens the processors banned
        ids = torch.zeros(1, 3, dtype=torch.long)
        outp = procs(ids, scores.clone())
        out[f"c{ci}_order"] = np.array([type(p).__name__ for p in procs])
        out[f"c{ci}_cfg"] = np.array([temp, top_k, top_p, eps], dtype=np.float64)
        out[f"c{ci}_scores"] = scores[0].numpy()
        out[f"c{ci}_out"] = outp[0].numpy()
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(os.path.join(OUT, "sampling_warpers.npz"), **out)
    print("sampling_warpers.npz:", len(out), "arrays, transformers", transformers.__version__)


# --------------------------------------------------------------------------------------------
# HF Llama building blocks (LlamaRMSNorm, apply_rotary_pos_emb, LlamaMLP, repeat_kv) of the image's transformers 5.15.0
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def gen_llama_blocks():
    import transformers
    from transformers.models.llama import modeling_llama as ml
    from transformers.models.llama.configuration_llama import LlamaConfig
    out = {"transformers_version": np.array(transformers.__version__)}
    g = torch.Generator().manual_seed(5)
    D, I, hd = 64, 160, 16
    for dt_name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        x = (torch.randn(1, 7, D, generator=g) * 2).to(dt)
        nw = (1 + 0.2 * torch.randn(D, generator=g)).to(dt)
        norm = ml.LlamaRMSNorm(D, eps=1e-5)
        norm.weight.data = nw.clone()
        out[f"{dt_name}_x"] = x.float().numpy()
        out[f"{dt_name}_norm_w"] = nw.float().numpy()
        out[f"{dt_name}_norm"] = norm.to(dt)(x).float().detach().numpy()
        cfg = LlamaConfig(hidden_size=D, intermediate_size=I, hidden_act="silu", mlp_bias=False)
        mlp = ml.LlamaMLP(cfg)
        for n in ("gate_proj", "up_proj", "down_proj"):
            wt = getattr(mlp, n).weight
            wt.data = (torch.randn(wt.shape, generator=g) * 0.1)
            out[f"{dt_name}_{n}"] = wt.data.to(dt).float().numpy()
        out[f"{dt_name}_mlp"] = mlp.to(dt)(x).float().detach().numpy()
        q = torch.randn(1, 4, 7, hd, generator=g).to(dt)
        k = torch.randn(1, 2, 7, hd, generator=g).to(dt)
        ang = torch.randn(1, 7, hd // 2, generator=g)
        emb = torch.cat((ang, ang), dim=-1)
        cos, sin = emb.cos().to(dt), emb.sin().to(dt)
        qr, kr = ml.apply_rotary_pos_emb(q, k, cos, sin)
        out[f"{dt_name}_q"], out[f"{dt_name}_k"] = q.float().numpy(), k.float().numpy()
        out[f"{dt_name}_cos"], out[f"{dt_name}_sin"] = cos[0].float().numpy(), sin[0].float().numpy()
        out[f"{dt_name}_q_rot"], out[f"{dt_name}_k_rot"] = qr.float().numpy(), kr.float().numpy()
        out[f"{dt_name}_k_rep"] = ml.repeat_kv(k, 2).float().numpy()
    np.savez_compressed(os.path.join(OUT, "llama_blocks.npz"), **out)
    print("llama_blocks.npz:", len(out), "arrays, transformers", transformers.__version__)


# ------------------------------------------------------------------ 10. chunk prompts (_prepare_inputs) ------------
def gen_prompts():
    """The reference's own `_prepare_inputs` (agents/infinisst.py:225-268) over two tokenizers whose `apply_chat_template` returns
    tensors as transformers 4.47 does: the hand-written stub (tests/stub_tokenizer.py) and a real transformers tokenizer with a
    Llama-3.1-shaped jinja chat template (tests/tiny_tokenizer.py).  Stored: ids of the first and of a later chunk and
    `system_prompt_size` for multipliers 1..4, llama31 and llama3 branches."""
    import importlib
    import tempfile
    import transformers
    sys.path.insert(0, os.path.abspath(os.path.join(OUT, "..")))
    from stub_tokenizer import StubTokenizer
    from tiny_tokenizer import build_tokenizer_dir
    from infinisst_amd import harness
    from infinisst_amd.config import toy_config
    ag = importlib.import_module("agents.infinisst")
    cfg = toy_config()

    class TensorTemplate:  # 4.47 semantics: return_tensors='pt' -> a LongTensor (batch, len)
        def __init__(self, tok):
            self.tok = tok
            self.eos_token_id = tok.eos_token_id

        def apply_chat_template(self, batch, **kw):
            kw.pop("return_tensors", None)
            r = self.tok.apply_chat_template(batch, **kw)
            if hasattr(r, "keys"):
                r = r["input_ids"]
            return torch.tensor(r, dtype=torch.long)

    hf = transformers.AutoTokenizer.from_pretrained(build_tokenizer_dir(tempfile.mkdtemp(), cfg), padding_side="right", use_fast=False)
    hf.pad_token = harness.PAD_TOKEN
    harness.preprocess_tokenizer(hf, 4)
    out = {}
    for tname, tok in (("stub", StubTokenizer(cfg)), ("hf", hf)):
        for llama31 in (1, 0):
            for m in (1, 2, 3, 4):
                agent = object.__new__(ag.InfiniSST)
                agent.args = types.SimpleNamespace(block_size=cfg.block_size)
                agent.latency_multiplier, agent.source_lang, agent.target_lang = m, "English", "German"
                agent.tokenizer, agent.llama31 = TensorTemplate(tok), bool(llama31)
                agent.model = types.SimpleNamespace(device=torch.device("cpu"))
                first = agent._prepare_inputs(types.SimpleNamespace(speech_cache=None))
                later = agent._prepare_inputs(types.SimpleNamespace(speech_cache=object()))
                key = f"{tname}_l31{llama31}_m{m}"
                out[key + "_first"] = first[0].numpy()
                out[key + "_later"] = later[0].numpy()
                out[key + "_sys"] = np.array(agent.system_prompt_size)
    np.savez_compressed(os.path.join(OUT, "prompts.npz"), **out)
    print("prompts.npz", len(out))


# ------------------------------------------------------------------ 11. secondary pins against transformers classes ----
def gen_hf_conv_extractor():
    """Secondary cross-check of the oracle's restated fairseq conv extractor (mode=layer_norm, conv_bias): transformers'
    Wav2Vec2LayerNormConvLayer stack (the HF port of the same fairseq module, `feat_extract_norm="layer"`), fp32 and bf16."""
    import transformers
    from transformers.models.wav2vec2.modeling_wav2vec2 import Wav2Vec2FeatureEncoder
    from transformers import Wav2Vec2Config
    from infinisst_amd import synth
    from infinisst_amd.config import toy_config
    from oracle.speech_encoder import ENC
    cfg = toy_config()
    hc = Wav2Vec2Config(feat_extract_norm="layer", conv_dim=[c for c, _, _ in cfg.conv_layers], conv_kernel=[k for _, k, _ in cfg.conv_layers],
                        conv_stride=[s for _, _, s in cfg.conv_layers], conv_bias=True, feat_extract_activation="gelu",
                        num_feat_extract_layers=len(cfg.conv_layers))
    out = {}
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        w = synth.random_weights(cfg, dtype=dtype, std=0.3, norm_jitter=0.1, seed=77)
        fe = Wav2Vec2FeatureEncoder(hc).eval()
        for i, layer in enumerate(fe.conv_layers):
            p = f"{ENC}feature_extractor.conv_layers.{i}."
            layer.conv.weight.data = w[p + "0.weight"].float().clone()
            layer.conv.bias.data = w[p + "0.bias"].float().clone()
            layer.layer_norm.weight.data = w[p + "2.1.weight"].float().clone()
            layer.layer_norm.bias.data = w[p + "2.1.bias"].float().clone()
        fe = fe.to(dtype)
        audio = torch.from_numpy(synth.synthetic_audio(cfg.first_chunk_offset + cfg.chunk_samples, stream_id=5)).unsqueeze(0).to(dtype)
        with torch.no_grad():
            y = fe(audio)  # (1, C, T)
        out[f"{tag}_audio"] = audio.float().numpy()
        out[f"{tag}_out"] = y.float().numpy()
    out["seed"] = np.array(77)
    np.savez_compressed(os.path.join(OUT, "hf_conv_extractor.npz"), **out)
    print("hf_conv_extractor.npz", len(out), "transformers", transformers.__version__)


def gen_hf_llama_model():
    """Secondary cross-check of the oracle's restated decoder composition (layer order, residuals, causal mask with and without a
    non-empty cache, final norm, lm_head): transformers' LlamaForCausalLM (llama3 rotary scaling) on a prefill, a chunked second
    prefill over the cache and a one-token decode step.  fp32 (composition) and bf16 (rounding points).  HF rotates K before
    caching and the reference's patch re-rotates an unrotated cache at 0..T-1 -- identical while nothing is evicted, which is the
    case here."""
    import transformers
    from transformers import LlamaConfig, LlamaForCausalLM
    from infinisst_amd import synth
    from infinisst_amd.config import toy_config
    cfg = toy_config()
    rope = {"rope_type": "llama3", "rope_theta": cfg.rope_theta, "factor": cfg.rope_factor, "low_freq_factor": cfg.rope_low_freq_factor,
            "high_freq_factor": cfg.rope_high_freq_factor, "original_max_position_embeddings": cfg.rope_original_max_pos}
    hc = LlamaConfig(hidden_size=cfg.llm_dim, intermediate_size=cfg.llm_ffn, num_hidden_layers=cfg.llm_layers,
                     num_attention_heads=cfg.llm_heads, num_key_value_heads=cfg.llm_kv_heads, head_dim=cfg.llm_head_dim, vocab_size=cfg.vocab,
                     rms_norm_eps=cfg.rms_eps, max_position_embeddings=131072, rope_parameters=rope, attention_bias=False,
                     mlp_bias=False, tie_word_embeddings=False, attn_implementation="eager")
    out = {}
    g = torch.Generator().manual_seed(11)
    ids = [torch.randint(3, 900, (1, n), generator=g) for n in (23, 9, 1)]
    for tag, dtype in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        w = synth.random_weights(cfg, dtype=dtype, std=0.05, norm_jitter=0.05, seed=31)
        model = LlamaForCausalLM(hc).eval()
        sd = {k: v.float() for k, v in w.items() if k.startswith("model.layers.") or k in ("model.embed_tokens.weight", "model.norm.weight", "lm_head.weight")}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all("rotary" in m or "inv_freq" in m for m in missing), (missing, unexpected)
        model = model.to(dtype)
        past = None
        for step, x in enumerate(ids):
            with torch.no_grad():
                r = model(input_ids=x, past_key_values=past, use_cache=True)
            past = r.past_key_values
            out[f"{tag}_ids_{step}"] = x[0].numpy()
            out[f"{tag}_logits_{step}"] = r.logits[0].float().numpy()
    out["seed"] = np.array(31)
    np.savez_compressed(os.path.join(OUT, "hf_llama_model.npz"), **out)
    print("hf_llama_model.npz", len(out), "transformers", transformers.__version__)


def main():
    torch.set_num_threads(4)
    import transformers.models.llama.modeling_llama  # noqa: F401  (before the wandb stub: accelerate probes it)
    install_fairseq_stubs()
    install_misc_stubs()
    import importlib
    pse = importlib.import_module("model.patches.patch_speech_encoder")
    msp = importlib.import_module("model.speech_encoder")
    gen_masks(pse)
    gen_encoder(pse, msp)
    gen_encoder_abs_pos(pse, msp)
    gen_shrink(msp)
    gen_llm_attention()
    gen_agent()
    gen_splice()
    gen_beam_scorer()
    gen_beam_loop()
    gen_logits_processors()
    gen_sampling_warpers()
    gen_llama_blocks()
    gen_prompts()
    gen_hf_conv_extractor()
    gen_hf_llama_model()


if __name__ == "__main__":
    main()
