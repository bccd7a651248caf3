/* C ABI of libinfinisst_hip.so -- the MI355X (gfx950) drop-in for the per-chunk compute behind the reference's
 * SimulEval agent (LeiLiLab/InfiniSST agents/infinisst.py policy()).
 *
 * The reference is pure Python and has no FFI; its lower boundary is the call
 *     outputs = self.model.generate(input_ids, speech_batch, past_key_values, states, multiplier, ...)
 * at agents/infinisst.py:307-332 plus the two state objects it mutates (states.speech_cache,
 * states.past_key_values) and the agent-side KV eviction at agents/infinisst.py:340-361.  Every entry point
 * below names the reference interface it replaces.  Plain C types only: pointers, sizes, int status codes
 * (0 = ok, negative = error, text via isst_last_error).  All device work is enqueued on the caller's
 * hipStream_t (passed as void*); functions that return host results synchronise that stream before returning.
 */
#ifndef INFINISST_HIP_H
#define INFINISST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISST_OK 0
#define ISST_ERR_ARG (-1)
#define ISST_ERR_HIP (-2)
#define ISST_ERR_STATE (-3)
#define ISST_ERR_NOMEM (-4)
#define ISST_ERR_NOTFOUND (-5)

#define ISST_MAX_CONV 8
#define ISST_MAX_SHRINK 4
#define ISST_MAX_EOS 8

typedef struct isst_handle isst_handle;

/* Model + runtime geometry.  Replaces the constructor arguments the reference spreads over
 * agents/infinisst.py:150-181 (load_model), agents/options.py:1-41 and model/speech_encoder.py:99-121. */
typedef struct isst_config {
    /* wav2vec2 conv feature extractor: (dim, kernel, stride) per layer, layer-norm mode */
    int n_conv;
    int conv_dim[ISST_MAX_CONV], conv_k[ISST_MAX_CONV], conv_stride[ISST_MAX_CONV];
    int conv_bias;
    /* streaming transformer encoder (head_dim must be 64) */
    int enc_dim, enc_layers, enc_heads, enc_ffn;
    float enc_ln_eps;
    int block_size;          /* --block-size: frames per block at multiplier 1 */
    int max_cache_size;      /* --max-cache-size: encoder KV sliding window (frames) */
    int enc_rope_round_each; /* 1: rotary products rounded to bf16 one by one (module cast to bf16); 0: one rounding */
    /* length shrink convs (no bias) + projector */
    int n_shrink;
    int shrink_dim[ISST_MAX_SHRINK], shrink_k[ISST_MAX_SHRINK], shrink_stride[ISST_MAX_SHRINK];
    /* Llama decoder (head_dim must be 128) */
    int llm_dim, llm_layers, llm_heads, llm_kv_heads, llm_ffn, vocab;
    float rms_eps;
    /* ids the speech splice looks for (reference model/llm.py:181-183) and EOS set */
    int user_id, assistant_id, start_header_id;
    int n_eos;
    int eos_ids[ISST_MAX_EOS];
    /* capacity */
    int max_streams;         /* concurrent streams sharing the weights */
    int max_multiplier;      /* largest latency multiplier (blocks per call) */
    int max_prompt_len;      /* longest prompt of one call (first chunk incl. system prompt) */
    int max_new_tokens;      /* upper bound of max_new_tokens */
    int max_llm_cache_size;  /* --max-llm-cache-size (ring sized for this + one chunk) */
    int max_system_prompt;   /* pinned system-prompt capacity (--always-cache-system-prompt) */
    int debug_taps;          /* keep copies of intermediate activations for isst_debug_tap */
    int max_beams;           /* largest num_beams of a generate call (1 = greedy only); KV arenas are allocated per beam */
    int enc_abs_pos;         /* --rope 0 (agents/options.py:37-41): q / k are not rotated (hand isst_set_rope_tables cos = 1, sin = 0), the bf16 sinusoid
                                of each frame's stream position is added to the encoder input instead (isst_set_enc_position_table) */
} isst_config;

/* generation arguments of one call; mirrors the keyword arguments at agents/infinisst.py:307-332 */
typedef struct isst_gen_params {
    int multiplier;                    /* multiplier= */
    int max_new_tokens;                /* max_new_tokens= */
    int no_repeat_ngram_size;          /* no_repeat_ngram_size= */
    int encoder_no_repeat_ngram_size;  /* encoder_no_repeat_ngram_size= */
    float repetition_penalty;          /* repetition_penalty= */
    const int* suppress_tokens;        /* suppress_tokens= (bad_words_ids), may be NULL */
    int n_suppress;
    int system_prompt_size;            /* >0: pin this many leading positions of a FRESH stream (keep-system-prompt) */
    int num_beams;                     /* num_beams= ; 0 or 1: greedy; >1: beam search (model/patches/patch_hf.py:687-967) */
    float length_penalty;              /* BeamSearchScorer length_penalty (HF default 1.0; 0 is read as 1.0) */
    int pcm_on_device;                 /* nonzero: pcm[i] are DEVICE pointers (fp32, n_samples each): the caller already holds the audio in
                                        * HBM.  0: host pointers, moved by the library as agents/infinisst.py:222 `.to(device)` does */
    /* the sample branch (agents/infinisst.py:311-315 -> patch_hf.py:606-624 -> HF _sample [3P]); with num_beams > 1: beam sample (patch_hf.py:871-875) --
     * max(2, 1 + n_eos) * num_beams draws WITHOUT replacement from the softmax over all beams' warped scores replace the top-k; draw j of step s uses the
     * uniform of step counter 64 s + j */
    int do_sample;                     /* do_sample= ; 0: greedy argmax */
    float temperature;                 /* temperature= (1.0 or <= 0: off) */
    int top_k;                         /* top_k= (0: off) */
    float top_p;                       /* top_p= (>= 1.0: off) */
    float epsilon_cutoff;              /* epsilon_cutoff= (0: off) */
    unsigned long long seed;           /* the draw of (stream, chunk, step) is isst_op_sample_uniform(seed, stream id, chunks so far, step) */
} isst_gen_params;

typedef struct isst_stream_info {
    int llm_cache_len;   /* = states.past_key_values[0][0].size(2) */
    int llm_sys_len;     /* pinned prefix inside llm_cache_len */
    int enc_n_steps;     /* = states.speech_cache.n_steps */
    int enc_cache_len;   /* = states.speech_cache.layers[i].k.size(1) */
    int chunks;
} isst_stream_info;

/* ---- lifetime (replaces InfiniSST.load_model, agents/infinisst.py:130-183) ---- */
int isst_create(const isst_config* cfg, isst_handle** out);
void isst_destroy(isst_handle* h);
const char* isst_last_error(isst_handle* h); /* h may be NULL: last create error */

/* One tensor of the reference checkpoint (`pytorch_model.bin` keys, agents/infinisst.py:179-180), bf16 bits,
 * row-major, host or device pointer.  The library copies and re-lays it out (MFMA-fragment-major tiles). */
int isst_load_weight(isst_handle* h, const char* name, const void* data, int ndim, const int64_t* shape, int on_device);
/* Rotary tables computed by the host with the third-party semantics it wants to reproduce:
 * encoder (rotary_embedding_torch, patch_speech_encoder.py:631,:824): fp32 cos/sin [enc_rows][32];
 * LLM (HF LlamaRotaryEmbedding llama3, patch_llm.py:290-299): bf16 cos/sin [llm_rows][64] (first half of the dims). */
int isst_set_rope_tables(isst_handle* h, const float* enc_cos, const float* enc_sin, int enc_rows, const uint16_t* llm_cos,
                         const uint16_t* llm_sin, int llm_rows);
/* cfg.enc_abs_pos only (--rope 0): the sinusoid the reference adds to the encoder input (sinusoidal_positional_embedding,
 * patch_speech_encoder.py:448-461), bf16 bits [rows][enc_dim].  The reference holds the positions themselves in bf16, so the table has one row per
 * bf16 integer: rows 0..255 = positions 0..255; row 256 + i = the position whose bf16 bit pattern is 0x4380 + i (256, 258, ..., 512, 516, ...).
 * ISST_ENC_POS_ROWS rows reach position 2^24 (93 hours of audio); a stream that runs past the last row given fails loudly. */
#define ISST_ENC_POS_ROWS 2305
int isst_set_enc_position_table(isst_handle* h, const uint16_t* table, int rows);
int isst_finalize_weights(isst_handle* h); /* fails listing the first missing tensor; needs the tables above */

/* ---- per-utterance state (replaces S2TAgentStates.reset / build_states, agents/infinisst.py:50-67,115-123) ---- */
int isst_stream_open(isst_handle* h, int* stream_id);
int isst_stream_reset(isst_handle* h, int stream_id);
int isst_stream_close(isst_handle* h, int stream_id);
int isst_stream_info_get(isst_handle* h, int stream_id, isst_stream_info* out);

/* State import: resume a stream from saved caches -- the library-side form of handing a `states.speech_cache` /
 * `states.past_key_values` back to the model (agents/infinisst.py:50-67, :334-336).  Tests and bench.py also use it to put a stream
 * into its steady state (KV ~ max_llm_cache_size, encoder window saturated, rings about to wrap) without running 40 chunks.
 * Host pointers, bf16 bits.  One call per layer; every layer must be given the same total / sys_len / ring_start (resp. len / n_steps / ring_start).
 *   llm:  k, v [kv_heads][total][128], UNROTATED keys in logical order (= past_key_values[layer][0/1][0]); the first sys_len entries go to the
 *         pinned region; ring_start = physical ring slot logical position sys_len is written to (any value in [0, ring capacity));
 *   enc:  k, v [heads][len][64] (= speech_cache.layers[layer].k / .v), n_steps = speech_cache.n_steps (frames consumed so far);
 *   audio history: the last (receptive field - 1 = 399) samples consumed, bf16 (= the tail of speech_cache.src). */
int isst_stream_import_llm_kv(isst_handle* h, int stream_id, int layer, const uint16_t* k, const uint16_t* v, int total, int sys_len, int ring_start);
int isst_stream_import_enc_kv(isst_handle* h, int stream_id, int layer, const uint16_t* k, const uint16_t* v, int len, int n_steps, int ring_start);
int isst_stream_import_audio_history(isst_handle* h, int stream_id, const uint16_t* samples, int n);

/* ---- the hot path: model.generate(...) for n streams (agents/infinisst.py:307-332) ----
 * pcm[i]: n_samples new fp32 samples of stream i as prepared by _prepare_speech (zero-padded to a multiple of
 *   block_size/4*1280 samples; WITHOUT the 399-sample first-chunk offset: the library keeps that history itself);
 *   n_samples is the same for all streams of a call.
 * prompt_ids[i]/prompt_lens[i]: this chunk's prompt (agents/infinisst.py:225-268).
 * prev_target_ids[i]/n_prev[i]: encoder_input_ids = last <=lookback target ids (:298-301).
 * forced_tokens[i] (may be NULL): teacher forcing for tests -- token k replaces the argmax of step k.
 * out_ids[i] receives the generated ids = outputs.sequences[0, len(prompt):] (up to max_new_tokens),
 * out_lens[i] their count.  logits_out (may be NULL): raw fp32 last-position logits, [n][max_new_tokens][vocab]. */
int isst_generate(isst_handle* h, const isst_gen_params* p, int n, const int* stream_ids, const float* const* pcm,
                  int n_samples, const int* const* prompt_ids, const int* prompt_lens, const int* const* prev_target_ids,
                  const int* n_prev, const int* const* forced_tokens, const int* n_forced, int* const* out_ids, int* out_lens,
                  float* logits_out, void* hip_stream);

/* LLM-KV eviction of one stream (agents/infinisst.py:354-361): keep the first keep_prefix entries and the last
 * new_cache_size entries; the rest is dropped and the remaining keys re-index (positions 0..T-1). */
int isst_kv_evict(isst_handle* h, int stream_id, int new_cache_size, int keep_prefix);

/* Encoder only (speech_encoder.encode_speech, model/speech_encoder.py:219-236) for one stream: advances the
 * stream's speech cache and returns the projected features [S][llm_dim] as bf16 bits.  Test aid. */
int isst_encode_speech(isst_handle* h, int stream_id, const float* pcm, int n_samples, int multiplier, uint16_t* out_features,
                       int* out_rows, void* hip_stream);

/* copy of an intermediate activation of the last call (needs cfg.debug_taps): names "conv_out", "post_proj",
 * "enc_layer_<i>", "enc_out", "shrink", "speech", "llm_embed", "llm_layer_<i>", "llm_final". bf16 bits. */
int isst_debug_tap(isst_handle* h, const char* name, uint16_t* dst, int64_t max_elems, int64_t* got_elems);
/* In-situ timing of the dominant kernel for the roofline (bench.py): between _begin and _end every one-token gate/up GEMV
 * of isst_generate is bracketed by a HIP event pair on the caller's stream; _end synchronises `hip_stream` and returns the
 * average bracket in microseconds (kernel execution + the dispatch latency of that launch) and the number of launches. */
int isst_profile_begin(isst_handle* h);
/* the same bracket around the gate/up launch of every Llama pass of rows_lo..rows_hi rows (isst_profile_begin = 1..1): the prefill pass of a many-stream
 * step is the widest dense contraction of the path, and back-to-back launches of it alone run at lower clocks than it does inside a step */
int isst_profile_begin_rows(isst_handle* h, int rows_lo, int rows_hi);
int isst_profile_end(isst_handle* h, void* hip_stream, double* avg_us, int64_t* launches);

/* unrotated K and V (128 bf16 each) of logical position `pos`, kv head `kv_head`, layer `layer` in the arena of beam
 * `beam` of a stream (beam 0 for greedy streams).  Test aid for the KV ring / beam bookkeeping. */
int isst_debug_read_kv(isst_handle* h, int stream_id, int beam, int layer, int kv_head, int pos, uint16_t* k_out, uint16_t* v_out);

/* Beam-search test aid for ONE-stream calls: between _begin and _end every beam step of isst_generate records, per beam row, the device's
 * top `n_keep = max(2, 1 + n_eos) * num_beams` processed log-probs (values and token ids, before the beam score is added;
 * patch_hf.py:833-878) and the beam scores; with n_steps > 0 the search continues along the caller's (token, parent) choices
 * forced_tokens / forced_parents [n_steps][num_beams] instead of its own (teacher forcing against oracle/beam.py). */
int isst_debug_beam_trace_begin(isst_handle* h, int num_beams, const int* forced_tokens, const int* forced_parents, int n_steps);
int isst_debug_beam_trace_step(isst_handle* h, int step, int* rows, int* n_keep, float* top_val, int* top_idx, float* beam_scores, int max_elems);
int isst_debug_beam_trace_end(isst_handle* h, int* n_steps);

/* ---- per-kernel entry points (parity tests and micro-benchmarks); all pointers are DEVICE pointers ---- */
/* Host half of the splice (model/llm.py:86-113): the (user, assistant) header pairs of a prompt as a row map with the reference's slice
 * semantics -- row_src[t] >= 0: prompt token row_src[t]; < 0: speech feature -1 - row_src[t]; *n_rows <= len rows come out (fewer than len
 * when the encoder produced fewer features than the prompt has patch slots; surplus features are dropped).  Host pointers, row_src holds len ints. */
int isst_op_splice_map(const int* ids, int len, int user_id, int assistant_id, int start_header_id, int n_features, int* row_src, int* n_rows);
/* out[r] = speech_row[r] >= 0 ? speech[speech_row[r]] : table[ids[r]]   (rows of D bf16): the splice of model/llm.py:86-113 once the
 * host has turned the header pairs into a row map (speech_row may be NULL: plain embedding lookup). */
int isst_op_embed_splice(const int* ids, const int* speech_row, const uint16_t* table, const uint16_t* speech, uint16_t* out, int rows, int D,
                         void* hip_stream);
/* uni_mha_forward's attention core (patch_speech_encoder.py:797-915) for one stream and one layer: qkv [Q][3 * heads * 64] (q | k | v rows
 * after the projections, q NOT yet scaled), K ring [heads][cap][64] / V ring [heads][64][cap] holding min(prefix, max_cache) cached keys from
 * physical slot ring_start on; appends the Q new keys / values to the rings, returns out [Q][heads * 64].  rope tables fp32 [cap][32]. */
int isst_op_enc_attention(const uint16_t* qkv, uint16_t* kring, uint16_t* vring, int ring_start, int prefix, const float* rope_cos,
                          const float* rope_sin, int rope_round_each, uint16_t* out, int Q, int heads, int cap, int max_cache, int blocksize,
                          void* hip_stream);
/* llama_sdpa_attention_new_forward's core (patch_llm.py:258-332) for one stream and one layer: qkv [rows][(heads + 2 kv_heads) * 128] of `rows`
 * consecutive positions pos0 .. pos0+rows-1 over an arena K / rotated-K scratch / V, each [kv_heads][sys_cap + ring_cap][128], that holds pos0
 * cached entries (unrotated keys; logical p < sys_len in slot p, else sys_cap + (ring_start + p - sys_len) mod ring_cap); appends the rows' own
 * k / v, returns out [rows][heads * 128].  rot_keys: fill and use the rotated-key scratch (the library's default) or rotate on read.
 * rope tables bf16 [>= pos0 + rows][64].  Runs the prefill kernel for rows > 16 / (heads / kv_heads), the decode kernel + combine otherwise. */
int isst_op_llm_attention(const uint16_t* qkv, int rows, int pos0, uint16_t* kpool, uint16_t* krpool, uint16_t* vpool, int heads, int kv_heads,
                          int sys_cap, int ring_cap, int sys_len, int ring_start, const uint16_t* rope_cos, const uint16_t* rope_sin, int rot_keys,
                          uint16_t* out, void* hip_stream);
/* W [n_rows][K] row-major bf16 (conv_k > 0: Conv1d weight [n_rows][K/conv_k][conv_k]) -> packed tiles */
int isst_op_pack_weight(const uint16_t* w, uint16_t* packed, int n_rows, int K, int conv_k, void* hip_stream);
int64_t isst_op_packed_elems(int n_rows, int K);
/* gate_proj and up_proj ([ffn][K] each, ffn % 8 == 0) as SELF-PAIRED tiles: tile t = gate rows 8t..8t+7 | up rows 8t..8t+7, 2 * ffn packed rows in all; isst_op_gemm with
 * epi 8 (SwiGLU8, <= 16 rows) then gives out[8t + c] = silu(gate) * up with HF LlamaMLP's rounding points [3P] -- the bits of epi 5 on tile pairs.  What the library streams for
 * the one-row decode passes of reference model/llm.py:114-126 -> LlamaMLP (engine_llm.hip llm_forward). */
int isst_op_pack_gateup8(const uint16_t* gate, const uint16_t* up, uint16_t* packed, int ffn, int K, void* hip_stream);
/* out = epi(A @ W^T); epi: 0 none, 1 bias, 2 bias+gelu, 3 residual, 4 bias+residual, 5 swiglu (packed rows
 * alternate gate/up tiles), 6 fp32 out, 8 swiglu on self-paired tiles (isst_op_pack_gateup8; <= 16 rows).  Replaces torch F.linear / F.conv1d call sites (see gemm.hip).
 * norm_w != NULL (epi 0, 5, 6 only): LlamaRMSNorm(norm_w, norm_eps) is applied to the rows of A on load. */
int isst_op_gemm(const uint16_t* A, int64_t lda, const uint16_t* packed, const uint16_t* bias, const uint16_t* res,
                 int64_t ldres, void* out, int64_t ldo, int M, int N, int K, int n_valid, int epi, const uint16_t* norm_w,
                 float norm_eps, void* hip_stream);
/* the pair the decoder runs at 17..64 rows for o_proj / down_proj (HF LlamaDecoderLayer: residual + mlp/attn output, then the
 * next RMSNorm): slabs[ksplit][M][N] (fp32, caller scratch) = A @ W^T per K slice; x[M][N] = bf16(x + bf16(sum of slabs)) in place;
 * norm_w != NULL: out = LlamaRMSNorm(norm_w, norm_eps)(x).  N % 16 == 0; 16 < M <= 64: K % (256 * ksplit) == 0; M > 64 (dense kernel):
 * K % (64 * ksplit) == 0. */
int isst_op_gemm_splitk_rmsnorm(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* x, const uint16_t* norm_w,
                                uint16_t* out, float* slabs, int M, int N, int K, int ksplit, float norm_eps, void* hip_stream);
/* the same pair WITHOUT a launch between the two projections (13..64 rows, N % 32 == 0, K % 128 == 0 for the consumer):
 * isst_op_gemm_splitk_fused: slabs as above; the K-slice workgroup of a column block that arrives last sums the slabs and writes
 *   x = bf16(x + bf16(sum)) in place (the same bits as isst_op_gemm_splitk_rmsnorm's x) plus, per row and 32 columns, the sum of squares of
 *   the new x into ssq[M][N / 32]; tickets: N / 16 ints, zero before the first call (the kernel re-arms them);
 * isst_op_gemm_norm_ssq: out = epi(LlamaRMSNorm(norm_w, eps)(x) @ W^T) with the rows normalised while they are staged, 1/rms from ssq[M][K / 32]
 *   (epi: none, swiglu or f32). */
int isst_op_gemm_splitk_fused(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* x, float* slabs, float* ssq, int* tickets,
                              int M, int N, int K, int ksplit, void* hip_stream);
/* K slices with the same in-launch reduction and NO residual: out[M][N] = bf16(sum of the slabs) (q/k/v at 33..64 rows); norm_w != NULL: A = x is
 * normalised while staged as in isst_op_gemm_norm_ssq (ssq_in[M][K / 32]). */
int isst_op_gemm_splitk_plain(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* out, int64_t ldo, float* slabs, int* tickets,
                              int M, int N, int K, int ksplit, const uint16_t* norm_w, float norm_eps, float* ssq_in, void* hip_stream);
int isst_op_gemm_norm_ssq(const uint16_t* x, int64_t ldx, const uint16_t* packed, void* out, int64_t ldo, int M, int N, int K, int n_valid,
                          int epi, const uint16_t* norm_w, float norm_eps, float* ssq, void* hip_stream);
/* the encoder twin (wav2vec2 TransformerSentenceEncoderLayer: x = residual + Linear(.) with bias, then LayerNorm):
 * x = bf16(x + bf16(sum of slabs + bias)) in place; ln_w != NULL: out = LayerNorm(ln_w, ln_b, eps)(x). */
int isst_op_gemm_splitk_layernorm(const uint16_t* A, int64_t lda, const uint16_t* packed, const uint16_t* bias, uint16_t* x,
                                  const uint16_t* ln_w, const uint16_t* ln_b, uint16_t* out, float* slabs, int M, int N, int K,
                                  int ksplit, float eps, void* hip_stream);
/* profiling aid: override the GEMM launch heuristic (waves per workgroup, 16-row n-tiles per workgroup); 0 = automatic */
int isst_op_set_gemm_tuning(int waves_per_block, int ntiles_per_block);
/* profiling / test aid: workgroups the decoder attention wants chip-wide before a workgroup's slot span grows beyond 64; 0 = default.  1 = one span per
 * (stream, kv head), which also folds a shared-prefix beam group's per-beam keys into that workgroup (what 16+ streams x beams select by themselves);
 * n < 0 = the target -n - 1 with the beams' per-beam workgroups kept at any stream count (A/B runs).  Bits 16.. : the same target for the prefill form. */
int isst_op_set_attn_tuning(int target_workgroups);
/* test aid: row limits of the all-loads-first forms of the two reduce launches (csrc/rowops.hip rmsnorm_reduce_lf_kernel, layernorm_reduce_lf_kernel: the residual +
 * RMSNorm / LayerNorm passes that sum a projection's K slices); 0 = the stepwise kernels everywhere, -1 = the default (ISST_RMS_LF_ROWS / ISST_LN_LF_ROWS, else by
 * slice count).  Both forms carry the same bits: the tests run one against the other through this switch. */
int isst_op_set_reduce_tuning(int rms_lf_rows, int ln_lf_rows);
/* test entry: the beam search's per-row top-k over the processed log-probs (csrc/beam.hip; replaces torch.topk(next_token_scores, ...) of the reference's
 * _beam_search, patch_hf.py:863-879): scores [rows][ld] fp32 on the device, out_val / out_idx [rows][32]; ties go to the lowest index */
int isst_op_topk_rows(const float* scores, long ld, int vocab, int k, int rows, float* out_val, int* out_idx, void* hip_stream);
int isst_op_layernorm(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* out, int rows, int C, float eps,
                      int gelu, void* hip_stream);
int isst_op_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* out, int rows, int D, float eps, void* hip_stream);
int isst_op_conv0(const uint16_t* audio, const uint16_t* w, const uint16_t* bias, const uint16_t* ln_w, const uint16_t* ln_b,
                  uint16_t* out, int T, int C, int k, int stride, void* hip_stream);
/* the sample branch's host half, piece by piece (no GPU needed): HF's warpers Temperature -> TopK -> TopP -> Epsilon on one row of processed fp32
 * scores (warped in place, -inf = removed) and the inverse-CDF draw at u in [0, 1); the uniform of a (stream, chunk, step) */
int isst_op_warp_sample(float* scores, int vocab, float temperature, int top_k, float top_p, float epsilon_cutoff, double u, int* token);
double isst_op_sample_uniform(unsigned long long seed, int stream, int chunk, int step);
/* the warpers alone; min_tokens_to_keep = 1 in the sample branch, n_eos + 1 under beam search (HF `_get_logits_processor`) */
int isst_op_warp(float* scores, int vocab, float temperature, int top_k, float top_p, float epsilon_cutoff, int min_tokens_to_keep);
/* torch.multinomial(softmax(scores), k) without replacement as k sequential inverse-CDF draws (index order) at the given uniforms: fp32 softmax, fp64 running sums;
 * picked[j] = flat index of draw j; ISST_ERR_STATE when fewer than k entries have non-zero probability (torch raises there) */
int isst_op_multinomial_wor(const float* scores, long n, int k, const double* uniforms, long* picked);
/* logits [vocab] fp32 (modified in place) -> *out_token (device int) */
int isst_op_sample(float* logits, int vocab, const int* ids, int n_ids, const int* enc_ids, int n_enc, const int* suppress,
                   int n_suppress, float repetition_penalty, int ngram, int enc_ngram, int* out_token, void* hip_stream);

/* Merge of the decode attention's split-KV partials [rows][heads][n_splits][ISST_ATTN_SLAB] fp32, one slab per slot split:
 * floats 0..127 = unnormalised O, 128 = running max, 129 = sum, 130..131 = padding (528 B: 33 whole 16-byte stores)
 * (the flash-decoding reduction behind patch_llm.py:320-329's single softmax): as a pass of its own (-> out [rows][heads * 128] bf16) and
 * fused into the o_proj projection that follows (patch_llm.py:334): out = (res +) merged @ W_o^T for M <= 2 rows, K = heads * 128. */
#define ISST_ATTN_SLAB 132
int isst_op_attn_combine(const float* partial, uint16_t* out, int heads, int rows, int n_splits, void* hip_stream);
int isst_op_gemm_attn_merge(const float* partial, int n_splits, const uint16_t* packed, const uint16_t* res, int64_t ldres, uint16_t* out,
                            int64_t ldo, int M, int N, int K, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* INFINISST_HIP_H */
