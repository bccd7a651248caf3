// Launchers of the non-GEMM kernels (all asynchronous on `stream`, return ISST_* codes).
#pragma once
#include "common.h"

// per-stream view of the encoder KV ring for one chunk (same for all layers)
struct EncStreamView {
    int start;   // physical ring slot of logical key 0 (after trimming to max_cache_size)
    int prefix;  // frames consumed before this chunk (cache.n_steps); cached keys = min(prefix, max_cache)
};

// per-stream view of the LLM KV arena: [sys region: sys_cap slots][ring: ring_cap slots]
struct LlmStreamView {
    int sys_len;     // pinned system-prompt entries (logical positions 0..sys_len-1 live in the sys region)
    int ring_start;  // physical ring slot of logical position sys_len
    long kv_offset;  // element offset of this stream's arena inside the K (and V) pool, per layer-0 head-0 base
    int new_start;   // first logical position written by THIS launch (keys >= new_start are read from the qkv rows)
    int row0;        // row of this launch that holds position new_start
    int rot_keys;    // 1: keys older than this launch are read ALREADY ROTATED from the rotated-key arena (filled for this chunk) and the launch's own
                     // keys are added to it; 0: rotate on read, as the reference does every pass; 2 (prefill launches only, llm_attn_prefill_kernel):
                     // rotate on read AND store the rotated keys to the arena -- this launch is what fills it (no launch_llm_rope_cache pre-pass);
                     // every other kernel reads 2 as 1
    // beam search, shared-prefix form (n_beams > 1): the group's rows are the B beams of ONE stream at the same position.  Their arenas
    // (kv_offset + b * beam_stride) are identical below logical position tail_start, so the ordinary slot-split workgroups read those
    // keys ONCE, from arena 0, for all beams' columns; keys >= tail_start (written during this chunk: per beam) and the step's own key
    // (row row0 + b) belong to B extra workgroups, one per beam, which only that beam's columns listen to
    int n_beams;
    int tail_start;
    long beam_stride;
};

int launch_enc_add_position(bf16_t* x, const EncStreamView* ev, const bf16_t* table, int table_rows, int n, int Q, int D, hipStream_t s);
int launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t s);
int launch_audio_window(const float* pcm, const float* const* ptrs, const int* sids, const bf16_t* hist_pool, long histp, bf16_t* window, long winp,
                        int hist, int n_samples, int n, hipStream_t s);
int launch_audio_hist_save(const bf16_t* window, long winp, const int* sids, bf16_t* hist_pool, long histp, int hist, int win, int n, hipStream_t s);
int launch_conv0(const bf16_t* audio, long audio_batch, const bf16_t* w, const bf16_t* bias, const bf16_t* ln_w,
                 const bf16_t* ln_b, bf16_t* out, long out_batch, int T, int C, int k, int stride, int batch, hipStream_t s);
int launch_layernorm(const bf16_t* x, long ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, long ldo, int rows, int C,
                     float eps, int gelu, hipStream_t s);
// split-K epilogue of an encoder projection + the LayerNorm that follows: x += bf16(sum of fp32 slabs + bias) in place, out = LN(x)
int launch_layernorm_reduce(const float* slabs, long slab_stride, int n_slabs, const bf16_t* proj_bias, bf16_t* x, long ldx, const bf16_t* w,
                            const bf16_t* b, bf16_t* out, long ldo, int rows, int C, float eps, hipStream_t s);
int launch_rmsnorm(const bf16_t* x, long ldx, const int* rows_idx, const bf16_t* w, bf16_t* out, long ldo, int rows, int D,
                   float eps, hipStream_t s);
// split-K slabs (gemm_mid.hip EPI_PARTIAL) -> x += sum, then RMSNorm of the updated rows (w == null: update only)
// out = bf16(sum of fp32 slabs [n_slabs][rows][N])
void reduce_set_tuning(int rms_lf_rows, int ln_lf_rows);  // (test aid: isst_op_set_reduce_tuning)
int launch_slab_reduce(const float* slabs, long slab_stride, int n_slabs, bf16_t* out, long ldo, int rows, int N, hipStream_t s);
int launch_rmsnorm_reduce(const float* slabs, long slab_stride, int n_slabs, bf16_t* x, long ldx, const bf16_t* w, bf16_t* out, long ldo,
                          int rows, int D, float eps, hipStream_t s);
int launch_embed_splice(const int* ids, const int* speech_row, const bf16_t* table, const bf16_t* speech, bf16_t* out,
                        int rows, int D, hipStream_t s);

// ---- encoder attention (enc_attn.hip) ----
// qkv: [n_streams*Q][3*D] (q | k | v).  K/V rings: per stream `stream_stride` elements, layout [heads][cap][64].
// also appends the chunk's own k / v (from the qkv rows) to the rings: no separate append launch is needed
int launch_enc_attention(const bf16_t* qkv, bf16_t* kring, bf16_t* vring, long stream_stride,
                         const EncStreamView* sv, const float* rope_cos, const float* rope_sin, int rope_round_each,
                         bf16_t* out, int n_streams, int Q, int heads, int cap, int max_cache, int blocksize, hipStream_t s,
                         const bf16_t* rope_cs = nullptr);  // rope_cs: the packed bf16 form of both tables (enc_attn.hip EncTab; null: the fp32 tables are read)

// ---- LLM attention (llm_attn.hip) ----
struct LlmAttnDims {
    int heads, kv_heads;         // head_dim is 128
    int sys_cap, ring_cap;       // arena geometry (slots); per (layer, kv head) the arena is [sys_cap + ring_cap][128]
    long layer_stride;           // elements between layers inside one stream's arena = kv_heads*(sys_cap+ring_cap)*128
};
// causal attention of every row over its stream's keys 0..row_pos, fused with the q rotation and the append of
// the row's own (unrotated) k, v to the arena; RoPE applied to K on read.  qkv: [rows][(H + 2 KV) * 128].
// kpool: K [slots][128] per (stream, layer, kv head); vtpool: V, same layout (row per key).  groups[z] = (first row,
// row count): runs of consecutive rows of ONE stream, at most LLM_ATTN_GROUP_ROWS(G) rows each.
// partial: [rows][heads][slots/64][2 + 128] fp32.
#define LLM_ATTN_GROUP_ROWS(G) (16 / (G))
#define LLM_PREFILL_UNIT_GROUPS 6  // row groups of one stream a prefill workgroup serves (llm_attn.hip PREFILL_MAX_GROUPS: six consumer waves + two key loader waves)
void llm_attn_set_tuning(int target_wgs);  // workgroups wanted before slot spans grow beyond 64 (0 = default)
// Launch metadata of a ONE-group launch (one stream's decode step or short prefill) passed by value in the kernel arguments:
// the workgroups then start their key loads at once instead of after three dependent loads (groups -> row_stream/row_pos ->
// stream view); with one stream the attention kernel is nothing but such a chain of round trips.
struct LlmAttnOne {
    int enabled;
    int2 grp;        // (first row, row count)
    int pos0;        // position of the first row
    int pos_step;    // row k of the group sits at pos0 + k * pos_step: 1 = consecutive positions (a prompt, one decode row), 0 = all rows at pos0 (the beams of a stream)
    LlmStreamView v;
};
int launch_llm_attention(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv, const int2* groups,
                         int n_groups, int max_group_rows, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool,
                         float* partial, bf16_t* out, LlmAttnDims d, int layer, int rows, hipStream_t s, const LlmAttnOne* one = nullptr,
                         const int2* units = nullptr, int n_units = 0, int max_unit_groups = 0, int n_beam_wgs = 0, int* defer_combine = nullptr,
                         int* arrive_counters = nullptr);
// one stream's decode step (one row or <= 4 shared-prefix beams), attention + combine + o_proj + residual in one launch: > 0 (the prefix's slot splits) when covered
int llm_attn_oproj_supported(const LlmAttnDims& d, int rows, int n_groups, const LlmAttnOne* one, int N, int K, int n_cus, int n_beam_wgs);
int launch_llm_attn_oproj(const bf16_t* qkv, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool, float* partial,
                          LlmAttnDims d, int layer, const LlmAttnOne& one, int n_beam_wgs, const bf16_t* Wp, int N, int K, int n_valid, const bf16_t* res,
                          bf16_t* out, int ld, bf16_t* attn_row, unsigned* trow, unsigned* bar, int* err, int n_cus, hipStream_t s, unsigned* arrive_total,
                          unsigned* merge_total, int mode = 0, int delay = 0, unsigned arrive_bias = 0);
                         // arrive_counters != null (>= kv_heads zeroed ints): a ONE-group launch combines its splits itself (last-arriver form, llm_attn.hip)
                         // defer_combine != null: more than one slot split -> NO combine launch, *defer_combine = the split count and `partial` holds the
                         // (max, sum, O) slabs for the consumer to merge (GemmArgs::attn_partial); one split -> *defer_combine = 0, `out` is written  // units[z] = (first group, groups <= 8) of ONE
                                                                                              // stream: prefill launches share key tiles per unit
// the combine pass alone: partial [rows][heads][n_splits][2 + 128] fp32 -> out [rows][heads * 128] bf16
int launch_llm_attn_combine(const float* partial, bf16_t* out, int heads, int rows, int n_splits, hipStream_t s);
// Rotated-key arena for one chunk: for every listed stream and EVERY layer, krpool[slot] = RoPE(kpool[slot], logical position of the slot)
// for the `total` cached keys (views[i].new_start = total).  A key's logical position only changes when the host evicts, i.e. between
// chunks, so the 10 passes of a chunk (and the row groups of its prefill, which would each rotate the same keys again) read keys
// rotated once.  Identical bits to rotating on read.
int launch_llm_rope_cache(const LlmStreamView* sv, int n_streams, const bf16_t* rope_cos, const bf16_t* rope_sin, const bf16_t* kpool, bf16_t* krpool,
                          LlmAttnDims d, int layers, hipStream_t s);

// ---- sampling (sample.hip) ----
struct SampleStream {
    int n_ids;        // prompt + generated so far (this chunk)
    int n_enc;        // previous target ids (<= lookback)
    int ids_off;      // offsets into the ids / enc_ids pools
    int enc_off;
    int logits_row;   // row of the logits buffer
};
int launch_sample(float* logits, long ld_logits, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool,
                  const int* suppress, int n_suppress, float rep_penalty, int ngram, int enc_ngram, int* out_tokens,
                  float* scratch_val, int* scratch_idx /* 64 per stream each */, int n_streams, hipStream_t s);

// ---- beam search pieces (beam.hip) ----
#define BEAM_TOPK 32
struct KvCopyOp {
    long arena_offset;  // element offset of the arena (stream, beam) inside the K / V^T pools
    long buf_offset;    // element offset of the buffer inside kbuf / vbuf
    int p0, count;      // logical positions p0 .. p0+count-1
    int sys_len, ring_start;
    int to_arena;       // 0: arena -> buffer, 1: buffer -> arena
    int pad;
};
int launch_kv_positions_copy(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const KvCopyOp* ops, int n_ops, int max_count,
                             LlmAttnDims d, int layers, int tcap, hipStream_t s);
// the same copies from op lists that a kernel wrote (beam_select_kernel): `*count` ops, known only on the device; grid (kv heads, layers, op lanes)
int launch_kv_positions_copy_list(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const KvCopyOp* ops, const int* count,
                                  int max_ops, LlmAttnDims d, int layers, int tcap, hipStream_t s);
int launch_log_softmax(float* logits, long ld, int vocab, float* pmax, float* psum, int rows, hipStream_t s);

// the score rows of a beam step as beam_process_kernel leaves them (beam.hip): raw logits + log Z per row + the penalised entries in a side list
struct BeamScoreView {
    float* logz;       // [rows]
    int* side_tok;     // [rows][side_cap]
    float* side_val;
    int* side_n;       // [rows]
    int side_cap;
};
int launch_beam_scores(float* logits, long ld, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress, int n_suppress,
                       float rep_penalty, int ngram, int enc_ngram, float* pmax, float* psum, const BeamScoreView& v, int k, float* cval, int* cidx,
                       float* out_val, int* out_idx, int rows, hipStream_t s);

// ---- the beam scorer on the device (beam.hip beam_select_kernel; reference patch_hf.py:43-157 beam_search_process + :278-302 BeamHypotheses.add + [3P]
//      is_done + the cache reorder of :910-913) ----
#define BEAM_MAX_B 8
struct BeamDevStream {           // one per stream of a beam call; written by the host before the loop, kept up to date by the kernel
    float score[BEAM_MAX_B];     // beam scores (running sums of log-probs)
    double hyp_score[BEAM_MAX_B + 1];  // BeamHypotheses: score of every kept hypothesis, in insertion order (the order decides ties)
    int hyp_buf[BEAM_MAX_B + 1]; // tail buffer slot holding its KV, or < 0
    int hyp_n;
    int done;
    double worst;
    int free_bufs[2 * BEAM_MAX_B + 1];
    int n_free;
    int prompt_len;              // tokens of the prompt (the processors' ids: patch tokens included)
    int P0;                      // first logical KV position written by the decode phase
    int sid;                     // library stream id (arena / buffer offsets)
    int sys_len, ring_start;
    int n_enc;
    int pad;
};
struct BeamDecision {            // per (stream, beam) and step, in the pinned host log
    int tok, par;
    float score;
    float forced_lp;
};
// What the device scorer decided about one stream's tails at one step (kv_beam_reorder_kernel): new beam b continues par[b]; the hypotheses the step closed keep a
// copy of their (old) beam's tail in a buffer.  A record with n_move == n_save == 0 moves nothing (a finished stream, a step that only extended every beam).
struct BeamReorder {
    long arena0;              // element offset of the stream's arena 0 inside the pools (beam b: + b * stream_stride)
    long buf0;                // element offset of the stream's buffer 0 (buffer k: + k * tbuf_stride)
    int p0, count;            // the tail: logical positions p0 .. p0 + count - 1
    int sys_len, ring_start;
    int par[BEAM_MAX_B];
    int save_beam[BEAM_MAX_B], save_buf[BEAM_MAX_B];
    int n_save, n_move;
};
// ONE launch per beam step for every copy the step asks for (the op-list form stages parents through temporaries: two launches of 57 us and twice the bytes at 64
// streams x 4 beams).  Thread = one 16-byte chunk of one tail position of one (kv head, layer, stream): it loads that chunk from every arena it needs -- hypothesis
// saves first, then each moving beam's parent --, waits for all of them, and only then stores; chunks are disjoint between threads, so no copy can read what another
// has already overwritten.
int launch_kv_beam_reorder(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const BeamReorder* recs, int n_streams, int B,
                           long stream_stride, long tbuf_stride, LlmAttnDims d, int layers, int tcap, hipStream_t s);
struct BeamSelArgs {
    int n, B, n_keep, V, step, rows_per, max_ids, max_enc_ids;
    int n_eos, pad_tok;
    int eos[8];
    const float* top_val;        // [rows][BEAM_TOPK] candidates of this step (processed log-probs, before the beam score is added)
    const int* top_idx;
    BeamDevStream* st;           // [n]
    const double* powtab;        // powtab[l] = pow((double)l, length_penalty), filled by the HOST (one libm for both scorers)
    const int* seq_in;           // token sequences of the beams before this step: row (i * seq_in_rows_per + parent) x max_ids
    int seq_in_rows_per;
    int* seq_out;                // ... and after it: row i * B + b
    int *ids, *row_pos;          // metadata of the NEXT forward pass (rows i * B + b)
    LlmStreamView* views;
    SampleStream* samp;
    KvCopyOp *ops1, *ops2;       // position copies of this step: arena -> buffer (hypothesis tails, parents staged), then buffer -> arena
    int* op_counts;              // [2], zero before the launch
    BeamReorder* reorder;        // [n] non-null: the step's copies as one record per stream (kv_beam_reorder_kernel) instead of the two op lists
    long stream_stride, tbuf_stride;
    int max_beams, nbuf;
    float* log_val;              // pinned host log of this step: the candidates ...
    int* log_idx;
    BeamDecision* log_dec;       // ... the choices [n * B] ...
    int* log_done;               // ... per stream: 0 / 1 = BeamSearchScorer done flag, < 0 = failure (-1: an earlier launch of the pass raised the error word,
                                 //     -2: out of hypothesis buffers, -3: fewer than B non-EOS candidates, -4: forced choice out of range)
    int* log_seq;                // ... and the launch's sequence number, stored once every stream's part is in host memory
    int seq_value;
    int* ticket;                 // device counter, zero between launches
    const int* err_word;         // pinned error word of the fused attention launch (0 = fine)
    const int *force_tok, *force_par;  // teacher forcing of stream 0 (test aid): [force_steps][B]
    int force_steps;
    const float* logits;         // the score rows of this step (forced choices read theirs here): processed log-probs, or -- view.logz != null -- rows as
    long ld_logits;              //     beam_process_kernel leaves them
    BeamScoreView view;
};
int launch_beam_select(const BeamSelArgs& a, hipStream_t s);
// 30-bit digest of a stream's hypothesis bookkeeping, the same arithmetic on the device (beam_select_kernel, in the step's status word) and on the host
// (the follower of engine_llm.hip): number of kept hypotheses, the tail buffer each one holds in insertion order, the free list in order
__host__ __device__ inline unsigned beam_book_digest(int hyp_n, const int* hyp_buf, int n_free, const int* free_bufs) {
    unsigned d = 2166136261u;
    auto mix = [&](int v) { d = (d ^ (unsigned)(v + 0x40)) * 16777619u; };
    mix(hyp_n);
    for (int q = 0; q < hyp_n; ++q) mix(hyp_buf[q]);
    mix(n_free);
    for (int q = 0; q < n_free; ++q) mix(free_bufs[q]);
    return (d ^ (d >> 15)) & 0x3FFFFFFFu;
}
int launch_topk_rows(const float* scores, long ld, int vocab, int k, float* cval, int* cidx, float* out_val, int* out_idx, int rows,
                     hipStream_t s);
// the same in ONE launch; the tokens also go to the pinned array `host_tokens`, and `*host_seq` receives the launch's sequence number (tickets[1], kept on
// the device) once every stream's token is there.  tickets: 2 + max streams ints, zero before the first launch
// adv (one stream's greedy loop): the launch also PREPARES THE NEXT PASS on the device -- the sampled id appended to the stream's id list (ids_pool, ss[0].n_ids),
// written to ids[0], and its embedding row copied to `lx` (the decoder's input row: model/llm.py:114-115 embeds the last token only) -- so that the host
// neither uploads a metadata block nor launches the embedding kernel between two passes
struct SampleAdvance {
    int enabled;
    int* ids;              // [0] <- the sampled id
    int* ids_pool;         // the stream's id list (same memory as the const view the processors read)
    SampleStream* ss;      // [0].n_ids += 1
    const bf16_t* embed;   // [vocab][D]
    bf16_t* lx;            // [D] <- embed[id]
    int D;
};
int launch_sample_fused(float* logits, long ld_logits, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress,
                        int n_suppress, float rep_penalty, int ngram, int enc_ngram, int* out_tokens, float* scratch_val, int* scratch_idx, int* tickets,
                        int* host_tokens, int* host_seq, int n_streams, hipStream_t s, const SampleAdvance* adv = nullptr);
// processors only (stage 1 of launch_sample), in place on the rows named by ss[]
// warp.hip (host): HF's logits warpers (Temperature -> TopK -> TopP -> Epsilon) on one row of processed scores, then an inverse-CDF draw at `u`
int warp_and_sample(float* scores, int n, float temperature, int top_k, float top_p, float epsilon, double u);
void warp_scores(float* scores, int n, float temperature, int top_k, float top_p, float epsilon, int min_keep);
int multinomial_without_replacement(const float* scores, long n, int k, const double* u, long* picked);
double sample_uniform(uint64_t seed, int stream, int chunk, int step);
int launch_sample_process(float* logits, long ld_logits, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress,
                          int n_suppress, float rep_penalty, int ngram, int enc_ngram, int n_rows, hipStream_t s);
