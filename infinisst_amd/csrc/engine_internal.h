// Internal declarations shared by the four translation units of the host side of libinfinisst_hip.so (round 4: engine.hip was one 2 300-line file):
//   engine_core.hip    handle, configuration, weights, streams, caches (isst_create / _load_weight / _stream_* / _kv_evict / imports / debug reads)
//   engine_encode.hip  conv extractor + streaming encoder + shrink + projector (run_encoder, isst_encode_speech)
//   engine_llm.hip     decoder stack, greedy / sample loop, beam search (llm_forward, beam_decode, isst_generate)
//   engine_ops.hip     the per-kernel entry points of the C ABI (isst_op_*), the profiling hooks
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/infinisst_hip.h"
#include "common.h"
#include "kernels.h"

#define KV_OPS_SLOTS 8  // batches of beam-search KV position copies that may be enqueued between two stream synchronisations
#define LLM_KSPLIT_MAX 8
#ifndef LLM_SPLIT_TARGET_WGS
#define LLM_SPLIT_TARGET_WGS 768
#endif
#ifndef LLM_SPLIT_MAX_ROWS
#define LLM_SPLIT_MAX_ROWS 2048  // rows up to which o_proj / down_proj run split-K into slabs (beyond, the dense kernel has the workgroups)
#endif
#define ENC_SPLIT_MAX_ROWS 8192  // ... and the encoder's out_proj / fc2 (1024 columns: 4 column blocks of the dense kernel; the slab capacity bounds the slices)
#ifndef LLM_SLAB_ROWS
#define LLM_SLAB_ROWS 4096       // slices x rows the slab buffer holds
#endif
// rows up to which the decoder fuses RMSNorm into the following projection (the kernel supports GEMM_FUSED_NORM_MAX_M): every
// workgroup re-normalises all rows while its first weight fragments are in flight.  Measured per launch (profiles/prologue_probe.py):
// free at 1-2 rows, +0.8 us (gate/up) / +1.8 us (q/k/v) at 4 rows against 4.7 us for the norm launch it replaces, +11 us at 8 rows
// (64 KB of LDS per workgroup: two workgroups per CU); the 8017-workgroup lm_head pays the prologue once per workgroup ROUND
// (+13 us at 4 rows), so it only fuses up to 2 rows.
#define LLM_FUSED_NORM_MAX_ROWS 4
#define LLM_FUSED_NORM_MAX_ROWS_LM_HEAD 2
namespace isst_impl {

struct PackedLinear {
    bf16_t* wp = nullptr;
    bf16_t* bias = nullptr;
    int N = 0;        // packed rows (multiple of 16)
    int K = 0;
    int n_valid = 0;  // real output columns
};
struct Norm {
    bf16_t* w = nullptr;
    bf16_t* b = nullptr;
};
struct ConvLayer {
    PackedLinear lin;         // layers >= 1 (implicit GEMM)
    bf16_t* w_raw = nullptr;  // layer 0: [C][k]
    Norm ln;
    int dim = 0, k = 0, stride = 0;
};
struct EncLayer {
    Norm ln1, ln2;
    PackedLinear qkv, out, fc1, fc2;
};
struct LlmLayer {
    bf16_t* in_norm = nullptr;
    bf16_t* post_norm = nullptr;
    PackedLinear qkv, o, gateup, down;
    PackedLinear gateup8;  // gate / up once more, as self-paired tiles (gemm.hip EPI_SWIGLU8): what a ONE-row pass streams -- 1792 one-tile workgroups (7 per CU) instead of 896 tile pairs (3.5 per CU); wp == null: not kept (ISST_GATEUP8=0)
};
struct StreamState {
    bool open = false;
    int chunks = 0;
    int enc_start = 0, enc_len = 0, enc_steps = 0;
    int llm_sys = 0;         // pinned boundary: logical positions < llm_sys live in the sys region
    int llm_ring_start = 0;  // physical ring slot of logical position llm_sys
    int llm_total = 0;       // cached entries (logical positions 0..llm_total-1)
    int beams = 0;           // 0: not decided yet, 1: greedy, >1: beam search (all arenas of the stream stay in sync)
};
struct Tap {
    bf16_t* dev = nullptr;
    int64_t cap = 0, elems = 0;
};

extern std::string g_create_error;

}  // namespace isst_impl
using namespace isst_impl;

struct isst_handle {
    isst_config cfg{};
    std::string err;
    std::vector<void*> allocs;
    std::set<std::string> loaded;
    std::vector<std::string> expected;
    bool finalized = false, rope_set = false;

    // one decode step (metadata upload, decoder stack, sampling, token download) of a fixed row count as a replayable hipGraph
    struct DecodeGraph {
        hipGraphExec_t exec = nullptr;
        int rows = -1, n_suppress = 0, ngram = 0, enc_ngram = 0;
        float penalty = 0.f;
    } dgraph;
    bool rot_keys = true;   // ISST_ROT_KEYS=0: rotate cached keys on every read (the reference's schedule) instead of once per chunk
    int kv_ops_used = 0;      // slots of the pinned KV-copy op list handed out since the stream was last known idle (flush_copies)
    bool beam_shared = true;  // ISST_BEAM_SHARED=0: every beam reads its whole arena (B x the attention traffic) instead of sharing the prefix pass
    // The split-KV combine of a one-stream decode step is a launch of its own (4.75 us + a 2.6 us gap per layer and pass).  Two ways to remove
    // that launch are built, tested bit-identical (tests/test_gpu_engine.py, test_gpu_kernels.py) and measured -- neither wins on MI355X, so both
    // stay opt-in (profiles/r02/combine_fusion.txt):
    bool inline_combine = false;  // ISST_INLINE_COMBINE=1: the last workgroup of a kv head to arrive combines inside the attention launch (write-through
                                  // slabs, drained, agent-scope counter, sc1 reads): 32.39-32.66 ms per chunk against 32.09-32.18 -- the hand-off's
                                  // round trips through memory cost what the kernel boundary costs
    bool fuse_combine = false;    // ISST_FUSE_COMBINE=1: the o_proj GEMV merges the partials while it stages its A row (gemm.hip AMODE 3): every one of
                                  // its 256 workgroups re-reads all 316 KB of slabs through L2 -- 33.7 ms per chunk against 32.3
    bool fuse_attn_oproj = true;  // one stream's decode step: attention + combine + o_proj + residual as ONE launch (llm_attn.hip llm_attn_oproj_kernel; needs the
                                  // device to itself: N / 16 workgroups resident at once).  ISST_FUSE_ATTN_OPROJ=0: the three launches.  Bit-identical either way.
    int fuse_ao_mode = 0;         // ISST_FUSE_ATTN_OPROJ=2 -> 1: the fused launch stops after the combine, o_proj is its own launch (bisecting aid)
    bool fuse_ao_beams = true;    // ... also for the <= 4 beams of ONE stream (a shared-prefix group of B rows: 184 attention workgroups, one merging workgroup per (row, head), a
                                  // B-row GEMV).  Bit-identical (tests/test_gpu_beam.py, test_gpu_fullsize.py).  With the counted second hand-off it bought nothing (33.7-34.0 ms per
                                  // beam-4 chunk either way); with the tagged row 33.39 against 33.60 ms (profiles/r04/fused_attn_oproj_beam4_ab_v3.txt).  ISST_FUSE_ATTN_OPROJ=1: greedy only
    bool fuse_ao_test_timeout = false;  // ISST_FUSE_AO_TEST_TIMEOUT=1 (test aid): the launch waits for an arrival count that is never reached -- exercises the bounded waits and the error path
    bool fuse_ao_used = false;    // a fused launch was enqueued since the error word (tok_host[tok_cap + 8]) was last checked
    unsigned* fuse_bar = nullptr; // its hand-off counters (40 x 128 B, only ever grow)
    unsigned* fuse_row = nullptr; // its merged attention rows as {two bf16, tag} words: [4 rows][heads x 64] x 8 B
    unsigned fuse_arrive_total = 0, fuse_merge_total = 0;  // what its counters will read once every enqueued launch has run (the next launch's targets start here)
    int fuse_ao_delay = 4;        // ISST_FUSE_AO_DELAY (swept 0..16: 0-4 equal within noise, 30.93-31.06 ms per chunk; 12: 31.5; 16: 32.0): x ~0.4 us the waves without attention work hold their weight loads back
    int n_cus = 0;                // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool fuse_reduce = true;      // 13..64 rows -- no rmsnorm_reduce launches: the last K-slice workgroup of o_proj / down_proj sums the slabs and writes x
                                  // (+ sums of squares per row and 32 columns), the next projection normalises its rows while it stages them (gemm_mid.hip).
                                  // A/B on one box, ms per step: 16 streams 52.42 -> 51.28, 32: 66.92 -> 66.04, 64: 92.67 -> 92.49, one stream (22-row
                                  // prefill) equal -- the hand-off costs nearly what the launch costs.  ISST_FUSE_REDUCE=0 restores the reduce launches
    int qkv_slices = 0;           // ISST_QKV_SLICES: K slices of the q/k/v projection at 13..64 rows (in-launch reduction); 0 = by row count
    float* lssq = nullptr;        // [64][llm_dim / 32] sums of squares (GemmArgs::ssq)
    int* ltickets = nullptr;      // [ltickets_n] arrival counters (GemmArgs::tickets), one per 32-column block of the widest ticketed launch; zero between launches
    int ltickets_n = 0;
    bool rope_fuse = true;        // ISST_ROPE_FUSE=0: the rotated-key arena of a chunk is filled by a pre-pass over every layer's keys (llm_rope_cache_kernel) instead of by
                                  // the prefill attention's loader waves (LlmStreamView::rot_keys == 2)
    bool rope_side = false;       // ISST_ROPE_SIDE=1: the rotated-key pre-pass of a chunk (pure memory traffic) runs on a low-priority side stream beside the
                                  // speech encoder (MFMA-bound at many streams) and joins before the prefill.  Measured, one box, ms per step: 64 streams
                                  // 90.96 / 90.90 without, 91.17 / 90.69 with; 16 streams 51.11 / 51.21 -- nothing, stays off
    hipStream_t side = nullptr;
    hipEvent_t side_ev = nullptr, side_ev2 = nullptr;
    bool sync_at_end = false;     // ISST_SYNC_AT_END=1: isst_generate drains the stream before it returns even when every token reached the host through a
                                  // published sequence number (fused greedy tail, device beam scorer)
    bool use_graphs = false;  // ISST_GRAPH=1 enables.  Measured on MI355X (1 stream): 35.40 ms per chunk replayed vs 35.28 launched one by
                              // one -- the loop is GPU-bound, the host is ~0.6 ms ahead per pass, and a graph does not shorten the
                              // GPU-side kernel boundaries; it only saves host time (230 launches -> 1 per step)

    // in-situ timing of the dominant kernel (isst_profile_begin / _end): HIP event pairs around every decode-pass gate/up GEMV
    bool prof_on = false;
    int prof_rows_lo = 1, prof_rows_hi = 1;  // passes whose gate/up launch is bracketed (1..1: the decode GEMV, the roofline kernel)
    std::vector<hipEvent_t> prof_ev;  // pairs (start, stop)
    size_t prof_used = 0;

    // geometry
    int hist = 0;           // receptive field - 1 samples of audio history (399)
    int samples_per_frame = 0, chunk_samples = 0, shrink_factor = 1;
    int enc_cap = 0;        // encoder ring slots
    int sys_cap = 0, ring_cap = 0;
    int vocab_pad = 0;
    int max_ids = 0;        // prompt + generated ids per stream and call
    int max_enc_ids = 256;
    int n_new_max = 0, enc_rows_max = 0, llm_rows_max = 0;
    std::vector<int> conv_T;  // scratch

    // weights
    std::vector<ConvLayer> conv;
    Norm enc_ln_in;
    PackedLinear post_proj;
    std::vector<EncLayer> enc;
    Norm enc_ln_out;
    std::vector<ConvLayer> shrink;
    PackedLinear proj;
    bf16_t* embed = nullptr;
    std::vector<LlmLayer> llm;
    bf16_t* final_norm = nullptr;
    PackedLinear lm_head;
    float *enc_cos = nullptr, *enc_sin = nullptr;
    bf16_t* enc_cs = nullptr;   // both encoder tables as ONE packed bf16 table (enc_attn.hip EncTab), valid when every value handed to isst_set_rope_tables is a bf16 number
    bool enc_cs_valid = false;
    bf16_t *llm_cos = nullptr, *llm_sin = nullptr;
    int enc_rope_rows = 0, llm_rope_rows = 0;
    bf16_t* enc_pos = nullptr;  // cfg.enc_abs_pos (--rope 0): sinusoid rows [enc_pos_rows][enc_dim], one per bf16 integer position (isst_set_enc_position_table)
    int enc_pos_rows = 0;
    bf16_t* stage = nullptr;
    size_t stage_bytes = 0;

    // state pools
    std::vector<StreamState> streams;
    bf16_t* audio_hist = nullptr;               // [max_streams][hist]
    bf16_t *enc_k = nullptr, *enc_v = nullptr;  // [max_streams][enc_layers][heads][enc_cap][64]
    long enc_stream_stride = 0, enc_layer_stride = 0;
    bf16_t *llm_k = nullptr, *llm_v = nullptr;  // [max_streams][llm_layers][kv_heads][sys_cap+ring_cap][128]
    bf16_t* llm_kr = nullptr;                   // same geometry as llm_k: the keys rotated at their logical position of the current chunk (llm_attn.hip)
    long llm_stream_stride = 0;
    LlmAttnDims adims{};

    // workspace
    float* pcm_f32 = nullptr;   // [max_streams][n_new_max] samples of the call, then [max_streams] stream ids (ints)
    float* pcm_host = nullptr;  // pinned twin: one upload per call
    bf16_t *window = nullptr, *act_a = nullptr, *act_b = nullptr;
    bf16_t *ex = nullptr, *exn = nullptr, *eqkv = nullptr, *eattn = nullptr, *effn = nullptr, *speech = nullptr;
    bf16_t *lx = nullptr, *lxn = nullptr, *lqkv = nullptr, *lqrot = nullptr, *lattn = nullptr, *lact = nullptr, *llast = nullptr;
    float *lpartial = nullptr, *logits = nullptr;
    int* attn_cnt = nullptr;
    float* lslab = nullptr;  // split-K slabs of o_proj / down_proj at 17..512 rows: [slices][rows][llm_dim] fp32
    long lslab_elems = 0;    // fp32 elements lslab holds: every K-slice choice and every EPI_PARTIAL launch is checked against it
    int* out_tok = nullptr;
    float* samp_val = nullptr;  // partial argmax scratch, 64 per stream
    int* samp_idx = nullptr;
    unsigned char* meta_dev = nullptr;
    unsigned char* meta_host = nullptr;  // pinned
    size_t meta_bytes = 0;
    int* tok_host = nullptr;  // pinned: [NB] token ids of the last sampling tail, [NB] the sequence number of the last fused tail that has published all of them
    int* samp_tickets = nullptr;  // sample_fused_kernel's counters (device): [0] streams done, [1] sequence number, [2 + stream] parts done
    int tok_cap = 0;
    int samp_seq_expected = 0;  // fused tails launched so far (the device keeps the same count in samp_tickets[1])
    bool tail_advance = true;   // ISST_TAIL_ADVANCE=0: one stream's decode step uploads its metadata block and launches the embedding kernel, as before round 5
    bool fused_sample = true;   // ISST_FUSED_SAMPLE=0: three launches + D2H copy + stream synchronisation per token instead of one launch + a wait on pinned memory
    float* samp_host = nullptr;  // pinned [rows][vocab_pad]: processed scores of a sampling step (allocated on the first do_sample call)
    size_t samp_host_rows = 0;

    // beam search (max_beams > 1): one KV arena per (stream, beam), tail buffers, scoring scratch
    int max_beams = 1, tcap = 0, nbuf = 0;
    bf16_t *tbuf_k = nullptr, *tbuf_v = nullptr, *tbuf_kr = nullptr;  // [max_streams][nbuf][layers][kv][tcap][128]
    long tbuf_stride = 0;
    float *lse_max = nullptr, *lse_sum = nullptr, *cand_val = nullptr, *top_val = nullptr;
    int *cand_idx = nullptr, *top_idx = nullptr;
    float* top_val_host = nullptr;
    int* top_idx_host = nullptr;

    // beam search with the scorer on the device (beam.hip beam_select_kernel; engine_llm.hip beam_decode_device): the state of a call's streams, the n x B-row
    // metadata of its decode passes, the beams' token sequences (double buffered), the op lists of the position copies, and the pinned per-step log the
    // host follows the search through
    bool beam_device = true;                 // ISST_BEAM_DEVICE=0: the host scorer decides every step (the stream drains twice per step)
    unsigned char* meta_dev2 = nullptr;      // second metadata block (same carving as meta_dev)
    unsigned char* meta_host2 = nullptr;     // pinned
    int* bseq[2] = {nullptr, nullptr};       // [max_streams * max_beams][max_ids]
    BeamDevStream* bst_dev = nullptr;        // [max_streams]
    BeamDevStream* bst_host = nullptr;       // pinned staging of the same
    double* bpow_dev = nullptr;              // [max_new_tokens + 2] len^length_penalty, from the host's libm
    double* bpow_host = nullptr;             // pinned
    KvCopyOp* bops_dev[2] = {nullptr, nullptr};  // [2 NB], [NB]
    int* bop_counts_dev = nullptr;           // [max_new_tokens + 1][2]
    BeamReorder* breorder_dev = nullptr;     // [max_streams]: the step's copies as one record per stream (beam_one_copy)
    bool beam_one_copy = true;               // ISST_BEAM_ONE_COPY=0: the two op-list launches per step instead (A/B aid)
    int* bticket_dev = nullptr;
    int* bforce_dev = nullptr;               // forced (token, parent) choices of stream 0 (test aid): [2][max_new_tokens * max_beams]
    int* bforce_host = nullptr;              // pinned
    unsigned char* blog = nullptr;           // pinned log, blog_steps slots of blog_slot_bytes + the sequence word
    size_t blog_slot_bytes = 0;
    int blog_steps = 0;
    int bsel_seq = 0;                        // beam_select launches enqueued so far = the sequence number the last one publishes
    bool beam_lean_tail = true;              // ISST_BEAM_LEAN_TAIL=0: log-softmax applied in place + processors + top-k (four sweeps over the step's fp32 scores instead of two)
    BeamScoreView bview{};                   // log Z per row + the penalised entries' side lists (beam.hip beam_process_kernel)

    // beam-search test aid (isst_debug_beam_trace_*): per-step candidate lists of a ONE-stream call and optional teacher forcing
    struct BeamTraceStep { int rows, n_keep; std::vector<float> val; std::vector<int> idx; std::vector<float> score; };
    bool btrace_on = false;
    int btrace_beams = 0;
    std::vector<int> bforce_tok, bforce_par;  // [steps][beams]
    std::vector<BeamTraceStep> btrace;

    std::map<std::string, Tap> taps;
    long arena_off(int sid, int beam) const { return ((long)sid * max_beams + beam) * llm_stream_stride; }

    int fail(int code, const char* fmt, ...) {
        char buf[1024];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
    template <typename T>
    T* dalloc(size_t count, bool zero = false) {
        void* p = nullptr;
        size_t bytes = count * sizeof(T);
        if (bytes == 0) bytes = 16;
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
        if (zero) (void)hipMemset(p, 0, bytes);
        allocs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
};

#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return h->fail(ISST_ERR_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define CHK(x)                                                                        \
    do {                                                                              \
        int r_ = (x);                                                                 \
        if (r_ != ISST_OK) return h->fail(r_, "%s -> %d (%s:%d)", #x, r_, __FILE__, __LINE__); \
    } while (0)
#define NEED(p)                                                                  \
    do {                                                                         \
        if (!(p)) return h->fail(ISST_ERR_NOMEM, "device allocation failed: %s", #p); \
    } while (0)

namespace isst_impl {

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline int conv_out_len(int n, int k, int s) { return n < k ? 0 : (n - k) / s + 1; }
// position held by row `row` of the --rope 0 table (rows above 255 are consecutive bf16 bit patterns from 256.0 = 0x4380)
inline float enc_pos_row_value(int row) {
    if (row < 256) return (float)row;
    const uint32_t bits = (uint32_t)(0x4380 + row - 256) << 16;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}
// debug tap: keeps a copy of an intermediate (isst_config::debug_taps)
int tap(isst_handle* h, const std::string& name, const bf16_t* src, int64_t elems, hipStream_t st);
int check_ready(isst_handle* h);
// engine_llm.hip: one projection through the packed-weight GEMM dispatcher (gemm.hip launch_gemm), K-slice choice, the split-K forms
int gemm(isst_handle* h, const bf16_t* A, long lda, const PackedLinear& L, int epi, const bf16_t* res, long ldres, void* out, long ldo,
         int M, hipStream_t st, int batch = 1, long a_batch = 0, long out_batch = 0, long res_batch = 0,
         const bf16_t* norm_w = nullptr, float norm_eps = 0.f, float* ssq = nullptr);
int pick_ksplit(int K, int N, int rows, long slab_cap);
int gemm_partial(isst_handle* h, const bf16_t* A, long lda, const PackedLinear& L, float* slabs, int M, int ksplit, hipStream_t st,
                 bf16_t* x = nullptr, long ldx = 0, float* ssq = nullptr, bf16_t* plain_out = nullptr, long ld_plain = 0,
                 const bf16_t* norm_w = nullptr, float norm_eps = 0.f, float* ssq_in = nullptr);
// engine_encode.hip: conv extractor + encoder + shrink + projector for n streams; result in h->speech [n*S][llm_dim]
int run_encoder(isst_handle* h, int n, const int* sids, const float* const* pcm, bool pcm_on_device, int n_samples, int multiplier, hipStream_t st,
                int* out_S);
// engine_llm.hip: the speech splice as a row map (model/llm.py:86-113)
int splice_rows(const int* ids, int len, int user_id, int assistant_id, int start_header_id, int S, std::vector<int>& desc);
}  // namespace isst_impl

