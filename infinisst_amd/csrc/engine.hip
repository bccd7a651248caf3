// Host side of libinfinisst_hip.so: owns device weights (re-laid out for the MFMA GEMM), per-stream state
// (audio history, encoder KV rings, LLM KV arenas) and runs one chunk of the InfiniSST hot path for n streams:
//   conv feature extractor -> streaming encoder -> shrink + projector -> splice -> Llama prefill -> greedy decode.
// It is the MI355X replacement of `self.model.generate(...)` at reference agents/infinisst.py:307-332 (which reaches
// model/llm.py:51-126,192-270, model/speech_encoder.py:219-236, model/patches/patch_speech_encoder.py:228-933,
// model/patches/patch_llm.py:231-336 and model/patches/patch_hf.py:586-624).  No CPU fallback exists: every compute
// step is a HIP kernel launch on the caller's stream.
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
namespace {

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

static std::string g_create_error;

}  // namespace

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
    bool fuse_reduce = true;      // 13..64 rows -- no rmsnorm_reduce launches: the last K-slice workgroup of o_proj / down_proj sums the slabs and writes x
                                  // (+ sums of squares per row and 32 columns), the next projection normalises its rows while it stages them (gemm_mid.hip).
                                  // A/B on one box, ms per step: 16 streams 52.42 -> 51.28, 32: 66.92 -> 66.04, 64: 92.67 -> 92.49, one stream (22-row
                                  // prefill) equal -- the hand-off costs nearly what the launch costs.  ISST_FUSE_REDUCE=0 restores the reduce launches
    int qkv_slices = 0;           // ISST_QKV_SLICES: K slices of the q/k/v projection at 13..64 rows (in-launch reduction); 0 = by row count
    float* lssq = nullptr;        // [64][llm_dim / 32] sums of squares (GemmArgs::ssq)
    int* ltickets = nullptr;      // [ltickets_n] arrival counters (GemmArgs::tickets), one per 32-column block of the widest ticketed launch; zero between launches
    int ltickets_n = 0;
    bool rope_side = false;       // ISST_ROPE_SIDE=1: the rotated-key pre-pass of a chunk (pure memory traffic) runs on a low-priority side stream beside the
                                  // speech encoder (MFMA-bound at many streams) and joins before the prefill.  Measured, one box, ms per step: 64 streams
                                  // 90.96 / 90.90 without, 91.17 / 90.69 with; 16 streams 51.11 / 51.21 -- nothing, stays off
    hipStream_t side = nullptr;
    hipEvent_t side_ev = nullptr;
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

namespace {

const char* ENC = "model.speech_encoder.speech_encoder.";
const char* SHR = "model.speech_encoder.length_shrink.";
const char* PRJ = "model.speech_encoder.proj.";

int round_up(int x, int m) { return (x + m - 1) / m * m; }

bool alloc_linear(isst_handle* h, PackedLinear& L, int n_rows, int K, bool bias) {
    L.N = round_up(n_rows, 16);
    L.K = K;
    L.n_valid = n_rows;
    L.wp = h->dalloc<bf16_t>((size_t)L.N * K, true);
    if (bias) L.bias = h->dalloc<bf16_t>(round_up(n_rows, 8), true);
    return L.wp && (!bias || L.bias);
}
bool alloc_norm(isst_handle* h, Norm& n, int dim, bool bias = true) {
    n.w = h->dalloc<bf16_t>(round_up(dim, 8), true);
    if (bias) n.b = h->dalloc<bf16_t>(round_up(dim, 8), true);
    return n.w && (!bias || n.b);
}

void expected_names(isst_handle* h) {
    const isst_config& c = h->cfg;
    auto& e = h->expected;
    char b[256];
    for (int i = 0; i < c.n_conv; ++i) {
        snprintf(b, sizeof b, "%sfeature_extractor.conv_layers.%d.0.weight", ENC, i); e.push_back(b);
        if (c.conv_bias) { snprintf(b, sizeof b, "%sfeature_extractor.conv_layers.%d.0.bias", ENC, i); e.push_back(b); }
        snprintf(b, sizeof b, "%sfeature_extractor.conv_layers.%d.2.1.weight", ENC, i); e.push_back(b);
        snprintf(b, sizeof b, "%sfeature_extractor.conv_layers.%d.2.1.bias", ENC, i); e.push_back(b);
    }
    for (const char* s : {"layer_norm.weight", "layer_norm.bias", "post_extract_proj.weight", "post_extract_proj.bias",
                          "encoder.layer_norm.weight", "encoder.layer_norm.bias"}) {
        snprintf(b, sizeof b, "%s%s", ENC, s); e.push_back(b);
    }
    for (int i = 0; i < c.enc_layers; ++i)
        for (const char* s : {"self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight", "self_attn.k_proj.bias",
                              "self_attn.v_proj.weight", "self_attn.v_proj.bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                              "self_attn_layer_norm.weight", "self_attn_layer_norm.bias", "fc1.weight", "fc1.bias", "fc2.weight",
                              "fc2.bias", "final_layer_norm.weight", "final_layer_norm.bias"}) {
            snprintf(b, sizeof b, "%sencoder.layers.%d.%s", ENC, i, s); e.push_back(b);
        }
    for (int i = 0; i < c.n_shrink; ++i)
        for (const char* s : {"0.weight", "2.1.weight", "2.1.bias"}) {
            snprintf(b, sizeof b, "%sconv_layers.%d.%s", SHR, i, s); e.push_back(b);
        }
    snprintf(b, sizeof b, "%sweight", PRJ); e.push_back(b);
    snprintf(b, sizeof b, "%sbias", PRJ); e.push_back(b);
    e.push_back("model.embed_tokens.weight");
    for (int i = 0; i < c.llm_layers; ++i)
        for (const char* s : {"input_layernorm.weight", "self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
                              "self_attn.o_proj.weight", "post_attention_layernorm.weight", "mlp.gate_proj.weight",
                              "mlp.up_proj.weight", "mlp.down_proj.weight"}) {
            snprintf(b, sizeof b, "model.layers.%d.%s", i, s); e.push_back(b);
        }
    e.push_back("model.norm.weight");
    e.push_back("lm_head.weight");
}

int conv_out_len(int n, int k, int s) { return n < k ? 0 : (n - k) / s + 1; }

int validate_config(const isst_config& c, std::string& why) {
    auto bad = [&](const char* m) { why = m; return ISST_ERR_ARG; };
    if (c.n_conv < 1 || c.n_conv > ISST_MAX_CONV || c.n_shrink < 0 || c.n_shrink > ISST_MAX_SHRINK) return bad("n_conv / n_shrink out of range");
    if (c.enc_heads <= 0 || c.enc_dim != c.enc_heads * 64) return bad("encoder head_dim must be 64");
    if (c.llm_heads <= 0 || c.llm_kv_heads <= 0 || c.llm_heads % c.llm_kv_heads) return bad("llm heads / kv heads");
    { const int g = c.llm_heads / c.llm_kv_heads; if (g != 1 && g != 2 && g != 4) return bad("llm heads per kv head must be 1, 2 or 4"); }
    if (c.llm_dim % 32 || c.llm_ffn % 32 || c.enc_dim % 32 || c.enc_ffn % 32) return bad("hidden sizes must be multiples of 32");
    for (int i = 0; i < c.n_conv; ++i) {
        if (c.conv_dim[i] % 32 || c.conv_dim[i] > 512 || c.conv_k[i] < 1 || c.conv_k[i] > 16 || c.conv_stride[i] < 1) return bad("conv layer geometry");
    }
    for (int i = 0; i < c.n_shrink; ++i)
        if (c.shrink_dim[i] != c.enc_dim || c.shrink_k[i] < 1 || c.shrink_stride[i] != c.shrink_k[i]) return bad("shrink layers must keep enc_dim and have k == stride");
    if (c.max_streams < 1 || c.max_multiplier < 1 || c.max_prompt_len < 8 || c.max_new_tokens < 1 || c.max_llm_cache_size < 1) return bad("capacity fields");
    if (c.block_size % 4 || c.block_size < 4) return bad("block_size must be a multiple of 4");
    if (c.n_eos < 0 || c.n_eos > ISST_MAX_EOS) return bad("n_eos");
    if (c.vocab < 16) return bad("vocab");
    if (c.max_beams < 0 || c.max_beams > 8) return bad("max_beams must be 0..8");
    return ISST_OK;
}

}  // namespace

extern "C" const char* isst_last_error(isst_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" void isst_destroy(isst_handle* h) {
    if (!h) return;
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& kv : h->taps) if (kv.second.dev) (void)hipFree(kv.second.dev);
    if (h->stage) (void)hipFree(h->stage);
    if (h->meta_host) (void)hipHostFree(h->meta_host);
    if (h->tok_host) (void)hipHostFree(h->tok_host);
    if (h->samp_host) (void)hipHostFree(h->samp_host);
    if (h->pcm_host) (void)hipHostFree(h->pcm_host);
    if (h->top_val_host) (void)hipHostFree(h->top_val_host);
    if (h->top_idx_host) (void)hipHostFree(h->top_idx_host);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->side_ev) (void)hipEventDestroy(h->side_ev);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->dgraph.exec) (void)hipGraphExecDestroy(h->dgraph.exec);
    delete h;
}

extern "C" int isst_create(const isst_config* cfg, isst_handle** out) {
    if (!cfg || !out) { g_create_error = "null argument"; return ISST_ERR_ARG; }
    std::string why;
    if (int r = validate_config(*cfg, why)) { g_create_error = why; return r; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device visible (this library has no CPU path)"; return ISST_ERR_HIP; }
    isst_handle* h = new isst_handle();
    h->cfg = *cfg;
    if (const char* e = getenv("ISST_GRAPH")) h->use_graphs = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_ROT_KEYS")) h->rot_keys = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSE_COMBINE")) h->fuse_combine = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSE_REDUCE")) h->fuse_reduce = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_ROPE_SIDE")) h->rope_side = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSED_SAMPLE")) h->fused_sample = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_QKV_SLICES")) h->qkv_slices = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : 1;
    if (const char* e = getenv("ISST_INLINE_COMBINE")) h->inline_combine = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_BEAM_SHARED")) h->beam_shared = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_WIDE")) gemm_wide_set(atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : 1, 0);  // A/B runs: 0 = the 65..256-row passes on gemm_mid / gemm_tiled as before round 4 (process-wide)
    const isst_config& c = h->cfg;
    auto die = [&](int code) { g_create_error = h->err; isst_destroy(h); return code; };

    // ---- geometry ----
    int rf = 1, spf = 1;
    for (int i = c.n_conv - 1; i >= 0; --i) rf = (rf - 1) * c.conv_stride[i] + c.conv_k[i];
    for (int i = 0; i < c.n_conv; ++i) spf *= c.conv_stride[i];
    h->hist = rf - 1;
    h->samples_per_frame = spf;
    h->chunk_samples = c.block_size / 4 * 1280;  // int(block_size // 4 * 0.08 * 16000), agents/infinisst.py:201
    for (int i = 0; i < c.n_shrink; ++i) h->shrink_factor *= c.shrink_stride[i];
    if (h->chunk_samples % spf || (h->chunk_samples / spf) != c.block_size) { h->fail(ISST_ERR_ARG, "block_size %d does not match %d samples per chunk / %d samples per frame", c.block_size, h->chunk_samples, spf); return die(ISST_ERR_ARG); }
    if (c.block_size % h->shrink_factor) { h->fail(ISST_ERR_ARG, "block_size not divisible by the shrink factor"); return die(ISST_ERR_ARG); }
    h->n_new_max = h->chunk_samples * c.max_multiplier;
    h->enc_rows_max = c.max_streams * c.block_size * c.max_multiplier;
    h->llm_rows_max = c.max_streams * (c.max_prompt_len > 8 ? c.max_prompt_len : 8);
    h->enc_cap = round_up(c.max_cache_size + c.block_size * c.max_multiplier, 64);
    if (h->enc_cap > 1024) { h->fail(ISST_ERR_ARG, "encoder window %d > 1024 keys unsupported", h->enc_cap); return die(ISST_ERR_ARG); }
    h->sys_cap = round_up(c.max_system_prompt, 64);  // the attention kernel walks 64-slot splits of [sys region | ring]
    h->ring_cap = round_up(c.max_llm_cache_size + c.max_prompt_len + c.max_new_tokens + 8, 64);
    h->vocab_pad = round_up(c.vocab, 16);
    h->max_ids = c.max_prompt_len + c.max_new_tokens + 1;
    h->max_beams = c.max_beams < 1 ? 1 : c.max_beams;
    h->cfg.max_beams = h->max_beams;

    // ---- weights ----
    h->conv.resize(c.n_conv);
    int cin = 1;
    bool ok = true;
    for (int i = 0; i < c.n_conv; ++i) {
        ConvLayer& L = h->conv[i];
        L.dim = c.conv_dim[i]; L.k = c.conv_k[i]; L.stride = c.conv_stride[i];
        if (i == 0) {
            L.w_raw = h->dalloc<bf16_t>((size_t)L.dim * L.k, true);
            L.lin.bias = c.conv_bias ? h->dalloc<bf16_t>(L.dim, true) : nullptr;
            ok = ok && L.w_raw && (!c.conv_bias || L.lin.bias);
        } else {
            ok = ok && alloc_linear(h, L.lin, L.dim, cin * L.k, c.conv_bias != 0);
        }
        ok = ok && alloc_norm(h, L.ln, L.dim);
        cin = L.dim;
    }
    const int cdim = cin, D = c.enc_dim;
    ok = ok && alloc_norm(h, h->enc_ln_in, cdim) && alloc_linear(h, h->post_proj, D, cdim, true);
    h->enc.resize(c.enc_layers);
    for (auto& L : h->enc)
        ok = ok && alloc_norm(h, L.ln1, D) && alloc_norm(h, L.ln2, D) && alloc_linear(h, L.qkv, 3 * D, D, true) &&
             alloc_linear(h, L.out, D, D, true) && alloc_linear(h, L.fc1, c.enc_ffn, D, true) && alloc_linear(h, L.fc2, D, c.enc_ffn, true);
    ok = ok && alloc_norm(h, h->enc_ln_out, D);
    h->shrink.resize(c.n_shrink);
    for (int i = 0; i < c.n_shrink; ++i) {
        ConvLayer& L = h->shrink[i];
        L.dim = c.shrink_dim[i]; L.k = c.shrink_k[i]; L.stride = c.shrink_stride[i];
        ok = ok && alloc_linear(h, L.lin, L.dim, D * L.k, false) && alloc_norm(h, L.ln, L.dim);
    }
    const int DL = c.llm_dim, H = c.llm_heads, KV = c.llm_kv_heads;
    ok = ok && alloc_linear(h, h->proj, DL, D, true);
    h->embed = h->dalloc<bf16_t>((size_t)c.vocab * DL, true);
    ok = ok && h->embed;
    h->llm.resize(c.llm_layers);
    for (auto& L : h->llm) {
        L.in_norm = h->dalloc<bf16_t>(DL, true);
        L.post_norm = h->dalloc<bf16_t>(DL, true);
        ok = ok && L.in_norm && L.post_norm && alloc_linear(h, L.qkv, (H + 2 * KV) * 128, DL, false) && alloc_linear(h, L.o, DL, H * 128, false) &&
             alloc_linear(h, L.gateup, 2 * c.llm_ffn, DL, false) && alloc_linear(h, L.down, DL, c.llm_ffn, false);
        L.gateup.n_valid = c.llm_ffn;
    }
    h->final_norm = h->dalloc<bf16_t>(DL, true);
    ok = ok && h->final_norm && alloc_linear(h, h->lm_head, c.vocab, DL, false);
    if (!ok) { h->fail(ISST_ERR_NOMEM, "weight allocation failed"); return die(ISST_ERR_NOMEM); }
    expected_names(h);

    // ---- state pools ----
    h->streams.resize(c.max_streams);
    h->audio_hist = h->dalloc<bf16_t>((size_t)c.max_streams * round_up(h->hist, 8) + 8, true);
    h->enc_layer_stride = (long)c.enc_heads * h->enc_cap * 64;
    h->enc_stream_stride = h->enc_layer_stride * c.enc_layers;
    h->enc_k = h->dalloc<bf16_t>((size_t)h->enc_stream_stride * c.max_streams, true);
    h->enc_v = h->dalloc<bf16_t>((size_t)h->enc_stream_stride * c.max_streams, true);
    h->adims.heads = H; h->adims.kv_heads = KV; h->adims.sys_cap = h->sys_cap; h->adims.ring_cap = h->ring_cap;
    h->adims.layer_stride = (long)KV * (h->sys_cap + h->ring_cap) * 128;
    h->llm_stream_stride = h->adims.layer_stride * c.llm_layers;
    h->llm_k = h->dalloc<bf16_t>((size_t)h->llm_stream_stride * c.max_streams * h->max_beams, true);
    h->llm_v = h->dalloc<bf16_t>((size_t)h->llm_stream_stride * c.max_streams * h->max_beams, true);
    h->llm_kr = h->dalloc<bf16_t>((size_t)h->llm_stream_stride * c.max_streams * h->max_beams, true);
    h->enc_rope_rows = h->enc_cap;
    h->llm_rope_rows = h->sys_cap + h->ring_cap;
    h->enc_cos = h->dalloc<float>((size_t)h->enc_rope_rows * 32, true);
    h->enc_sin = h->dalloc<float>((size_t)h->enc_rope_rows * 32, true);
    h->llm_cos = h->dalloc<bf16_t>((size_t)h->llm_rope_rows * 64, true);
    h->llm_sin = h->dalloc<bf16_t>((size_t)h->llm_rope_rows * 64, true);

    // ---- workspace ----
    const int ns = c.max_streams;
    const int win = h->hist + h->n_new_max;
    const int T0 = conv_out_len(win, c.conv_k[0], c.conv_stride[0]);
    int cmax = 0;
    for (int i = 0; i < c.n_conv; ++i) cmax = c.conv_dim[i] > cmax ? c.conv_dim[i] : cmax;
    const size_t ER = h->enc_rows_max, LR = h->llm_rows_max;
    h->pcm_f32 = h->dalloc<float>((size_t)ns * h->n_new_max + ns);
    h->window = h->dalloc<bf16_t>((size_t)ns * round_up(win, 8));
    h->act_a = h->dalloc<bf16_t>((size_t)ns * T0 * cmax);
    h->act_b = h->dalloc<bf16_t>((size_t)ns * T0 * cmax);
    h->ex = h->dalloc<bf16_t>(ER * D); h->exn = h->dalloc<bf16_t>(ER * D); h->eqkv = h->dalloc<bf16_t>(ER * 3 * D);
    h->eattn = h->dalloc<bf16_t>(ER * D); h->effn = h->dalloc<bf16_t>(ER * c.enc_ffn);
    h->speech = h->dalloc<bf16_t>(ER * DL);
    h->lx = h->dalloc<bf16_t>(LR * DL); h->lxn = h->dalloc<bf16_t>(LR * DL); h->lqkv = h->dalloc<bf16_t>(LR * (H + 2 * KV) * 128);
    h->lqrot = h->dalloc<bf16_t>(LR * H * 128); h->lattn = h->dalloc<bf16_t>(LR * H * 128); h->lact = h->dalloc<bf16_t>(LR * c.llm_ffn);
    h->llast = h->dalloc<bf16_t>((size_t)ns * h->max_beams * DL);
    h->lpartial = h->dalloc<float>(LR * H * ((h->sys_cap + h->ring_cap) / 64) * ATTN_SLAB);
    h->lssq = h->dalloc<float>((size_t)64 * (DL / 32), true);
    h->ltickets_n = std::max(DL, (H + 2 * KV) * 128) / 32 + 16;  // a ticketed launch indexes tickets[blockIdx.x]; its narrowest workgroup spans 32 columns (gemm_mid NP = 1)
    h->ltickets = h->dalloc<int>((size_t)h->ltickets_n, true);
    h->attn_cnt = h->dalloc<int>(64, true);  // arrival counters of the in-kernel split-KV combine (llm_attn.hip), one per kv head; zero between launches
    h->lslab_elems = (long)LLM_SLAB_ROWS * std::max(DL, (H + 2 * KV) * 128);
    h->lslab = h->dalloc<float>((size_t)h->lslab_elems);
    const size_t NB = (size_t)ns * h->max_beams;  // decode rows of a beam step
    h->logits = h->dalloc<float>(NB * h->vocab_pad);
    h->out_tok = h->dalloc<int>(NB);
    h->samp_val = h->dalloc<float>(NB * 64);
    h->samp_idx = h->dalloc<int>(NB * 64);
    h->samp_tickets = h->dalloc<int>(NB + 2, true);
    h->tok_cap = (int)NB;
    if (h->max_beams > 1) {
        h->tcap = c.max_prompt_len > c.max_new_tokens ? c.max_prompt_len : c.max_new_tokens;
        h->nbuf = 2 * h->max_beams + 1;
        h->tbuf_stride = (long)c.llm_layers * KV * h->tcap * 128;
        h->tbuf_k = h->dalloc<bf16_t>((size_t)h->tbuf_stride * h->nbuf * ns);
        h->tbuf_v = h->dalloc<bf16_t>((size_t)h->tbuf_stride * h->nbuf * ns);
        h->tbuf_kr = h->dalloc<bf16_t>((size_t)h->tbuf_stride * h->nbuf * ns);
        h->lse_max = h->dalloc<float>(NB * 64);
        h->lse_sum = h->dalloc<float>(NB * 64);
        h->cand_val = h->dalloc<float>(NB * 64 * BEAM_TOPK);
        h->cand_idx = h->dalloc<int>(NB * 64 * BEAM_TOPK);
        h->top_val = h->dalloc<float>(NB * BEAM_TOPK);
        h->top_idx = h->dalloc<int>(NB * BEAM_TOPK);
        if (!h->tbuf_k || !h->tbuf_v || !h->tbuf_kr || !h->lse_max || !h->lse_sum || !h->cand_val || !h->cand_idx || !h->top_val || !h->top_idx) {
            h->fail(ISST_ERR_NOMEM, "beam search allocation failed"); return die(ISST_ERR_NOMEM);
        }
    }
    h->meta_bytes = (size_t)LR * 8 * sizeof(int) + NB * (sizeof(int) + sizeof(LlmStreamView) + sizeof(SampleStream) + sizeof(EncStreamView)) +
                    NB * (h->max_ids + h->max_enc_ids) * sizeof(int) + NB * 4 * sizeof(KvCopyOp) * KV_OPS_SLOTS + 65536 * sizeof(int) + 8192;
    h->meta_dev = h->dalloc<unsigned char>(h->meta_bytes);
    const void* must[] = {h->audio_hist, h->enc_k, h->enc_v, h->llm_k, h->llm_v, h->llm_kr, h->enc_cos, h->enc_sin, h->llm_cos, h->llm_sin, h->pcm_f32,
                          h->window, h->act_a, h->act_b, h->ex, h->exn, h->eqkv, h->eattn, h->effn, h->speech, h->lx, h->lxn, h->lqkv, h->lqrot,
                          h->lattn, h->lact, h->llast, h->lpartial, h->lslab, h->logits, h->out_tok, h->samp_val, h->samp_idx, h->meta_dev};
    for (const void* p : must)
        if (!p) { h->fail(ISST_ERR_NOMEM, "state/workspace allocation failed"); return die(ISST_ERR_NOMEM); }
    if (hipHostMalloc(reinterpret_cast<void**>(&h->meta_host), h->meta_bytes) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->tok_host), sizeof(int) * (NB + 16), hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||  // (kernels store tokens + a sequence number here that the host polls: fine-grained whatever HIP_HOST_COHERENT says)
        hipHostMalloc(reinterpret_cast<void**>(&h->pcm_host), sizeof(float) * ((size_t)c.max_streams * h->n_new_max + c.max_streams)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->top_val_host), sizeof(float) * NB * BEAM_TOPK, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&h->top_idx_host), sizeof(int) * NB * BEAM_TOPK, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
        h->fail(ISST_ERR_NOMEM, "pinned host allocation failed"); return die(ISST_ERR_NOMEM);
    }
    std::memset(h->tok_host, 0, sizeof(int) * (NB + 16));  // the published sequence number starts at 0 = "no fused tail yet" (samp_seq_expected counts from 1)
    if (hipDeviceSynchronize() != hipSuccess) { h->fail(ISST_ERR_HIP, "device sync after allocation failed"); return die(ISST_ERR_HIP); }
    *out = h;
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// weights
// --------------------------------------------------------------------------------------------
namespace {

bool shape_is(int ndim, const int64_t* s, std::initializer_list<int64_t> want) {
    if (ndim != (int)want.size()) return false;
    int i = 0;
    for (int64_t w : want) if (s[i++] != w) return false;
    return true;
}

int copy_vec(isst_handle* h, bf16_t* dst, const bf16_t* src, size_t n) {
    HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(bf16_t), hipMemcpyDeviceToDevice, 0));
    return ISST_OK;
}
int pack_into(isst_handle* h, PackedLinear& L, const bf16_t* src, int n_rows, int row_offset_tiles, int tile_stride, int tile_phase, int conv_k) {
    CHK(launch_pack_weight(src, L.wp, n_rows, L.K, row_offset_tiles, tile_stride, tile_phase, conv_k, 0));
    return ISST_OK;
}

}  // namespace

extern "C" int isst_load_weight(isst_handle* h, const char* name, const void* data, int ndim, const int64_t* shape, int on_device) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 3) return h ? h->fail(ISST_ERR_ARG, "isst_load_weight: bad argument") : ISST_ERR_ARG;
    const isst_config& c = h->cfg;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    const bf16_t* src = reinterpret_cast<const bf16_t*>(data);
    if (!on_device) {
        if (n * 2 > h->stage_bytes) {
            HIPCHK(hipDeviceSynchronize());
            if (h->stage) (void)hipFree(h->stage);
            h->stage = nullptr; h->stage_bytes = 0;
            HIPCHK(hipMalloc(reinterpret_cast<void**>(&h->stage), n * 2));
            h->stage_bytes = n * 2;
        }
        HIPCHK(hipDeviceSynchronize());  // previous pack kernel may still read the staging buffer
        HIPCHK(hipMemcpy(h->stage, data, n * 2, hipMemcpyHostToDevice));
        src = h->stage;
    }
    const std::string nm(name);
    auto bad_shape = [&]() { return h->fail(ISST_ERR_ARG, "unexpected shape for %s", name); };
    int li = -1;
    char suf[128] = {0};
    const int D = c.enc_dim, DL = c.llm_dim, H = c.llm_heads, KV = c.llm_kv_heads;
    const std::string enc(ENC), shr(SHR), prj(PRJ);
    int rc = ISST_ERR_NOTFOUND;
    if (nm.compare(0, enc.size(), enc) == 0) {
        const char* rest = name + enc.size();
        if (sscanf(rest, "feature_extractor.conv_layers.%d.%127s", &li, suf) == 2 && li >= 0 && li < c.n_conv) {
            ConvLayer& L = h->conv[li];
            const int cin = li == 0 ? 1 : h->conv[li - 1].dim;
            if (!strcmp(suf, "0.weight")) {
                if (!shape_is(ndim, shape, {L.dim, cin, L.k})) return bad_shape();
                rc = li == 0 ? copy_vec(h, L.w_raw, src, n) : pack_into(h, L.lin, src, L.dim, 0, 1, 0, L.k);
            } else if (!strcmp(suf, "0.bias")) {
                if (!c.conv_bias || !shape_is(ndim, shape, {L.dim})) return bad_shape();
                rc = copy_vec(h, L.lin.bias, src, n);
            } else if (!strcmp(suf, "2.1.weight") || !strcmp(suf, "2.1.bias")) {
                if (!shape_is(ndim, shape, {L.dim})) return bad_shape();
                rc = copy_vec(h, suf[4] == 'w' ? L.ln.w : L.ln.b, src, n);
            }
        } else if (sscanf(rest, "encoder.layers.%d.%127s", &li, suf) == 2 && li >= 0 && li < c.enc_layers) {
            EncLayer& L = h->enc[li];
            struct { const char* n; int part; } qkv[] = {{"self_attn.q_proj", 0}, {"self_attn.k_proj", 1}, {"self_attn.v_proj", 2}};
            for (auto& q : qkv) {
                const std::string wn = std::string(q.n) + ".weight", bn = std::string(q.n) + ".bias";
                if (wn == suf) { if (!shape_is(ndim, shape, {D, D})) return bad_shape(); rc = pack_into(h, L.qkv, src, D, q.part * D / 16, 1, 0, 0); }
                if (bn == suf) { if (!shape_is(ndim, shape, {D})) return bad_shape(); rc = copy_vec(h, L.qkv.bias + (size_t)q.part * D, src, n); }
            }
            if (!strcmp(suf, "self_attn.out_proj.weight")) { if (!shape_is(ndim, shape, {D, D})) return bad_shape(); rc = pack_into(h, L.out, src, D, 0, 1, 0, 0); }
            if (!strcmp(suf, "self_attn.out_proj.bias")) { if (!shape_is(ndim, shape, {D})) return bad_shape(); rc = copy_vec(h, L.out.bias, src, n); }
            if (!strcmp(suf, "fc1.weight")) { if (!shape_is(ndim, shape, {c.enc_ffn, D})) return bad_shape(); rc = pack_into(h, L.fc1, src, c.enc_ffn, 0, 1, 0, 0); }
            if (!strcmp(suf, "fc1.bias")) { if (!shape_is(ndim, shape, {c.enc_ffn})) return bad_shape(); rc = copy_vec(h, L.fc1.bias, src, n); }
            if (!strcmp(suf, "fc2.weight")) { if (!shape_is(ndim, shape, {D, c.enc_ffn})) return bad_shape(); rc = pack_into(h, L.fc2, src, D, 0, 1, 0, 0); }
            if (!strcmp(suf, "fc2.bias")) { if (!shape_is(ndim, shape, {D})) return bad_shape(); rc = copy_vec(h, L.fc2.bias, src, n); }
            struct { const char* n; bf16_t* p; } norms[] = {{"self_attn_layer_norm.weight", L.ln1.w}, {"self_attn_layer_norm.bias", L.ln1.b},
                                                            {"final_layer_norm.weight", L.ln2.w}, {"final_layer_norm.bias", L.ln2.b}};
            for (auto& q : norms)
                if (!strcmp(suf, q.n)) { if (!shape_is(ndim, shape, {D})) return bad_shape(); rc = copy_vec(h, q.p, src, n); }
        } else {
            const int cdim = h->conv.back().dim;
            if (!strcmp(rest, "layer_norm.weight") || !strcmp(rest, "layer_norm.bias")) {
                if (!shape_is(ndim, shape, {cdim})) return bad_shape();
                rc = copy_vec(h, rest[11] == 'w' ? h->enc_ln_in.w : h->enc_ln_in.b, src, n);
            } else if (!strcmp(rest, "post_extract_proj.weight")) {
                if (!shape_is(ndim, shape, {D, cdim})) return bad_shape();
                rc = pack_into(h, h->post_proj, src, D, 0, 1, 0, 0);
            } else if (!strcmp(rest, "post_extract_proj.bias")) {
                if (!shape_is(ndim, shape, {D})) return bad_shape();
                rc = copy_vec(h, h->post_proj.bias, src, n);
            } else if (!strcmp(rest, "encoder.layer_norm.weight") || !strcmp(rest, "encoder.layer_norm.bias")) {
                if (!shape_is(ndim, shape, {D})) return bad_shape();
                rc = copy_vec(h, rest[19] == 'w' ? h->enc_ln_out.w : h->enc_ln_out.b, src, n);
            }
        }
    } else if (nm.compare(0, shr.size(), shr) == 0) {
        if (sscanf(name + shr.size(), "conv_layers.%d.%127s", &li, suf) == 2 && li >= 0 && li < c.n_shrink) {
            ConvLayer& L = h->shrink[li];
            if (!strcmp(suf, "0.weight")) { if (!shape_is(ndim, shape, {L.dim, D, L.k})) return bad_shape(); rc = pack_into(h, L.lin, src, L.dim, 0, 1, 0, L.k); }
            if (!strcmp(suf, "2.1.weight")) { if (!shape_is(ndim, shape, {L.dim})) return bad_shape(); rc = copy_vec(h, L.ln.w, src, n); }
            if (!strcmp(suf, "2.1.bias")) { if (!shape_is(ndim, shape, {L.dim})) return bad_shape(); rc = copy_vec(h, L.ln.b, src, n); }
        }
    } else if (nm.compare(0, prj.size(), prj) == 0) {
        if (nm == prj + "weight") { if (!shape_is(ndim, shape, {DL, D})) return bad_shape(); rc = pack_into(h, h->proj, src, DL, 0, 1, 0, 0); }
        if (nm == prj + "bias") { if (!shape_is(ndim, shape, {DL})) return bad_shape(); rc = copy_vec(h, h->proj.bias, src, n); }
    } else if (nm == "model.embed_tokens.weight") {
        if (!shape_is(ndim, shape, {c.vocab, DL})) return bad_shape();
        rc = copy_vec(h, h->embed, src, n);
    } else if (nm == "model.norm.weight") {
        if (!shape_is(ndim, shape, {DL})) return bad_shape();
        rc = copy_vec(h, h->final_norm, src, n);
    } else if (nm == "lm_head.weight") {
        if (!shape_is(ndim, shape, {c.vocab, DL})) return bad_shape();
        rc = pack_into(h, h->lm_head, src, c.vocab, 0, 1, 0, 0);
    } else if (sscanf(name, "model.layers.%d.%127s", &li, suf) == 2 && li >= 0 && li < c.llm_layers) {
        LlmLayer& L = h->llm[li];
        if (!strcmp(suf, "input_layernorm.weight")) { if (!shape_is(ndim, shape, {DL})) return bad_shape(); rc = copy_vec(h, L.in_norm, src, n); }
        if (!strcmp(suf, "post_attention_layernorm.weight")) { if (!shape_is(ndim, shape, {DL})) return bad_shape(); rc = copy_vec(h, L.post_norm, src, n); }
        if (!strcmp(suf, "self_attn.q_proj.weight")) { if (!shape_is(ndim, shape, {H * 128, DL})) return bad_shape(); rc = pack_into(h, L.qkv, src, H * 128, 0, 1, 0, 0); }
        if (!strcmp(suf, "self_attn.k_proj.weight")) { if (!shape_is(ndim, shape, {KV * 128, DL})) return bad_shape(); rc = pack_into(h, L.qkv, src, KV * 128, H * 8, 1, 0, 0); }
        if (!strcmp(suf, "self_attn.v_proj.weight")) { if (!shape_is(ndim, shape, {KV * 128, DL})) return bad_shape(); rc = pack_into(h, L.qkv, src, KV * 128, (H + KV) * 8, 1, 0, 0); }
        if (!strcmp(suf, "self_attn.o_proj.weight")) { if (!shape_is(ndim, shape, {DL, H * 128})) return bad_shape(); rc = pack_into(h, L.o, src, DL, 0, 1, 0, 0); }
        if (!strcmp(suf, "mlp.gate_proj.weight")) { if (!shape_is(ndim, shape, {c.llm_ffn, DL})) return bad_shape(); rc = pack_into(h, L.gateup, src, c.llm_ffn, 0, 2, 0, 0); }
        if (!strcmp(suf, "mlp.up_proj.weight")) { if (!shape_is(ndim, shape, {c.llm_ffn, DL})) return bad_shape(); rc = pack_into(h, L.gateup, src, c.llm_ffn, 0, 2, 1, 0); }
        if (!strcmp(suf, "mlp.down_proj.weight")) { if (!shape_is(ndim, shape, {DL, c.llm_ffn})) return bad_shape(); rc = pack_into(h, L.down, src, DL, 0, 1, 0, 0); }
    }
    if (rc == ISST_ERR_NOTFOUND) return h->fail(rc, "tensor %s is not part of the hot path", name);
    if (rc != ISST_OK) return rc;
    h->loaded.insert(nm);
    h->finalized = false;
    return ISST_OK;
}

extern "C" int isst_set_rope_tables(isst_handle* h, const float* enc_cos, const float* enc_sin, int enc_rows, const uint16_t* llm_cos,
                                    const uint16_t* llm_sin, int llm_rows) {
    if (!h || !enc_cos || !enc_sin || !llm_cos || !llm_sin) return h ? h->fail(ISST_ERR_ARG, "null rope table") : ISST_ERR_ARG;
    if (enc_rows < h->enc_rope_rows || llm_rows < h->llm_rope_rows)
        return h->fail(ISST_ERR_ARG, "rope tables too short: need %d encoder rows and %d llm rows", h->enc_rope_rows, h->llm_rope_rows);
    HIPCHK(hipMemcpy(h->enc_cos, enc_cos, (size_t)h->enc_rope_rows * 32 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->enc_sin, enc_sin, (size_t)h->enc_rope_rows * 32 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->llm_cos, llm_cos, (size_t)h->llm_rope_rows * 64 * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->llm_sin, llm_sin, (size_t)h->llm_rope_rows * 64 * 2, hipMemcpyHostToDevice));
    h->rope_set = true;
    return ISST_OK;
}

// position held by row `row` of the --rope 0 table (rows above 255 are consecutive bf16 bit patterns from 256.0 = 0x4380)
static float enc_pos_row_value(int row) {
    if (row < 256) return (float)row;
    const uint32_t bits = (uint32_t)(0x4380 + row - 256) << 16;
    float f;
    std::memcpy(&f, &bits, sizeof f);
    return f;
}
extern "C" int isst_set_enc_position_table(isst_handle* h, const uint16_t* table, int rows) {
    if (!h || !table) return h ? h->fail(ISST_ERR_ARG, "isst_set_enc_position_table: null table") : ISST_ERR_ARG;
    if (!h->cfg.enc_abs_pos) return h->fail(ISST_ERR_STATE, "isst_set_enc_position_table: the handle was created with rotary positions (enc_abs_pos 0)");
    if (rows < 257 || rows > ISST_ENC_POS_ROWS) return h->fail(ISST_ERR_ARG, "isst_set_enc_position_table: %d rows, expected 257..%d", rows, ISST_ENC_POS_ROWS);
    if (!h->enc_pos) {
        h->enc_pos = h->dalloc<bf16_t>((size_t)ISST_ENC_POS_ROWS * h->cfg.enc_dim, true);
        if (!h->enc_pos) return h->fail(ISST_ERR_NOMEM, "isst_set_enc_position_table: device allocation failed");
    }
    HIPCHK(hipMemcpy(h->enc_pos, table, (size_t)rows * h->cfg.enc_dim * sizeof(bf16_t), hipMemcpyHostToDevice));
    h->enc_pos_rows = rows;
    return ISST_OK;
}

extern "C" int isst_finalize_weights(isst_handle* h) {
    if (!h) return ISST_ERR_ARG;
    for (const auto& n : h->expected)
        if (!h->loaded.count(n)) return h->fail(ISST_ERR_STATE, "missing tensor %s", n.c_str());
    if (!h->rope_set) return h->fail(ISST_ERR_STATE, "rotary tables not set (isst_set_rope_tables)");
    if (h->cfg.enc_abs_pos && !h->enc_pos_rows) return h->fail(ISST_ERR_STATE, "enc_abs_pos is set and the position table is not (isst_set_enc_position_table)");
    HIPCHK(hipDeviceSynchronize());
    if (h->stage) { (void)hipFree(h->stage); h->stage = nullptr; h->stage_bytes = 0; }
    h->finalized = true;
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// streams
// --------------------------------------------------------------------------------------------
extern "C" int isst_stream_open(isst_handle* h, int* stream_id) {
    if (!h || !stream_id) return ISST_ERR_ARG;
    for (size_t i = 0; i < h->streams.size(); ++i)
        if (!h->streams[i].open) {
            h->streams[i] = StreamState();
            h->streams[i].open = true;
            *stream_id = (int)i;
            return isst_stream_reset(h, (int)i);
        }
    return h->fail(ISST_ERR_STATE, "all %d stream slots are open", (int)h->streams.size());
}
extern "C" int isst_stream_reset(isst_handle* h, int id) {
    if (!h || id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h ? h->fail(ISST_ERR_ARG, "bad stream id %d", id) : ISST_ERR_ARG;
    h->streams[id] = StreamState();
    h->streams[id].open = true;
    // first-chunk offset: 79 + 320 zeros in front of the first samples (agents/infinisst.py:216-218)
    HIPCHK(hipMemsetAsync(h->audio_hist + (size_t)id * round_up(h->hist, 8), 0, (size_t)h->hist * 2, 0));
    HIPCHK(hipStreamSynchronize(0));
    return ISST_OK;
}
extern "C" int isst_stream_close(isst_handle* h, int id) {
    if (!h || id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h ? h->fail(ISST_ERR_ARG, "bad stream id %d", id) : ISST_ERR_ARG;
    h->streams[id].open = false;
    return ISST_OK;
}
extern "C" int isst_stream_info_get(isst_handle* h, int id, isst_stream_info* out) {
    if (!h || !out || id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h ? h->fail(ISST_ERR_ARG, "bad stream id %d", id) : ISST_ERR_ARG;
    const StreamState& s = h->streams[id];
    out->llm_cache_len = s.llm_total;
    out->llm_sys_len = s.llm_sys < s.llm_total ? s.llm_sys : s.llm_total;
    out->enc_n_steps = s.enc_steps;
    out->enc_cache_len = s.enc_len;
    out->chunks = s.chunks;
    return ISST_OK;
}

extern "C" int isst_kv_evict(isst_handle* h, int id, int new_cache_size, int keep_prefix) {
    if (!h || id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h ? h->fail(ISST_ERR_ARG, "bad stream id %d", id) : ISST_ERR_ARG;
    StreamState& s = h->streams[id];
    if (new_cache_size < 0 || keep_prefix < 0) return h->fail(ISST_ERR_ARG, "negative size");
    if (keep_prefix != 0 && keep_prefix != s.llm_sys)
        return h->fail(ISST_ERR_STATE, "keep_prefix %d differs from the pinned system prompt (%d entries); pin it with gen_params.system_prompt_size on the first chunk", keep_prefix, s.llm_sys);
    if (s.llm_total < s.llm_sys) return h->fail(ISST_ERR_STATE, "cache shorter than its pinned prefix");
    const int ring_len = s.llm_total - s.llm_sys;
    if (new_cache_size > ring_len)
        return h->fail(ISST_ERR_STATE, "new_cache_size %d exceeds the %d evictable entries (overlap with the pinned prefix is undefined in the reference)", new_cache_size, ring_len);
    if (keep_prefix == 0) s.llm_sys = 0;  // nothing pinned any more: logical position 0 is the ring start
    const int drop = ring_len - new_cache_size;
    s.llm_ring_start = (s.llm_ring_start + drop) % h->ring_cap;
    s.llm_total = s.llm_sys + new_cache_size;
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// debug taps
// --------------------------------------------------------------------------------------------
namespace {
int tap(isst_handle* h, const std::string& name, const bf16_t* src, int64_t elems, hipStream_t st) {
    if (!h->cfg.debug_taps) return ISST_OK;
    Tap& t = h->taps[name];
    if (t.cap < elems) {
        HIPCHK(hipStreamSynchronize(st));
        if (t.dev) (void)hipFree(t.dev);
        t.dev = nullptr; t.cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&t.dev), (size_t)elems * 2));
        t.cap = elems;
    }
    t.elems = elems;
    HIPCHK(hipMemcpyAsync(t.dev, src, (size_t)elems * 2, hipMemcpyDeviceToDevice, st));
    return ISST_OK;
}

int gemm(isst_handle* h, const bf16_t* A, long lda, const PackedLinear& L, int epi, const bf16_t* res, long ldres, void* out, long ldo,
         int M, hipStream_t st, int batch = 1, long a_batch = 0, long out_batch = 0, long res_batch = 0,
         const bf16_t* norm_w = nullptr, float norm_eps = 0.f, float* ssq = nullptr) {
    GemmArgs g{};
    g.ssq = ssq; g.ssq_n = ssq ? L.K / 32 : 0;
    g.A = A; g.lda = lda; g.a_batch = a_batch;
    g.Wp = L.wp; g.bias = L.bias;
    g.res = res; g.ldres = ldres; g.res_batch = res_batch;
    g.out = out; g.ldo = ldo; g.out_batch = out_batch;
    g.M = M; g.N = L.N; g.K = L.K; g.batch = batch; g.epi = epi; g.n_valid = L.n_valid;
    g.norm_w = norm_w; g.norm_eps = norm_eps;
    CHK(launch_gemm(g, st));
    return ISST_OK;
}

// K slices over workgroups for the narrow-N projections (o_proj, down_proj).  17..64 rows (gemm_mid.hip; measured,
// profiles/r01/mid_probe.txt): 2 slices for K = 4096, 4 for K = 14336, more only add slab traffic.  65..512 rows (gemm_tiled.hip:
// 32 column blocks x ceil(rows / 128) row blocks walk all of K alone otherwise -- 217 us for down_proj whatever the row count,
// profiles/rows_probe.py): enough slices for ~384 workgroups.
// (the dense kernel keeps 3 workgroups per CU resident, 768 chip-wide: slicing up to that count instead of 384 and up to 2048 rows instead of
//  1024 gave, same box, 64 / 32 / 16 / 8 streams 100.6 -> 99.7 / 71.5 -> 70.0 / 53.8 -> 53.4 / 45.9 -> 45.3 ms per chunk; 1152 was worse again at 16-32)
int pick_ksplit(int K, int N, int rows, long slab_cap) {
    if (rows <= 64) {
        // (N <= 2048 = the encoder's out_proj / fc2, 16..32 column blocks: fc2 20.1 us as GEMM + LayerNorm, 16.6 / 14.7 / 16.5 us with 2 / 4 / 8 slices
        //  + the reducing LayerNorm; out_proj 14.2 -> 11.5 us with 2: profiles/enc_probe.py)
        // (33..64 rows, N = K = 4096 -- o_proj of a many-stream decode pass -- with the in-launch reduction: 2 slices of 32 columns 16.8 us, 4 slices of
        //  64 columns 15.3; down_proj stays at 4 x 64 columns: 29.2 against 34.7 for 8 and 45.0 for 2; profiles/fused_ks_probe.py)
        for (int s = (K >= 8192 || (N <= 2048 && K >= 4096) || (rows > 32 && K >= 4096)) ? 4 : 2; s > 1; s >>= 1)
            if (K % (256 * s) == 0 && K / (256 * s) >= 1) return s;
        return 1;
    }
    // (slab_cap: fp32 elements of the caller's slab buffer -- isst_handle::lslab_elems; a slice count is only chosen if its slabs fit)
    if (gemm_wide_enabled() && rows <= 256 && (long)N * K >= (8L << 20)) {  // (= gemm_wide_preferred: shorter weight streams keep gemm_tiled and its slice choice below)
        // gemm_wide.hip (65..256 rows: one 8-wave workgroup per CU, 128 columns, all rows): enough K slices to give most of the 256 CUs a workgroup and
        // no more -- a second round of workgroups doubles the launch (profiles/r04/wide_probe.txt, GEMM + reducing norm, us, 128 rows: q/k/v 38.7 / 29.1 /
        // 25.6 / 33.1 for 1 / 2 / 4 / 8 slices (48 column blocks); down_proj 45.7 / 35.7 for 4 / 8 (32 column blocks))
        const long blocks = (N + 127) / 128;
        int s = 1;
        // ... and no fewer than 16 K-steps per slice: the launch's fixed cost (ring fill 3.3 us, epilogue + slab traffic) is not amortised below that
        // (o_proj, K = 4096: 21.6 us with 4 slices = 128 workgroups against 22.4 with 8 = 256 at 128 rows, 27.3 / 30.5 at 256: profiles/r04/wide_probe_v4*)
        while (s < LLM_KSPLIT_MAX && blocks * s * 2 <= 256 && (long)s * 2 * rows * N <= slab_cap && K % (64 * s * 2) == 0 && K / (64 * s * 2) >= 16) s *= 2;
        return s;
    }
    if (gemm_dense_would_run(rows, N, K)) {
        // 256 x 256 tiles, one workgroup per CU (gemm_dense.hip): rounds of 1/s-length tiles + the slab traffic each further slice adds (write + read of
        // rows x N fp32: about 4 % of a round per slice at these shapes).  profiles/dense_split_probe.py, GEMM + reducing norm, us:
        //   1408 rows  o_proj 82.7 / 62.4 / 79.9 / 102.0 for 1 / 2 / 4 / 8 slices, down_proj 241 / 161 / 178 / 183;   704 rows  o_proj 79 / 52 / 43 / 59, down 248 / 139 / 94 / 116
        const long tiles = (long)((N + 255) / 256) * ((rows + 255) / 256);
        int best = 1;
        double best_cost = 1e30;
        for (int s = 1; s <= LLM_KSPLIT_MAX; s *= 2) {
            if (K % (64 * s) != 0 || K / (64 * s) < 4 || (s > 1 && (long)s * rows * N > slab_cap)) break;
            const double cost = (double)((tiles * s + 255) / 256) / s + 0.04 * s;
            if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
        }
        return best;
    }
    const int blocks = ((N + 127) / 128) * ((rows + 127) / 128);
    int s = 1;
    while (s < LLM_KSPLIT_MAX && blocks * s * 2 <= LLM_SPLIT_TARGET_WGS && (long)s * 2 * rows * N <= slab_cap && K % (64 * s * 2) == 0 && K / (64 * s * 2) >= 8) s *= 2;
    return s;
}
// slabs[ksplit][M][N] (fp32) = A @ W^T per K slice; reduced by launch_rmsnorm_reduce
// x != null (13..64 rows): the launch also reduces -- x = bf16(x + bf16(sum of the slabs)) by the last K-slice workgroup of every column block,
// sums of squares of the new x per row and 32 columns into ssq (GemmArgs::tickets)
int gemm_partial(isst_handle* h, const bf16_t* A, long lda, const PackedLinear& L, float* slabs, int M, int ksplit, hipStream_t st,
                 bf16_t* x = nullptr, long ldx = 0, float* ssq = nullptr, bf16_t* plain_out = nullptr, long ld_plain = 0,
                 const bf16_t* norm_w = nullptr, float norm_eps = 0.f, float* ssq_in = nullptr) {
    GemmArgs g{};
    if (x) { g.res = x; g.ldres = ldx; g.ssq = ssq; g.ssq_n = L.n_valid / 32; g.tickets = h->ltickets; }
    if (plain_out) {  // q/k/v in K slices: the last slice workgroup writes bf16(sum) to plain_out; A may be normalised while staged (norm_w + ssq_in)
        g.res = plain_out; g.ldres = ld_plain; g.reduce_plain = 1; g.tickets = h->ltickets;
        g.norm_w = norm_w; g.norm_eps = norm_eps; g.ssq = ssq_in; g.ssq_n = ssq_in ? L.K / 32 : 0;
    }
    g.A = A; g.lda = lda; g.Wp = L.wp;
    g.out = slabs; g.ldo = L.n_valid; g.out_batch = (long)M * L.n_valid;
    g.M = M; g.N = L.N; g.K = L.K; g.batch = 1; g.epi = EPI_PARTIAL; g.n_valid = L.n_valid; g.ksplit = ksplit;
    if (slabs == h->lslab && (long)(ksplit > 1 ? ksplit : 1) * M * L.n_valid > h->lslab_elems)
        return h->fail(ISST_ERR_STATE, "split-K launch of %d slices x %d rows x %d columns exceeds the slab buffer (%ld fp32)", ksplit, M, L.n_valid, h->lslab_elems);
    if (g.tickets && (L.N + 31) / 32 > h->ltickets_n) return h->fail(ISST_ERR_STATE, "ticketed split-K launch over %d columns needs %d arrival counters, %d allocated", L.N, (L.N + 31) / 32, h->ltickets_n);
    CHK(launch_gemm(g, st));
    return ISST_OK;
}

// conv extractor + encoder + shrink + projector for n streams; result in h->speech [n*S][llm_dim]
int run_encoder(isst_handle* h, int n, const int* sids, const float* const* pcm, bool pcm_on_device, int n_samples, int multiplier, hipStream_t st,
                int* out_S) {
    const isst_config& c = h->cfg;
    const int hist = h->hist, win = hist + n_samples, winp = round_up(hist + h->n_new_max, 8);
    const int histp = round_up(hist, 8);
    // ---- audio: [history | new samples] per stream, bf16 (agents/infinisst.py:222) ----
    // (samples and stream ids go up in ONE copy from a pinned staging block; one kernel builds every stream's window.  Audio the caller already
    //  holds in HBM -- isst_gen_params::pcm_on_device -- is read in place: only the n pointers and stream ids go up)
    const int* sids_dev;
    if (pcm_on_device) {
        static_assert(sizeof(const float*) == 8, "pointer table layout");
        std::memcpy(h->pcm_host, pcm, (size_t)n * sizeof(const float*));
        std::memcpy(h->pcm_host + 2 * (size_t)n, sids, (size_t)n * sizeof(int));
        HIPCHK(hipMemcpyAsync(h->pcm_f32, h->pcm_host, (size_t)n * 12, hipMemcpyHostToDevice, st));
        sids_dev = reinterpret_cast<const int*>(h->pcm_f32 + 2 * (size_t)n);
        CHK(launch_audio_window(nullptr, reinterpret_cast<const float* const*>(h->pcm_f32), sids_dev, h->audio_hist, histp, h->window, winp, hist, n_samples, n, st));
    } else {
        for (int i = 0; i < n; ++i) std::memcpy(h->pcm_host + (size_t)i * n_samples, pcm[i], (size_t)n_samples * sizeof(float));
        std::memcpy(h->pcm_host + (size_t)n * n_samples, sids, (size_t)n * sizeof(int));
        HIPCHK(hipMemcpyAsync(h->pcm_f32, h->pcm_host, ((size_t)n * n_samples + n) * sizeof(float), hipMemcpyHostToDevice, st));
        sids_dev = reinterpret_cast<const int*>(h->pcm_f32 + (size_t)n * n_samples);
        CHK(launch_audio_window(h->pcm_f32, nullptr, sids_dev, h->audio_hist, histp, h->window, winp, hist, n_samples, n, st));
    }
    // ---- conv stack ----
    std::vector<int> T(c.n_conv);
    int len = win;
    for (int i = 0; i < c.n_conv; ++i) { len = conv_out_len(len, c.conv_k[i], c.conv_stride[i]); T[i] = len; }
    const int Q = T.back();
    if (Q != n_samples / h->samples_per_frame) return h->fail(ISST_ERR_STATE, "conv stack produced %d frames for %d samples", Q, n_samples);
    bf16_t* cur = h->act_a;
    bf16_t* nxt = h->act_b;
    CHK(launch_conv0(h->window, winp, h->conv[0].w_raw, h->conv[0].lin.bias, h->conv[0].ln.w, h->conv[0].ln.b, cur, (long)T[0] * c.conv_dim[0],
                     T[0], c.conv_dim[0], c.conv_k[0], c.conv_stride[0], n, st));
    for (int i = 1; i < c.n_conv; ++i) {
        const ConvLayer& L = h->conv[i];
        const int cin = h->conv[i - 1].dim;
        CHK(gemm(h, cur, (long)L.stride * cin, L.lin, c.conv_bias ? EPI_BIAS : EPI_NONE, nullptr, 0, nxt, L.dim, T[i], st, n, (long)T[i - 1] * cin,
                 (long)T[i] * L.dim));
        // LN + GELU in place; the last layer writes a dense [n*Q][C] block (out_batch == T*C)
        CHK(launch_layernorm(nxt, L.dim, L.ln.w, L.ln.b, nxt, L.dim, n * T[i], L.dim, 1e-5f, 1, st));
        std::swap(cur, nxt);
    }
    const int cdim = h->conv.back().dim, D = c.enc_dim, ER = n * Q;
    CHK(tap(h, "conv_out", cur, (int64_t)ER * cdim, st));
    // history for the next chunk: last `hist` samples of the window
    CHK(launch_audio_hist_save(h->window, winp, sids_dev, h->audio_hist, histp, hist, win, n, st));
    // ---- LayerNorm + post_extract_proj (patch_speech_encoder.py:268-269,:301) ----
    CHK(launch_layernorm(cur, cdim, h->enc_ln_in.w, h->enc_ln_in.b, nxt, cdim, ER, cdim, 1e-5f, 0, st));
    CHK(gemm(h, nxt, cdim, h->post_proj, EPI_BIAS, nullptr, 0, h->ex, D, ER, st));
    CHK(tap(h, "post_proj", h->ex, (int64_t)ER * D, st));
    // ---- per-stream ring views: trim to max_cache_size before the layer calls (:516-520) ----
    EncStreamView* ev_host = reinterpret_cast<EncStreamView*>(h->meta_host);
    for (int i = 0; i < n; ++i) {
        StreamState& s = h->streams[sids[i]];
        if (s.enc_len > c.max_cache_size) {
            s.enc_start = (s.enc_start + s.enc_len - c.max_cache_size) % h->enc_cap;
            s.enc_len = c.max_cache_size;
        }
        ev_host[i].start = s.enc_start;
        ev_host[i].prefix = s.enc_steps;
    }
    EncStreamView* ev = reinterpret_cast<EncStreamView*>(h->meta_dev);
    HIPCHK(hipMemcpyAsync(ev, ev_host, sizeof(EncStreamView) * n, hipMemcpyHostToDevice, st));
    if (c.enc_abs_pos) {  // --rope 0: the frames' stream positions go into the input instead of into q / k (patch_speech_encoder.py:488-493)
        for (int i = 0; i < n; ++i) {
            const long last = (long)h->streams[sids[i]].enc_steps + Q - 1;
            const long covered = h->enc_pos_rows <= 256 ? h->enc_pos_rows - 1 : (long)enc_pos_row_value(h->enc_pos_rows - 1);
            if (last > covered) return h->fail(ISST_ERR_STATE, "stream %d is at frame %ld, the position table ends at %ld", sids[i], last, covered);
        }
        CHK(launch_enc_add_position(h->ex, ev, h->enc_pos, h->enc_pos_rows, n, Q, D, st));
    }
    const int bs = c.block_size * multiplier;
    // the n streams of a call must be laid out with ONE stream stride between ring bases: use per-stream pointers
    // via a base + sid * stride scheme -> requires contiguous slots; general case: launch per stream
    bool contiguous = true;
    for (int i = 1; i < n; ++i) contiguous = contiguous && (sids[i] == sids[0] + i);
    // 17..1024 rows (1..21 streams): out_proj and fc2 (N = 1024: 16 column blocks on gemm_mid, 8 on the dense kernel) split K into fp32
    // slabs that the LayerNorm which follows anyway sums up (rowops.hip layernorm_kernel's prologue) -- the encoder twin of the decoder's
    // split path
#ifndef ISST_ESPLIT_MIN_ROWS
#define ISST_ESPLIT_MIN_ROWS 16
#endif
    const bool esplit = ER > ISST_ESPLIT_MIN_ROWS && ER <= ENC_SPLIT_MAX_ROWS;
    const int s_out = esplit ? pick_ksplit(D, D, ER, h->lslab_elems) : 1, s_fc2 = esplit ? pick_ksplit(c.enc_ffn, D, ER, h->lslab_elems) : 1;
    const long eslab = (long)ER * D;
    const EncLayer* pend = nullptr;  // layer whose fc2 slabs h->ex still lacks
    for (int l = 0; l < c.enc_layers; ++l) {
        const EncLayer& L = h->enc[l];
        if (pend) {
            CHK(launch_layernorm_reduce(h->lslab, eslab, s_fc2, pend->fc2.bias, h->ex, D, L.ln1.w, L.ln1.b, h->exn, D, ER, D, c.enc_ln_eps, st));
            if (h->cfg.debug_taps) CHK(tap(h, "enc_layer_" + std::to_string(l - 1), h->ex, (int64_t)ER * D, st));
            pend = nullptr;
        } else {
            CHK(launch_layernorm(h->ex, D, L.ln1.w, L.ln1.b, h->exn, D, ER, D, c.enc_ln_eps, 0, st));
        }
        CHK(gemm(h, h->exn, D, L.qkv, EPI_BIAS, nullptr, 0, h->eqkv, 3 * D, ER, st));
        if (contiguous) {
            bf16_t* kb = h->enc_k + (size_t)sids[0] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
            bf16_t* vb = h->enc_v + (size_t)sids[0] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
            CHK(launch_enc_attention(h->eqkv, kb, vb, h->enc_stream_stride, ev, h->enc_cos, h->enc_sin, c.enc_rope_round_each, h->eattn, n, Q,
                                     c.enc_heads, h->enc_cap, c.max_cache_size, bs, st));
        } else {
            for (int i = 0; i < n; ++i) {
                bf16_t* kb = h->enc_k + (size_t)sids[i] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
                bf16_t* vb = h->enc_v + (size_t)sids[i] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
                CHK(launch_enc_attention(h->eqkv + (size_t)i * Q * 3 * D, kb, vb, 0, ev + i, h->enc_cos, h->enc_sin, c.enc_rope_round_each,
                                         h->eattn + (size_t)i * Q * D, 1, Q, c.enc_heads, h->enc_cap, c.max_cache_size, bs, st));
            }
        }
        if (s_out > 1) {
            CHK(gemm_partial(h, h->eattn, D, L.out, h->lslab, ER, s_out, st));
            CHK(launch_layernorm_reduce(h->lslab, eslab, s_out, L.out.bias, h->ex, D, L.ln2.w, L.ln2.b, h->exn, D, ER, D, c.enc_ln_eps, st));
        } else {
            CHK(gemm(h, h->eattn, D, L.out, EPI_BIAS_RES, h->ex, D, h->ex, D, ER, st));
            CHK(launch_layernorm(h->ex, D, L.ln2.w, L.ln2.b, h->exn, D, ER, D, c.enc_ln_eps, 0, st));
        }
        CHK(gemm(h, h->exn, D, L.fc1, EPI_BIAS_GELU, nullptr, 0, h->effn, c.enc_ffn, ER, st));
        if (s_fc2 > 1) {
            CHK(gemm_partial(h, h->effn, c.enc_ffn, L.fc2, h->lslab, ER, s_fc2, st));
            pend = &L;
        } else {
            CHK(gemm(h, h->effn, c.enc_ffn, L.fc2, EPI_BIAS_RES, h->ex, D, h->ex, D, ER, st));
            if (h->cfg.debug_taps) CHK(tap(h, "enc_layer_" + std::to_string(l), h->ex, (int64_t)ER * D, st));
        }
    }
    if (pend) {
        CHK(launch_layernorm_reduce(h->lslab, eslab, s_fc2, pend->fc2.bias, h->ex, D, h->enc_ln_out.w, h->enc_ln_out.b, h->exn, D, ER, D, c.enc_ln_eps, st));
        if (h->cfg.debug_taps) CHK(tap(h, "enc_layer_" + std::to_string(c.enc_layers - 1), h->ex, (int64_t)ER * D, st));
    } else {
        CHK(launch_layernorm(h->ex, D, h->enc_ln_out.w, h->enc_ln_out.b, h->exn, D, ER, D, c.enc_ln_eps, 0, st));
    }
    CHK(tap(h, "enc_out", h->exn, (int64_t)ER * D, st));
    for (int i = 0; i < n; ++i) {
        StreamState& s = h->streams[sids[i]];
        s.enc_len += Q;
        s.enc_steps += Q;
    }
    // ---- length shrink (k == stride: rows [k frames x D] are contiguous) + projector ----
    bf16_t* a = h->exn;
    bf16_t* b = h->eattn;
    int rows = ER;
    for (int i = 0; i < c.n_shrink; ++i) {
        const ConvLayer& L = h->shrink[i];
        rows /= L.k;
        CHK(gemm(h, a, (long)L.k * D, L.lin, EPI_NONE, nullptr, 0, b, D, rows, st));
        CHK(launch_layernorm(b, D, L.ln.w, L.ln.b, b, D, rows, D, 1e-5f, 1, st));
        std::swap(a, b);
        if (b == h->exn) b = h->eqkv;  // keep `a` (current) and `b` distinct scratch buffers
    }
    CHK(tap(h, "shrink", a, (int64_t)rows * D, st));
    CHK(gemm(h, a, D, h->proj, EPI_BIAS, nullptr, 0, h->speech, c.llm_dim, rows, st));
    CHK(tap(h, "speech", h->speech, (int64_t)rows * c.llm_dim, st));
    *out_S = Q / h->shrink_factor;
    return ISST_OK;
}

struct StepMeta {
    int *row_stream, *row_pos, *ids, *speech_row, *last_rows;
    int2* groups;  // attention row groups of this launch
    int2* units;   // prefill: runs of <= 8 consecutive groups of one stream (llm_attn_prefill_kernel)
    LlmStreamView* views;
    SampleStream* samp;
    int *ids_pool, *enc_pool, *suppress;
    KvCopyOp* ops;  // position copies of a beam step
    size_t step_bytes;      // bytes from the block start up to (not including) the suppress list
    size_t suppress_offset;
};
// carve the metadata block (same offsets on host and device)
StepMeta carve(isst_handle* h, unsigned char* base) {
    StepMeta m;
    const size_t LR = h->llm_rows_max, ns = (size_t)h->cfg.max_streams * h->max_beams;
    unsigned char* p = base + 4096;  // first 4 KiB: encoder views
    auto take = [&](size_t bytes) { unsigned char* r = p; p += (bytes + 15) / 16 * 16; return r; };
    m.row_stream = reinterpret_cast<int*>(take(LR * 4)); m.row_pos = reinterpret_cast<int*>(take(LR * 4));
    m.ids = reinterpret_cast<int*>(take(LR * 4)); m.speech_row = reinterpret_cast<int*>(take(LR * 4));
    m.last_rows = reinterpret_cast<int*>(take(ns * 4));
    m.groups = reinterpret_cast<int2*>(take(LR * 8));
    m.units = reinterpret_cast<int2*>(take(LR * 8));
    m.views = reinterpret_cast<LlmStreamView*>(take(ns * sizeof(LlmStreamView)));
    m.samp = reinterpret_cast<SampleStream*>(take(ns * sizeof(SampleStream)));
    m.ids_pool = reinterpret_cast<int*>(take(ns * h->max_ids * 4));
    m.enc_pool = reinterpret_cast<int*>(take(ns * h->max_enc_ids * 4));
    m.ops = reinterpret_cast<KvCopyOp*>(take(ns * 4 * sizeof(KvCopyOp) * KV_OPS_SLOTS));
    m.step_bytes = (size_t)(p - base);
    m.suppress_offset = m.step_bytes;
    m.suppress = reinterpret_cast<int*>(take(65536 * 4));
    return m;
}

// one forward pass of the decoder stack over `rows` token rows, logits for `n_last` rows
int llm_forward(isst_handle* h, const StepMeta& d, int rows, int n_last, int n_groups, int max_group_rows, bool splice, const char* tap_prefix,
                hipStream_t st, const StepMeta* hm = nullptr, int n_units = 0, int max_unit_groups = 0, int n_beam_wgs = 0) {
    const isst_config& c = h->cfg;
    const int DL = c.llm_dim, H = c.llm_heads, KV = c.llm_kv_heads;
    // one group (one stream's decode step): its metadata travels in the kernel arguments (llm_attn.hip LlmAttnOne)
    LlmAttnOne one{};
    if (hm && n_groups == 1) {
        const int2 g0 = hm->groups[0];
        bool consecutive = true;
        for (int k = 1; k < g0.y; ++k) consecutive = consecutive && hm->row_pos[g0.x + k] == hm->row_pos[g0.x] + k;
        if (consecutive) {
            one.enabled = 1;
            one.grp = g0;
            one.pos0 = hm->row_pos[g0.x];
            one.v = hm->views[hm->row_stream[g0.x]];
        }
    }
    CHK(launch_embed_splice(d.ids, splice ? d.speech_row : nullptr, h->embed, h->speech, h->lx, rows, DL, st));
    if (tap_prefix) CHK(tap(h, std::string(tap_prefix) + "embed", h->lx, (int64_t)rows * DL, st));
    // 17..64 rows (one stream's prefill, a 64-stream decode pass): o_proj and down_proj split K over workgroups and the
    // residual + RMSNorm kernel that follows reduces the slabs (gemm_mid.hip); `pending`: lx still lacks the previous
    // layer's down_proj slabs
    const bool split_rows = rows > ISST_MID_MIN_ROWS && rows <= LLM_SPLIT_MAX_ROWS;
    const int so = split_rows ? pick_ksplit(H * 128, DL, rows, h->lslab_elems) : 1, sd = split_rows ? pick_ksplit(c.llm_ffn, DL, rows, h->lslab_elems) : 1;
    const int sq = (rows > (gemm_wide_enabled() ? 64 : 128) && rows <= LLM_SPLIT_MAX_ROWS) ? pick_ksplit(DL, (H + 2 * KV) * 128, rows, h->lslab_elems) : 1;  // (65..256 rows: gemm_wide's 48 column blocks in 4 slices)
    const long slab = (long)rows * DL;
    // 13..64 rows: no residual + RMSNorm launches (gemm_mid.hip: the producer reduces, the consumer normalises while staging)
    const bool fr = h->fuse_reduce && split_rows && rows <= 64 && so > 1 && sd > 1 && DL % 128 == 0;
    // ... and q/k/v in K slices with the same in-launch reduction (192 workgroups of 32 columns leave a quarter of the CUs idle and give the others one
    // workgroup each; ISST_QKV_SLICES=1 forces the unsplit launch)
    // A/B on one box, ms per step: 64 streams 92.63 (1 slice) / 91.33 (2) / 92.61 (4); 16 streams 51.24 / 51.52 / 50.84 -- two slices from 33 rows
    const int qs = h->qkv_slices > 0 ? h->qkv_slices : (rows > 32 ? 2 : 1);
    const int sqf = (fr && qs > 1 && DL % (256 * qs) == 0) ? qs : 1;
    bool pending = false, pending_fused = false;
    // in-situ timing (isst_profile_begin / _begin_rows): an event pair around the gate/up launch of every layer of a pass whose row count is in the
    // profiled range -- opens the bracket and hands back the event that closes it
    auto prof_open = [&](hipEvent_t* close) -> int {
        *close = nullptr;
        if (!h->prof_on || rows < h->prof_rows_lo || rows > h->prof_rows_hi) return ISST_OK;
        if (h->prof_used + 2 > h->prof_ev.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                HIPCHK(hipEventCreate(&e));
                h->prof_ev.push_back(e);
            }
        }
        HIPCHK(hipEventRecord(h->prof_ev[h->prof_used], st));
        *close = h->prof_ev[h->prof_used + 1];
        h->prof_used += 2;
        return ISST_OK;
    };
    for (int l = 0; l < c.llm_layers; ++l) {
        const LlmLayer& L = h->llm[l];
        hipEvent_t pe = nullptr;
        // decode shapes: RMSNorm is applied inside the projection's A-fragment load (gemm.hip NORM); larger row counts
        // (prefill, many streams) run the norm kernel once instead of once per workgroup
        const bool fuse = rows <= LLM_FUSED_NORM_MAX_ROWS;
        if (fuse) {
            CHK(gemm(h, h->lx, DL, L.qkv, EPI_NONE, nullptr, 0, h->lqkv, (H + 2 * KV) * 128, rows, st, 1, 0, 0, 0, L.in_norm, c.rms_eps));
        } else {
            bool qkv_done = false;
            if (pending_fused) {  // lx is complete (the down_proj launch reduced its own slabs); q/k/v normalises it on the way into LDS
                if (tap_prefix && h->cfg.debug_taps) CHK(tap(h, std::string(tap_prefix) + "layer_" + std::to_string(l - 1), h->lx, (int64_t)rows * DL, st));
                pending_fused = false;
                if (sqf > 1)
                    CHK(gemm_partial(h, h->lx, DL, L.qkv, h->lslab, rows, sqf, st, nullptr, 0, nullptr, h->lqkv, (H + 2 * KV) * 128, L.in_norm, c.rms_eps, h->lssq));
                else
                    CHK(gemm(h, h->lx, DL, L.qkv, EPI_NONE, nullptr, 0, h->lqkv, (H + 2 * KV) * 128, rows, st, 1, 0, 0, 0, L.in_norm, c.rms_eps, h->lssq));
                qkv_done = true;
            } else if (pending) {
                CHK(launch_rmsnorm_reduce(h->lslab, slab, sd, h->lx, DL, L.in_norm, h->lxn, DL, rows, DL, c.rms_eps, st));
                if (tap_prefix && h->cfg.debug_taps) CHK(tap(h, std::string(tap_prefix) + "layer_" + std::to_string(l - 1), h->lx, (int64_t)rows * DL, st));
                pending = false;
            } else {
                CHK(launch_rmsnorm(h->lx, DL, nullptr, L.in_norm, h->lxn, DL, rows, DL, c.rms_eps, st));
            }
            if (qkv_done) {
            } else if (sqf > 1) {  // (first layer: the norm launch ran; the K slices still pay)
                CHK(gemm_partial(h, h->lxn, DL, L.qkv, h->lslab, rows, sqf, st, nullptr, 0, nullptr, h->lqkv, (H + 2 * KV) * 128));
            } else if (sq > 1) {  // 129..1024 rows: the 48 column blocks of the dense kernel get K slices; a small pass sums the slabs to bf16
                CHK(gemm_partial(h, h->lxn, DL, L.qkv, h->lslab, rows, sq, st));
                CHK(launch_slab_reduce(h->lslab, (long)rows * (H + 2 * KV) * 128, sq, h->lqkv, (H + 2 * KV) * 128, rows, (H + 2 * KV) * 128, st));
            } else {
                CHK(gemm(h, h->lxn, DL, L.qkv, EPI_NONE, nullptr, 0, h->lqkv, (H + 2 * KV) * 128, rows, st));
            }
        }
        // one or two decode rows: no combine launch -- o_proj merges the split-KV partials while it stages its A row (gemm.hip AMODE 3)
        int merge_splits = 0;
        const bool merge_in_oproj = h->fuse_combine && !h->inline_combine && so == 1 && rows <= ATTN_MERGE_MAX_ROWS && n_units == 0 && n_beam_wgs == 0;
        CHK(launch_llm_attention(h->lqkv, d.row_stream, d.row_pos, d.views, d.groups, n_groups, max_group_rows, h->llm_cos, h->llm_sin, h->llm_k,
                                 h->llm_kr, h->llm_v, h->lpartial, h->lattn, h->adims, l, rows, st, &one, n_units > 0 ? d.units : nullptr, n_units,
                                 max_unit_groups, n_beam_wgs, merge_in_oproj ? &merge_splits : nullptr, h->inline_combine ? h->attn_cnt : nullptr));
        if (so > 1 && fr) {
            CHK(gemm_partial(h, h->lattn, H * 128, L.o, h->lslab, rows, so, st, h->lx, DL, h->lssq));
            CHK(prof_open(&pe));
            CHK(gemm(h, h->lx, DL, L.gateup, EPI_SWIGLU, nullptr, 0, h->lact, c.llm_ffn, rows, st, 1, 0, 0, 0, L.post_norm, c.rms_eps, h->lssq));
            if (pe) HIPCHK(hipEventRecord(pe, st));
        } else if (so > 1) {
            CHK(gemm_partial(h, h->lattn, H * 128, L.o, h->lslab, rows, so, st));
            CHK(launch_rmsnorm_reduce(h->lslab, slab, so, h->lx, DL, L.post_norm, h->lxn, DL, rows, DL, c.rms_eps, st));
            CHK(prof_open(&pe));
            CHK(gemm(h, h->lxn, DL, L.gateup, EPI_SWIGLU, nullptr, 0, h->lact, c.llm_ffn, rows, st));
            if (pe) HIPCHK(hipEventRecord(pe, st));
        } else {
            if (merge_splits > 0) {
                GemmArgs g{};
                g.A = h->lattn; g.lda = H * 128; g.Wp = L.o.wp; g.res = h->lx; g.ldres = DL; g.out = h->lx; g.ldo = DL;
                g.M = rows; g.N = L.o.N; g.K = L.o.K; g.batch = 1; g.epi = EPI_RES; g.n_valid = L.o.n_valid;
                g.attn_partial = h->lpartial; g.attn_splits = merge_splits;
                CHK(launch_gemm(g, st));
            } else {
                CHK(gemm(h, h->lattn, H * 128, L.o, EPI_RES, h->lx, DL, h->lx, DL, rows, st));
            }
            if (fuse) {
                CHK(prof_open(&pe));
                CHK(gemm(h, h->lx, DL, L.gateup, EPI_SWIGLU, nullptr, 0, h->lact, c.llm_ffn, rows, st, 1, 0, 0, 0, L.post_norm, c.rms_eps));
                if (pe) HIPCHK(hipEventRecord(pe, st));
            } else {
                CHK(launch_rmsnorm(h->lx, DL, nullptr, L.post_norm, h->lxn, DL, rows, DL, c.rms_eps, st));
                CHK(prof_open(&pe));
                CHK(gemm(h, h->lxn, DL, L.gateup, EPI_SWIGLU, nullptr, 0, h->lact, c.llm_ffn, rows, st));
                if (pe) HIPCHK(hipEventRecord(pe, st));
            }
        }
        if (sd > 1 && fr) {  // (the last layer too: lx is complete when the launch ends, and lm_head can take the sums of squares)
            CHK(gemm_partial(h, h->lact, c.llm_ffn, L.down, h->lslab, rows, sd, st, h->lx, DL, h->lssq));
            pending_fused = l + 1 < c.llm_layers;
            if (!pending_fused && tap_prefix && h->cfg.debug_taps) CHK(tap(h, std::string(tap_prefix) + "layer_" + std::to_string(l), h->lx, (int64_t)rows * DL, st));
        } else if (sd > 1) {  // (the last layer too: the slabs are summed by the final norm's launch below -- unsplit, its 32 column blocks walk all of K alone)
            CHK(gemm_partial(h, h->lact, c.llm_ffn, L.down, h->lslab, rows, sd, st));
            pending = true;
        } else {
            CHK(gemm(h, h->lact, c.llm_ffn, L.down, EPI_RES, h->lx, DL, h->lx, DL, rows, st));
            if (tap_prefix && h->cfg.debug_taps) CHK(tap(h, std::string(tap_prefix) + "layer_" + std::to_string(l), h->lx, (int64_t)rows * DL, st));
        }
    }
    bool final_normed = false;
    if (pending) {  // the last layer's down_proj slabs: lx += sum; on a decode pass (every row is a last row) the same launch writes the final norm
        const bool fold = !splice && n_last == rows && n_last > LLM_FUSED_NORM_MAX_ROWS_LM_HEAD;
        CHK(launch_rmsnorm_reduce(h->lslab, slab, sd, h->lx, DL, fold ? h->final_norm : nullptr, h->llast, DL, rows, DL, c.rms_eps, st));
        if (tap_prefix && h->cfg.debug_taps) CHK(tap(h, std::string(tap_prefix) + "layer_" + std::to_string(c.llm_layers - 1), h->lx, (int64_t)rows * DL, st));
        pending = false;
        final_normed = fold;
    }
    if (final_normed) {
        CHK(gemm(h, h->llast, DL, h->lm_head, EPI_F32, nullptr, 0, h->logits, h->vocab_pad, n_last, st));
    } else if (splice) {  // prefill: the last prompt row of every stream is gathered and normalised
        CHK(launch_rmsnorm(h->lx, DL, d.last_rows, h->final_norm, h->llast, DL, n_last, DL, c.rms_eps, st));
        if (tap_prefix) CHK(tap(h, std::string(tap_prefix) + "final", h->llast, (int64_t)n_last * DL, st));
        CHK(gemm(h, h->llast, DL, h->lm_head, EPI_F32, nullptr, 0, h->logits, h->vocab_pad, n_last, st));
    } else if (n_last <= LLM_FUSED_NORM_MAX_ROWS_LM_HEAD) {  // decode: rows == last rows, final norm fused into the lm_head projection
        CHK(gemm(h, h->lx, DL, h->lm_head, EPI_F32, nullptr, 0, h->logits, h->vocab_pad, n_last, st, 1, 0, 0, 0, h->final_norm, c.rms_eps));
    } else if (fr && n_last == rows) {  // 13..64 decode rows: the last down_proj launch left the sums of squares; lm_head normalises while it stages
        CHK(gemm(h, h->lx, DL, h->lm_head, EPI_F32, nullptr, 0, h->logits, h->vocab_pad, n_last, st, 1, 0, 0, 0, h->final_norm, c.rms_eps, h->lssq));
    } else {
        CHK(launch_rmsnorm(h->lx, DL, nullptr, h->final_norm, h->llast, DL, n_last, DL, c.rms_eps, st));
        CHK(gemm(h, h->llast, DL, h->lm_head, EPI_F32, nullptr, 0, h->logits, h->vocab_pad, n_last, st));
    }
    return ISST_OK;
}

// The speech splice as a row map.  The reference rebuilds the embedding sequence with torch.cat of three SLICES per (user, assistant) header
// pair (model/llm.py:86-113):   filled = cat(filled[:u+3], speech[index : index + (a-u-5)], filled[a-2:])
// Slices clamp, so when the encoder produced FEWER features than the prompt has patch slots -- the padded last segment of an utterance at
// multiplier m > 1 brings 12..12(m-1) features for 12m slots -- the sequence simply gets shorter: the surplus patch rows never reach the
// decoder; surplus FEATURES are dropped (:105).  Restated literally on row descriptors: desc[t] >= 0: prompt token desc[t]; < 0: feature -1 - desc[t].
int splice_rows(const int* ids, int len, int user_id, int assistant_id, int start_header_id, int S, std::vector<int>& desc) {
    std::vector<int> users, assists;
    for (int t = 1; t < len; ++t) {
        if (ids[t - 1] != start_header_id) continue;
        if (ids[t] == user_id) users.push_back(t);
        if (ids[t] == assistant_id) assists.push_back(t);
    }
    desc.resize(len);
    for (int t = 0; t < len; ++t) desc[t] = t;
    int index = 0;
    for (size_t q = 0; q < users.size() && q < assists.size(); ++q) {
        const int u = users[q], a = assists[q], cnt = a - u - 5;
        if (cnt < 0) return ISST_ERR_ARG;
        const int cur = (int)desc.size();
        const int head = std::min(u + 3, cur), tail = std::min(std::max(a - 2, 0), cur);
        std::vector<int> nd(desc.begin(), desc.begin() + head);
        for (int k = std::min(index, S); k < std::min(index + cnt, S); ++k) nd.push_back(-1 - k);
        nd.insert(nd.end(), desc.begin() + tail, desc.end());
        desc.swap(nd);
        index += cnt;
    }
    return ISST_OK;
}

int check_ready(isst_handle* h) {
    if (!h->finalized) return h->fail(ISST_ERR_STATE, "weights not finalized (isst_finalize_weights)");
    return ISST_OK;
}

}  // namespace


// --------------------------------------------------------------------------------------------
// beam search (reference model/patches/patch_hf.py:43-302, 687-967; production decoding mode, beam = 4)
// --------------------------------------------------------------------------------------------
namespace {

struct BeamHyp {
    double score;
    std::vector<int> tokens;  // prompt + generated, without the EOS that closed it
    int fed;                  // generated tokens whose KV exists (positions P0 .. P0 + fed - 1)
    int buf;                  // tail buffer slot holding that KV, or -1 - beam when it still sits in arena `beam`
};
struct BeamHyps {  // BeamHypotheses, patch_hf.py:278-302 + [3P] is_done (early_stopping = False)
    int num_beams;
    double length_penalty;
    std::vector<BeamHyp> beams;
    double worst = 1e9;
    // returns the buffer slot freed by an evicted (or rejected) hypothesis, or -1
    int add(BeamHyp hyp, double sum_logprobs, int generated_len) {
        hyp.score = sum_logprobs / std::pow((double)generated_len, length_penalty);
        if ((int)beams.size() < num_beams || hyp.score > worst) {
            beams.push_back(std::move(hyp));
            if ((int)beams.size() > num_beams) {
                std::vector<std::pair<double, int>> ranked;
                for (size_t i = 0; i < beams.size(); ++i) ranked.push_back({beams[i].score, (int)i});
                std::sort(ranked.begin(), ranked.end());
                const int freed = beams[ranked[0].second].buf;
                beams.erase(beams.begin() + ranked[0].second);
                worst = ranked[1].first;
                return freed >= 0 ? freed : -1;
            }
            worst = std::min(hyp.score, worst);
            return -1;
        }
        return hyp.buf >= 0 ? hyp.buf : -1;
    }
    bool is_done(double best_sum_logprobs, int cur_len, int prompt_len) const {
        if ((int)beams.size() < num_beams) return false;
        const double highest = best_sum_logprobs / std::pow((double)(cur_len - prompt_len), length_penalty);
        return worst >= highest;
    }
};

struct BeamStream {  // host state of one stream during a beam call
    std::vector<std::vector<int>> seq;  // per beam: prompt + generated
    std::vector<float> score;
    BeamHyps hyps;
    bool done = false;
    std::vector<int> free_bufs;
};

void push_copy(isst_handle* h, std::vector<KvCopyOp>& ops, int sid, int beam, int buf, int p0, int count, bool to_arena) {
    if (count <= 0) return;
    const StreamState& s = h->streams[sid];
    KvCopyOp op{};
    op.arena_offset = h->arena_off(sid, beam);
    op.buf_offset = ((long)sid * h->nbuf + buf) * h->tbuf_stride;
    op.p0 = p0; op.count = count;
    op.sys_len = s.llm_sys; op.ring_start = s.llm_ring_start;
    op.to_arena = to_arena ? 1 : 0;
    ops.push_back(op);
}
// enqueue `ops` (all of them are independent of each other) and clear the list.  The pinned op list has KV_OPS_SLOTS slots used in
// turn, so that a batch does not have to wait for the previous one's upload: the caller synchronises the stream once per beam step
// (candidate download) and calls kv_ops_synced(); a slot is only reused after such a point
int flush_copies(isst_handle* h, std::vector<KvCopyOp>& ops, const StepMeta& mh, const StepMeta& md, hipStream_t st) {
    if (ops.empty()) return ISST_OK;
    const size_t cap = (size_t)h->cfg.max_streams * h->max_beams * 4;
    for (size_t o = 0; o < ops.size(); o += cap) {
        const int n = (int)std::min(cap, ops.size() - o);
        int max_count = 0;
        for (int i = 0; i < n; ++i) max_count = std::max(max_count, ops[o + i].count);
        if (h->kv_ops_used >= KV_OPS_SLOTS) {  // every slot may still be read by an upload in flight
            HIPCHK(hipStreamSynchronize(st));
            h->kv_ops_used = 0;
        }
        const size_t slot = (size_t)h->kv_ops_used++ * cap;
        std::memcpy(mh.ops + slot, ops.data() + o, sizeof(KvCopyOp) * n);
        HIPCHK(hipMemcpyAsync(md.ops + slot, mh.ops + slot, sizeof(KvCopyOp) * n, hipMemcpyHostToDevice, st));
        CHK(launch_kv_positions_copy(h->llm_k, h->llm_v, h->rot_keys ? h->llm_kr : nullptr, h->tbuf_k, h->tbuf_v, h->tbuf_kr, md.ops + slot, n, max_count, h->adims,
                                     h->cfg.llm_layers, h->tcap, st));
    }
    ops.clear();
    return ISST_OK;
}

// decode phase of a beam call; the prefill (on arena 0 of every stream) has already produced h->logits rows 0..n-1
int beam_decode(isst_handle* h, const isst_gen_params* p, int n, const int* stream_ids, const int* const* prompt_ids, const int* prompt_lens,
                const std::vector<int>& rows_len /* KV entries the prompt wrote (<= prompt_lens after a short splice) */, const int* const* prev_target_ids, const int* n_prev, const std::vector<int>& total0, int* const* out_ids, int* out_lens,
                StepMeta& mh, StepMeta& md, hipStream_t st) {
    const isst_config& c = h->cfg;
    const int B = p->num_beams, V = c.vocab;
    const int n_keep = std::max(2, 1 + c.n_eos) * B;
    if (n_keep > BEAM_TOPK) return h->fail(ISST_ERR_ARG, "beam search keeps %d candidates per step, at most %d are supported", n_keep, BEAM_TOPK);
    const double lp = p->length_penalty == 0.f ? 1.0 : (double)p->length_penalty;
    std::vector<BeamStream> bs(n);
    std::vector<KvCopyOp> ops;
    if (h->kv_ops_used) {  // batches of an earlier call that ended without a synchronisation in between (finalize)
        HIPCHK(hipStreamSynchronize(st));
        h->kv_ops_used = 0;
    }
    for (int i = 0; i < n; ++i) {
        bs[i].seq.assign(B, std::vector<int>(prompt_ids[i], prompt_ids[i] + prompt_lens[i]));
        bs[i].score.assign(B, -1e9f);
        bs[i].score[0] = 0.f;
        bs[i].hyps.num_beams = B;
        bs[i].hyps.length_penalty = lp;
        for (int b = B; b < h->nbuf; ++b) bs[i].free_bufs.push_back(b);  // slots 0..B-1 are reorder temporaries
        // the prompt's KV was written to arena 0: replicate it into the other beams' arenas (they are identical before it)
        push_copy(h, ops, stream_ids[i], 0, 0, total0[i], rows_len[i], false);
    }
    CHK(flush_copies(h, ops, mh, md, st));
    for (int i = 0; i < n; ++i)
        for (int b = 1; b < B; ++b) push_copy(h, ops, stream_ids[i], b, 0, total0[i], rows_len[i], true);
    CHK(flush_copies(h, ops, mh, md, st));

    int step = 0;  // tokens already chosen per beam
    while (true) {
        const int rows_per = step == 0 ? 1 : B;  // step 0: only beam 0 carries a finite score (:767-771)
        const int rows = n * rows_per;
        // ---- log_softmax -> processors (on log-probs) -> per-row top-k ----
        for (int i = 0; i < n; ++i)
            for (int b = 0; b < rows_per; ++b) {
                const int r = i * rows_per + b;
                const std::vector<int>& sq = bs[i].seq[b];
                std::memcpy(mh.ids_pool + (size_t)r * h->max_ids, sq.data(), sq.size() * 4);
                const int ne = n_prev ? n_prev[i] : 0;
                if (b == 0 && ne) std::memcpy(mh.enc_pool + (size_t)i * h->max_enc_ids, prev_target_ids[i], (size_t)ne * 4);
                mh.samp[r].n_ids = (int)sq.size(); mh.samp[r].n_enc = ne;
                mh.samp[r].ids_off = r * h->max_ids; mh.samp[r].enc_off = i * h->max_enc_ids; mh.samp[r].logits_row = r;
            }
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
        CHK(launch_log_softmax(h->logits, h->vocab_pad, V, h->lse_max, h->lse_sum, rows, st));
        CHK(launch_sample_process(h->logits, h->vocab_pad, md.samp, md.ids_pool, md.enc_pool, md.suppress, p->n_suppress, p->repetition_penalty,
                                  p->no_repeat_ngram_size, p->encoder_no_repeat_ngram_size, rows, st));
        if (p->do_sample) {
            // beam SAMPLE (:871-875): the processed log-probs of every row come to the host; there the warpers (part of the processor list under do_sample),
            // + beam score, softmax over a stream's rows_per x V scores, n_keep draws without replacement (warp.hip) -- written into the same candidate
            // arrays the top-k fills below (value = warped log-prob of the drawn token, so that value + beam score is the reference's gathered score)
            HIPCHK(hipMemcpyAsync(h->samp_host, h->logits, (size_t)rows * h->vocab_pad * sizeof(float), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            h->kv_ops_used = 0;
            std::vector<float> flat((size_t)rows_per * V);
            std::vector<double> us(n_keep);
            std::vector<long> picked(n_keep);
            for (int i = 0; i < n; ++i) {
                for (int b = 0; b < rows_per; ++b) {
                    float* row = h->samp_host + (size_t)(i * rows_per + b) * h->vocab_pad;
                    warp_scores(row, V, p->temperature, p->top_k, p->top_p, p->epsilon_cutoff, c.n_eos + 1);  // min_tokens_to_keep of beam methods
                    const float bsc = bs[i].score[b];
                    for (int v2 = 0; v2 < V; ++v2) flat[(size_t)b * V + v2] = row[v2] + bsc;
                }
                for (int j = 0; j < n_keep; ++j) us[j] = sample_uniform(p->seed, stream_ids[i], h->streams[stream_ids[i]].chunks, 64 * step + j);
                if (multinomial_without_replacement(flat.data(), (long)rows_per * V, n_keep, us.data(), picked.data()) != ISST_OK)
                    return h->fail(ISST_ERR_STATE, "beam sample: fewer than %d tokens with non-zero probability at step %d (torch.multinomial raises here too)", n_keep, step);
                // candidate j of the stream goes to the slot (row of its beam, next free column); unused slots are marked invalid
                std::vector<int> used(rows_per, 0);
                for (int b = 0; b < rows_per; ++b)
                    for (int j = 0; j < BEAM_TOPK; ++j) h->top_idx_host[(i * rows_per + b) * BEAM_TOPK + j] = -1;
                for (int j = 0; j < n_keep; ++j) {
                    const int b = (int)(picked[j] / V), tok = (int)(picked[j] % V);
                    const int r = i * rows_per + b, slot = used[b]++;
                    h->top_val_host[r * BEAM_TOPK + slot] = h->samp_host[(size_t)r * h->vocab_pad + tok];
                    h->top_idx_host[r * BEAM_TOPK + slot] = tok;
                }
            }
        } else {
            // (the final selection stores its <= 32 candidates per row straight into the pinned host arrays: no device copy, no two D2H launches per step)
            CHK(launch_topk_rows(h->logits, h->vocab_pad, V, n_keep, h->cand_val, h->cand_idx, h->top_val_host, h->top_idx_host, rows, st));
            HIPCHK(hipStreamSynchronize(st));
            h->kv_ops_used = 0;  // every earlier copy batch has run
        }

        // ---- scorer (beam_search_process, :43-157) ----
        bool all_done = true;
        std::vector<std::vector<int>> parents(n), next_tok(n);
        for (int i = 0; i < n; ++i) {
            BeamStream& S = bs[i];
            const int prompt_len = prompt_lens[i];
            const int P0 = total0[i] + rows_len[i];  // first position written by the decode phase
            std::vector<int>& ntok = next_tok[i];
            std::vector<int>& npar = parents[i];
            if (S.done) {
                // a finished batch entry is skipped by the scorer (patch_hf.py:83-92: pad tokens, zero scores, its hypotheses untouched)
                // while the other streams of the call go on; its rows still ride through the forward pass (their KV beyond the
                // winner's tail is never read: llm_total is set from the winning hypothesis)
                const int pad = c.n_eos ? c.eos_ids[0] : 0;
                for (int b = 0; b < B; ++b) { ntok.push_back(pad); npar.push_back(b); S.seq[b].push_back(pad); }
                S.score.assign(B, 0.f);
                continue;
            }
            struct Cand { float val; long flat; };
            std::vector<Cand> cands;
            for (int b = 0; b < rows_per; ++b)
                for (int j = 0; j < n_keep; ++j) {
                    const int r = i * rows_per + b;
                    const int idx = h->top_idx_host[r * BEAM_TOPK + j];
                    if (idx < 0 || idx >= V) continue;
                    cands.push_back({h->top_val_host[r * BEAM_TOPK + j] + S.score[b], (long)b * V + idx});
                }
            if (h->btrace_on && i == 0) {
                isst_handle::BeamTraceStep ts;
                ts.rows = rows_per; ts.n_keep = n_keep;
                for (int b = 0; b < rows_per; ++b) {
                    for (int j = 0; j < n_keep; ++j) {
                        ts.val.push_back(h->top_val_host[b * BEAM_TOPK + j]);
                        ts.idx.push_back(h->top_idx_host[b * BEAM_TOPK + j]);
                    }
                    ts.score.push_back(S.score[b]);
                }
                h->btrace.push_back(std::move(ts));
            }
            std::stable_sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b2) { return a.val > b2.val || (a.val == b2.val && a.flat < b2.flat); });
            if ((int)cands.size() > n_keep) cands.resize(n_keep);
            const int cur_len = (int)S.seq[0].size() + 1;
            std::vector<float> nscore;
            for (size_t rank = 0; rank < cands.size(); ++rank) {
                const int b = (int)(cands[rank].flat / V), tok = (int)(cands[rank].flat % V);
                bool is_eos = false;
                for (int e = 0; e < c.n_eos; ++e) is_eos = is_eos || tok == c.eos_ids[e];
                if (is_eos) {
                    if ((int)rank >= B) continue;
                    BeamHyp hyp;
                    hyp.tokens = S.seq[b];
                    hyp.fed = step;
                    hyp.buf = -1;
                    if (step > 0) {  // keep a copy of that beam's tail (the reference clones the whole KV cache, :113-120)
                        if (S.free_bufs.empty()) return h->fail(ISST_ERR_STATE, "beam search ran out of hypothesis buffers");
                        hyp.buf = S.free_bufs.back();
                        S.free_bufs.pop_back();
                        push_copy(h, ops, stream_ids[i], b, hyp.buf, P0, step, false);
                    } else {
                        hyp.buf = -1 - 0;  // empty tail: nothing to keep
                    }
                    const int freed = S.hyps.add(std::move(hyp), (double)cands[rank].val, cur_len - prompt_len);
                    if (freed >= B) S.free_bufs.push_back(freed);
                } else {
                    nscore.push_back(cands[rank].val);
                    ntok.push_back(tok);
                    npar.push_back(b);
                }
                if ((int)ntok.size() == B) break;
            }
            if ((int)ntok.size() < B) return h->fail(ISST_ERR_STATE, "beam search: fewer than %d non-EOS candidates", B);
            if (!cands.empty()) S.done = S.done || S.hyps.is_done((double)cands[0].val, cur_len, prompt_len);
            if (h->btrace_on && i == 0 && (size_t)(step + 1) * B <= h->bforce_tok.size()) {
                // teacher forcing (test aid): continue with the caller's (token, parent) choices; a beam's score is its parent's score plus
                // the processed log-prob of the forced token, read back from the device's score row
                for (int b = 0; b < B; ++b) {
                    const int tok = h->bforce_tok[(size_t)step * B + b], par = h->bforce_par[(size_t)step * B + b];
                    if (tok < 0 || tok >= V || par < 0 || par >= rows_per) return h->fail(ISST_ERR_ARG, "forced beam choice (%d, %d) out of range at step %d", tok, par, step);
                    float lp = 0.f;
                    if (p->do_sample) lp = h->samp_host[(size_t)(i * rows_per + par) * h->vocab_pad + tok];  // (the warped row is on the host)
                    else HIPCHK(hipMemcpy(&lp, h->logits + (size_t)(i * rows_per + par) * h->vocab_pad + tok, sizeof(float), hipMemcpyDeviceToHost));
                    ntok[b] = tok; npar[b] = par; nscore[b] = S.score[par] + lp;
                }
            }
            // input_ids = cat(input_ids[beam_idx], tokens)  (:899)
            std::vector<std::vector<int>> nseq(B);
            for (int b = 0; b < B; ++b) { nseq[b] = S.seq[npar[b]]; nseq[b].push_back(ntok[b]); }
            S.seq.swap(nseq);
            S.score = nscore;
            all_done = all_done && S.done;
        }
        CHK(flush_copies(h, ops, mh, md, st));  // hypothesis tails first: the reorder below overwrites arenas
        ++step;
        // ---- reorder the tails (:910-913): new beam b continues parent npar[b].  Like the reference this happens BEFORE
        //      the stop test, so that finalize sees arena b == beam b ----
        if (step - 1 > 0) {
            for (int i = 0; i < n; ++i) {
                const int P0 = total0[i] + rows_len[i];
                std::set<int> needed;
                for (int b = 0; b < B; ++b) if (parents[i][b] != b) needed.insert(parents[i][b]);
                for (int src : needed) push_copy(h, ops, stream_ids[i], src, src, P0, step - 1, false);
            }
            CHK(flush_copies(h, ops, mh, md, st));
            for (int i = 0; i < n; ++i) {
                const int P0 = total0[i] + rows_len[i];
                for (int b = 0; b < B; ++b) if (parents[i][b] != b) push_copy(h, ops, stream_ids[i], b, parents[i][b], P0, step - 1, true);
            }
            CHK(flush_copies(h, ops, mh, md, st));
        }
        if (all_done || step >= p->max_new_tokens) break;  // :920
        // ---- next forward pass: one row per (stream, beam).  Shared-prefix form (llm_attn.hip, LlmStreamView::n_beams): the B rows of a
        //      stream are ONE attention group over arena 0 for everything older than this chunk's generated tokens (identical in all
        //      arenas) plus one workgroup per beam for the <= 4 tiles that differ; otherwise every beam is its own group over its arena ----
        const int nr = n * B;
        const bool shared = h->beam_shared && B * (c.llm_heads / c.llm_kv_heads) <= 16 && (p->max_new_tokens + 15) / 16 + 1 <= 4;  // (isst_generate's `beams_share_prefix`: the pre-pass relies on it)
        for (int i = 0; i < n; ++i) {
            const StreamState& ss = h->streams[stream_ids[i]];
            for (int b = 0; b < B; ++b) {
                const int r = i * B + b;
                mh.row_stream[r] = shared ? i * B : r;  // view index
                mh.row_pos[r] = total0[i] + rows_len[i] + step - 1;
                mh.ids[r] = bs[i].seq[b].back();
                mh.last_rows[r] = r;
                mh.views[r].sys_len = ss.llm_sys;
                mh.views[r].ring_start = ss.llm_ring_start;
                mh.views[r].kv_offset = h->arena_off(stream_ids[i], b);
                mh.views[r].new_start = mh.row_pos[r];
                mh.views[r].row0 = shared ? i * B : r;
                mh.views[r].rot_keys = h->rot_keys ? 1 : 0;  // every beam's arena carries its rotated keys (pre-pass over all arenas + position copies)
                mh.views[r].n_beams = shared ? B : 0;
                mh.views[r].tail_start = total0[i] + rows_len[i];
                mh.views[r].beam_stride = h->llm_stream_stride;
                if (shared) {
                    mh.groups[i].x = i * B;
                    mh.groups[i].y = B;
                } else {
                    mh.groups[r].x = r;
                    mh.groups[r].y = 1;
                }
            }
        }
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
        CHK(llm_forward(h, md, nr, nr, shared ? n : nr, shared ? B : 1, false, nullptr, st, &mh, 0, 0, shared ? B : 0));
    }

    // ---- finalize (:159-275): open beams become hypotheses, the best one wins ----
    for (int i = 0; i < n; ++i) {
        BeamStream& S = bs[i];
        const int prompt_len = prompt_lens[i];
        const int P0 = total0[i] + rows_len[i];
        if (!S.done)
            for (int b = 0; b < B; ++b) {
                BeamHyp hyp;
                hyp.tokens = S.seq[b];
                hyp.fed = (int)S.seq[b].size() - prompt_len - 1;  // the last chosen token was never fed
                hyp.buf = -1 - b;
                const int freed = S.hyps.add(std::move(hyp), (double)S.score[b], (int)S.seq[b].size() - prompt_len);
                if (freed >= B) S.free_bufs.push_back(freed);
            }
        if (S.hyps.beams.empty()) return h->fail(ISST_ERR_STATE, "beam search produced no hypothesis");
        size_t best = 0;
        for (size_t q = 1; q < S.hyps.beams.size(); ++q)
            if (S.hyps.beams[q].score >= S.hyps.beams[best].score) best = q;  // sorted(...).pop(): the last of equal scores
        const BeamHyp& win = S.hyps.beams[best];
        // make every arena of the stream hold the winner's tail
        if (win.fed > 0) {
            int src_buf = win.buf;
            int skip_beam = -1;
            if (win.buf < 0) {  // still in an arena: stage it through temporary 0
                skip_beam = -1 - win.buf;
                push_copy(h, ops, stream_ids[i], skip_beam, 0, P0, win.fed, false);
                CHK(flush_copies(h, ops, mh, md, st));
                src_buf = 0;
            }
            for (int b = 0; b < B; ++b)
                if (b != skip_beam) push_copy(h, ops, stream_ids[i], b, src_buf, P0, win.fed, true);
            CHK(flush_copies(h, ops, mh, md, st));
        }
        StreamState& ss = h->streams[stream_ids[i]];
        ss.llm_total = P0 + win.fed;
        ss.chunks++;
        const int max_length = prompt_len + p->max_new_tokens;
        std::vector<int> outv(win.tokens.begin() + prompt_len, win.tokens.end());
        if ((int)win.tokens.size() < std::min((int)win.tokens.size() + 1, max_length)) outv.push_back(c.n_eos ? c.eos_ids[0] : 0);
        for (size_t q = 0; q < outv.size(); ++q) out_ids[i][q] = outv[q];
        out_lens[i] = (int)outv.size();
    }
    HIPCHK(hipStreamSynchronize(st));
    return ISST_OK;
}

}  // namespace

extern "C" int isst_encode_speech(isst_handle* h, int stream_id, const float* pcm, int n_samples, int multiplier, uint16_t* out_features,
                                  int* out_rows, void* hip_stream) {
    if (!h) return ISST_ERR_ARG;
    CHK(check_ready(h));
    if (stream_id < 0 || stream_id >= (int)h->streams.size() || !h->streams[stream_id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", stream_id);
    if (!pcm || n_samples <= 0 || n_samples % h->chunk_samples || n_samples > h->n_new_max || multiplier < 1 || multiplier > h->cfg.max_multiplier)
        return h->fail(ISST_ERR_ARG, "n_samples must be a positive multiple of %d and <= %d", h->chunk_samples, h->n_new_max);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    int S = 0;
    const float* pp[1] = {pcm};
    CHK(run_encoder(h, 1, &stream_id, pp, false, n_samples, multiplier, st, &S));
    h->streams[stream_id].chunks++;
    if (out_features) HIPCHK(hipMemcpyAsync(out_features, h->speech, (size_t)S * h->cfg.llm_dim * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (out_rows) *out_rows = S;
    return ISST_OK;
}

// ISST_HOST_TRACE=1: one line per isst_generate on stderr with the host-side timeline of the call (us): gap since the previous call returned, entry -> encoder
// and prefill enqueued, the waits for each token, the host work between a token's arrival and the next pass being enqueued
struct HostTrace {
    bool on = false;
    std::chrono::steady_clock::time_point t_entry, t_prev_return;
    bool have_prev = false;
    double enq_first = 0, wait = 0, between = 0, enq_pass = 0;
    int waits = 0;
    static double us(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
};
static thread_local HostTrace g_ht;  // (per thread: isst_generate may run on several handles from several threads)
extern "C" int isst_generate(isst_handle* h, const isst_gen_params* p, int n, const int* stream_ids, const float* const* pcm, int n_samples,
                             const int* const* prompt_ids, const int* prompt_lens, const int* const* prev_target_ids, const int* n_prev,
                             const int* const* forced_tokens, const int* n_forced, int* const* out_ids, int* out_lens, float* logits_out,
                             void* hip_stream) {
    if (!h) return ISST_ERR_ARG;
    static const bool ht_env = std::getenv("ISST_HOST_TRACE") && std::atoi(std::getenv("ISST_HOST_TRACE")) > 0;
    g_ht.on = ht_env;
    if (g_ht.on) { g_ht.t_entry = std::chrono::steady_clock::now(); g_ht.enq_first = g_ht.wait = g_ht.between = g_ht.enq_pass = 0; g_ht.waits = 0; }
    CHK(check_ready(h));
    const isst_config& c = h->cfg;
    if (!p || !stream_ids || !pcm || !prompt_ids || !prompt_lens || !out_ids || !out_lens) return h->fail(ISST_ERR_ARG, "null argument");
    if (n < 1 || n > c.max_streams) return h->fail(ISST_ERR_ARG, "n = %d streams, capacity %d", n, c.max_streams);
    if (n_samples <= 0 || n_samples % h->chunk_samples || n_samples > h->n_new_max)
        return h->fail(ISST_ERR_ARG, "n_samples %d must be a positive multiple of %d and <= %d", n_samples, h->chunk_samples, h->n_new_max);
    if (p->multiplier < 1 || p->multiplier > c.max_multiplier || p->max_new_tokens < 1 || p->max_new_tokens > c.max_new_tokens)
        return h->fail(ISST_ERR_ARG, "multiplier / max_new_tokens out of configured range");
    if (p->n_suppress < 0 || p->n_suppress > 65536 || (p->n_suppress && !p->suppress_tokens)) return h->fail(ISST_ERR_ARG, "suppress_tokens");
    if (p->no_repeat_ngram_size < 0 || p->encoder_no_repeat_ngram_size < 0 || p->no_repeat_ngram_size > 64 || p->encoder_no_repeat_ngram_size > 64)
        return h->fail(ISST_ERR_ARG, "ngram sizes");
    const int B = p->num_beams > 1 ? p->num_beams : 1;
    if (B > h->max_beams) return h->fail(ISST_ERR_ARG, "num_beams %d exceeds the configured max_beams %d", B, h->max_beams);
    bool any_forced = false;
    if (forced_tokens && n_forced)
        for (int i = 0; i < n; ++i) any_forced = any_forced || (forced_tokens[i] != nullptr && n_forced[i] > 0);
    if (B > 1 && (any_forced || logits_out)) return h->fail(ISST_ERR_ARG, "forced_tokens / logits_out are greedy-only test aids");
    if (p->do_sample && (p->top_k < 0 || !(p->top_p > 0.f) || p->epsilon_cutoff < 0.f || p->epsilon_cutoff >= 1.f))
        return h->fail(ISST_ERR_ARG, "sampling arguments out of range (top_k >= 0, top_p > 0, 0 <= epsilon_cutoff < 1)");
    if (p->do_sample && h->samp_host_rows < (size_t)n * B) {  // (beam sample: the processed scores of every beam's row come to the host)
        if (h->samp_host) (void)hipHostFree(h->samp_host);
        h->samp_host = nullptr;
        h->samp_host_rows = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&h->samp_host), (size_t)n * B * h->vocab_pad * sizeof(float)) != hipSuccess) return h->fail(ISST_ERR_NOMEM, "pinned score buffer");
        h->samp_host_rows = (size_t)n * B;
    }
    for (int i = 0; i < n; ++i) {
        const int id = stream_ids[i];
        if (id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", id);
        for (int j = 0; j < i; ++j) if (stream_ids[j] == id) return h->fail(ISST_ERR_ARG, "stream %d listed twice", id);
        if (prompt_lens[i] < 1 || prompt_lens[i] > c.max_prompt_len) return h->fail(ISST_ERR_ARG, "prompt length %d (max %d)", prompt_lens[i], c.max_prompt_len);
        if (n_prev && (n_prev[i] < 0 || n_prev[i] > h->max_enc_ids)) return h->fail(ISST_ERR_ARG, "too many previous target ids");
        const StreamState& s = h->streams[id];
        if (s.beams != 0 && s.beams != B && s.llm_total > 0)
            return h->fail(ISST_ERR_STATE, "stream %d was started with num_beams %d; reset it before switching to %d", id, s.beams, B);
        const int total = s.llm_total;
        int sys = s.llm_sys;
        if (total == 0 && p->system_prompt_size > 0) {
            if (p->system_prompt_size > h->sys_cap || p->system_prompt_size > prompt_lens[i]) return h->fail(ISST_ERR_ARG, "system_prompt_size %d (capacity %d, prompt %d)", p->system_prompt_size, h->sys_cap, prompt_lens[i]);
            sys = p->system_prompt_size;
        }
        if (total + prompt_lens[i] + p->max_new_tokens - sys > h->ring_cap)
            return h->fail(ISST_ERR_STATE, "stream %d: LLM cache of %d entries + this chunk exceeds the ring (%d); evict first", id, total, h->ring_cap);
        for (int t = 0; t < prompt_lens[i]; ++t)
            if (prompt_ids[i][t] < 0 || prompt_ids[i][t] >= c.vocab) return h->fail(ISST_ERR_ARG, "prompt token %d out of range", prompt_ids[i][t]);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);

    // ---- 1. speech encoder (model/llm.py:69-81) ----
    int S = 0;
    CHK(run_encoder(h, n, stream_ids, pcm, p->pcm_on_device != 0, n_samples, p->multiplier, st, &S));

    // ---- 2. prefill rows, speech splice map (model/llm.py:86-113) ----
    StepMeta mh = carve(h, h->meta_host), md = carve(h, h->meta_dev);
    std::vector<int> total0(n), gen_count(n, 0), row0(n), rows_len(n);  // rows_len: decoder rows (= KV entries) of the prompt after the splice
    std::vector<char> done(n, 0);
    int R = 0, n_groups = 0, n_units = 0, max_unit_groups = 0;
    const int gmax = LLM_ATTN_GROUP_ROWS(c.llm_heads / c.llm_kv_heads);
    for (int i = 0; i < n; ++i) {
        StreamState& s = h->streams[stream_ids[i]];
        if (s.llm_total == 0) s.llm_sys = p->system_prompt_size > 0 ? p->system_prompt_size : 0;
        s.beams = B;
        total0[i] = s.llm_total;
        mh.views[i].sys_len = s.llm_sys;
        mh.views[i].ring_start = s.llm_ring_start;
        mh.views[i].kv_offset = h->arena_off(stream_ids[i], 0);
        mh.views[i].new_start = total0[i];
        mh.views[i].row0 = R;
        mh.views[i].rot_keys = h->rot_keys ? 1 : 0;
        mh.views[i].n_beams = 0;
        row0[i] = R;
        const int len = prompt_lens[i];
        const int* ids = prompt_ids[i];
        std::vector<int> desc;
        if (splice_rows(ids, len, c.user_id, c.assistant_id, c.start_header_id, S, desc) != ISST_OK)
            return h->fail(ISST_ERR_ARG, "stream %d: malformed prompt (an assistant header before the end of its user turn)", stream_ids[i]);
        const int elen = (int)desc.size();
        if (elen < 1) return h->fail(ISST_ERR_ARG, "stream %d: empty prompt after the speech splice", stream_ids[i]);
        rows_len[i] = elen;
        for (int t = 0; t < elen; ++t) {
            mh.row_stream[R + t] = i;
            mh.row_pos[R + t] = total0[i] + t;
            mh.ids[R + t] = desc[t] >= 0 ? ids[desc[t]] : 0;
            mh.speech_row[R + t] = desc[t] >= 0 ? -1 : i * S + (-1 - desc[t]);
        }
        mh.last_rows[i] = R + elen - 1;
        const int g_first = n_groups;
        for (int t = 0; t < elen; t += gmax) {  // attention row groups: consecutive rows of one stream
            mh.groups[n_groups].x = R + t;
            mh.groups[n_groups].y = std::min(gmax, elen - t);
            ++n_groups;
        }
        for (int g0 = g_first; g0 < n_groups; g0 += 8) {  // units: runs of <= 8 groups of this stream share their key tiles
            mh.units[n_units].x = g0;
            mh.units[n_units].y = std::min(8, n_groups - g0);
            max_unit_groups = std::max(max_unit_groups, mh.units[n_units].y);
            ++n_units;
        }
        R += elen;
        // sampling context (the processors see the prompt's ids, patch tokens included: input_ids is not shortened)
        std::memcpy(mh.ids_pool + (size_t)i * h->max_ids, ids, (size_t)len * 4);
        const int ne = n_prev ? n_prev[i] : 0;
        if (ne) std::memcpy(mh.enc_pool + (size_t)i * h->max_enc_ids, prev_target_ids[i], (size_t)ne * 4);
        mh.samp[i].n_ids = len; mh.samp[i].n_enc = ne;
        mh.samp[i].ids_off = i * h->max_ids; mh.samp[i].enc_off = i * h->max_enc_ids; mh.samp[i].logits_row = i;
    }
    if (p->n_suppress) {
        std::memcpy(mh.suppress, p->suppress_tokens, (size_t)p->n_suppress * 4);
        HIPCHK(hipMemcpyAsync(h->meta_dev + mh.suppress_offset, h->meta_host + mh.suppress_offset, (size_t)p->n_suppress * 4, hipMemcpyHostToDevice, st));
    }
    // Beam search: in the shared-prefix form (beam_decode: the B beams of a stream are ONE attention group that reads everything older than this chunk's
    // generated tokens from arena 0) the rotated copies of arenas 1 .. B-1 are never read below tail_start -- their tails are written by the appends and the
    // position copies of this chunk -- so the pre-pass rotates arena 0 only: 64 streams x 4 beams 7.9 -> 2.0 ms per chunk (profiles/r04/trace_busy_prof64x4_*).
    const bool beams_share_prefix = B > 1 && h->beam_shared && B * (c.llm_heads / c.llm_kv_heads) <= 16 && (p->max_new_tokens + 15) / 16 + 1 <= 4;  // = beam_decode's `shared`
    const int rope_views = n * ((B > 1 && !beams_share_prefix) ? B : 1);
    if (B > 1 && h->rot_keys && !beams_share_prefix) {  // the rotated-key pre-pass below also covers arenas 1 .. B-1 (the prefill itself only reads views 0 .. n-1)
        int nv = n;
        for (int i = 0; i < n; ++i)
            for (int b = 1; b < B; ++b) {
                mh.views[nv] = mh.views[i];
                mh.views[nv].kv_offset = h->arena_off(stream_ids[i], b);
                ++nv;
            }
    }
    bool any_cached = false;
    for (int i = 0; i < n; ++i) any_cached = any_cached || total0[i] > 0;
    if (h->rot_keys && any_cached && h->rope_side && st != nullptr) {
        // The host is far ahead of the GPU here (the encoder above is milliseconds of queued work), so the metadata upload and the pre-pass, issued on
        // the side stream now, run BESIDE the encoder; the prefill waits for both.  (The caller's stream was idle when this call began -- every call ends
        // with a synchronisation -- so the pre-pass cannot overtake an earlier writer of the caches.)
        if (!h->side) {
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
            HIPCHK(hipEventCreateWithFlags(&h->side_ev, hipEventDisableTiming));
        }
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, h->side));
        CHK(launch_llm_rope_cache(md.views, rope_views, h->llm_cos, h->llm_sin, h->llm_k, h->llm_kr, h->adims, c.llm_layers, h->side));
        HIPCHK(hipEventRecord(h->side_ev, h->side));
        HIPCHK(hipStreamWaitEvent(st, h->side_ev, 0));
    } else {
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
        // rotate the cached keys of every layer ONCE for this chunk (positions are fixed until the next eviction)
        // (beam search: views n .. n*B-1, written above, are the other beams' arenas of the same streams -- they hold the same cached keys)
        if (h->rot_keys && any_cached) CHK(launch_llm_rope_cache(md.views, rope_views, h->llm_cos, h->llm_sin, h->llm_k, h->llm_kr, h->adims, c.llm_layers, st));
    }
    CHK(llm_forward(h, md, R, n, n_groups, gmax, true, "llm_", st, &mh, n_units, max_unit_groups));
    if (B > 1)
        return beam_decode(h, p, n, stream_ids, prompt_ids, prompt_lens, rows_len, prev_target_ids, n_prev, total0, out_ids, out_lens, mh, md, st);

    // ---- 3. greedy loop (patch_hf.py:606-624 -> HF _sample) ----
    std::vector<int> active(n);
    for (int i = 0; i < n; ++i) active[i] = i;

    // sampling tail of a pass over `na` rows: (test aid: logits download) -> processors + argmax -> token ids to the host
    bool tail_fused = false;  // the tail enqueued last went the fused way (wait_tokens then waits on pinned memory)
    auto sample_tail = [&](int na) -> int {
        if (logits_out)
            for (int r = 0; r < na; ++r)
                HIPCHK(hipMemcpyAsync(logits_out + ((size_t)active[r] * p->max_new_tokens + gen_count[active[r]]) * c.vocab,
                                      h->logits + (size_t)r * h->vocab_pad, (size_t)c.vocab * sizeof(float), hipMemcpyDeviceToHost, st));
        if (p->do_sample) {  // processors on the device, the processed rows to the host: warpers + draw happen there (warp.hip) after the synchronisation
            CHK(launch_sample_process(h->logits, h->vocab_pad, md.samp, md.ids_pool, md.enc_pool, md.suppress, p->n_suppress, p->repetition_penalty,
                                      p->no_repeat_ngram_size, p->encoder_no_repeat_ngram_size, na, st));
            HIPCHK(hipMemcpyAsync(h->samp_host, h->logits, (size_t)na * h->vocab_pad * sizeof(float), hipMemcpyDeviceToHost, st));
            return ISST_OK;
        }
        // (up to 16 rows.  With many streams the tail is 4096 blocks whose per-block publish fences -- an L2 write-back each -- cost more than the two launches they
        //  save: 64 streams 84.0-84.5 ms per step fused against 83.4-83.5 with the three-launch tail, while one stream gains 0.18 ms per chunk)
        tail_fused = h->fused_sample && na <= 16;
        if (tail_fused) {  // one launch: processors, argmax, the tokens straight into pinned host memory, then the tail's sequence number behind them
            CHK(launch_sample_fused(h->logits, h->vocab_pad, c.vocab, md.samp, md.ids_pool, md.enc_pool, md.suppress, p->n_suppress, p->repetition_penalty,
                                    p->no_repeat_ngram_size, p->encoder_no_repeat_ngram_size, h->out_tok, h->samp_val, h->samp_idx, h->samp_tickets, h->tok_host,
                                    h->tok_host + h->tok_cap, na, st));
            ++h->samp_seq_expected;
            return ISST_OK;
        }
        // (three launches; the last one stores the tokens straight into the pinned host array -- no D2H launch behind it)
        CHK(launch_sample(h->logits, h->vocab_pad, c.vocab, md.samp, md.ids_pool, md.enc_pool, md.suppress, p->n_suppress, p->repetition_penalty,
                          p->no_repeat_ngram_size, p->encoder_no_repeat_ngram_size, h->tok_host, h->samp_val, h->samp_idx, na, st));
        return ISST_OK;
    };
    // the tokens of the tail that was enqueued last are on the host.  Fused tail: wait for its sequence number in pinned memory (the kernel stores it, system scope,
    // after the tokens) -- the host sees the tokens a completion-signal round trip earlier than through hipStreamSynchronize; a stream that has drained WITHOUT
    // publishing (a failed launch) ends the wait with an error instead of a hang.  Test aids with pending D2H copies and the sample branch synchronise as before.
    auto wait_tokens = [&]() -> int {
        if (!tail_fused || p->do_sample || logits_out) {
            HIPCHK(hipStreamSynchronize(st));
            if (tail_fused && !p->do_sample && *reinterpret_cast<volatile int*>(h->tok_host + h->tok_cap) != h->samp_seq_expected) {
                const int want = h->samp_seq_expected;
                h->samp_seq_expected = *reinterpret_cast<volatile int*>(h->tok_host + h->tok_cap);  // the stream has drained: adopt the device's count, later calls start in step
                return h->fail(ISST_ERR_HIP, "sampling tail %d did not publish its tokens", want);
            }
            return ISST_OK;
        }
        volatile int* seq = h->tok_host + h->tok_cap;
        for (unsigned long spins = 1;; ++spins) {
            if (*seq == h->samp_seq_expected) break;
            if ((spins & 0x3FFFF) == 0) {  // every ~quarter million polls: has the stream drained without the number arriving?
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    if (*seq == h->samp_seq_expected) break;
                    const int want = h->samp_seq_expected;
                    h->samp_seq_expected = *seq;  // drained without publishing: adopt the device's count, later calls start in step
                    return h->fail(ISST_ERR_HIP, "sampling tail %d did not publish its tokens", want);
                }
                if (q != hipErrorNotReady) return h->fail(ISST_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        return ISST_OK;
    };
    // one decode step over nr rows: metadata upload -> decoder stack -> sampling tail.  Every pointer and dimension in it is
    // the same from step to step (the per-step values live in the metadata block), so it is captured once per row count and
    // replayed: ~230 launches become one graph launch
    auto decode_step = [&](int nr) -> int {
        const bool graph_ok = h->use_graphs && st != nullptr && !logits_out && !c.debug_taps && !h->prof_on && !p->do_sample;  // (the NULL stream cannot be captured)
        if (!graph_ok) {
            HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
            CHK(llm_forward(h, md, nr, nr, nr, 1, false, nullptr, st, &mh));
            return sample_tail(nr);
        }
        isst_handle::DecodeGraph& g = h->dgraph;
        bool captured_now = false;
        const int seq_before = h->samp_seq_expected;  // the host's count of fused tails moves only with a launch that was really enqueued (every error path below restores it)
        if (!g.exec || g.rows != nr || g.n_suppress != p->n_suppress || g.ngram != p->no_repeat_ngram_size ||
            g.enc_ngram != p->encoder_no_repeat_ngram_size || g.penalty != p->repetition_penalty) {
            captured_now = true;  // (sample_tail below counts the fused tail once; the launch that follows is its first execution)
            if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
            hipGraph_t graph = nullptr;
            HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            int rc = ISST_OK;
            if (hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st) != hipSuccess) rc = ISST_ERR_HIP;
            if (rc == ISST_OK) rc = llm_forward(h, md, nr, nr, nr, 1, false, nullptr, st, nullptr);  // metadata from device memory: nothing frozen in the arguments
            if (rc == ISST_OK) rc = sample_tail(nr);
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (rc != ISST_OK || ce != hipSuccess || !graph) {
                if (graph) (void)hipGraphDestroy(graph);
                h->samp_seq_expected = seq_before;
                return rc != ISST_OK ? rc : h->fail(ISST_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(ce));
            }
            const hipError_t ie = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (ie != hipSuccess) { g.exec = nullptr; h->samp_seq_expected = seq_before; return h->fail(ISST_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ie)); }
            g.rows = nr; g.n_suppress = p->n_suppress; g.ngram = p->no_repeat_ngram_size; g.enc_ngram = p->encoder_no_repeat_ngram_size;
            g.penalty = p->repetition_penalty;
        }
        tail_fused = h->fused_sample && nr <= 16 && !p->do_sample;  // (what sample_tail chose when this row count was captured)
        const hipError_t le = hipGraphLaunch(g.exec, st);
        if (le != hipSuccess) { h->samp_seq_expected = seq_before; return h->fail(ISST_ERR_HIP, "hipGraphLaunch: %s", hipGetErrorString(le)); }
        if (!captured_now && tail_fused) ++h->samp_seq_expected;  // a replay ran the captured fused tail once more (counted only once it is enqueued)
        return ISST_OK;
    };

    if (const int rc = sample_tail(n)) return rc;
    std::chrono::steady_clock::time_point ht_a, ht_b;
    if (g_ht.on) { ht_a = std::chrono::steady_clock::now(); g_ht.enq_first = HostTrace::us(g_ht.t_entry, ht_a); }
    while (true) {
        const int na = (int)active.size();
        if (g_ht.on) ht_a = std::chrono::steady_clock::now();
        CHK(wait_tokens());
        if (g_ht.on) { ht_b = std::chrono::steady_clock::now(); g_ht.wait += HostTrace::us(ht_a, ht_b); g_ht.waits++; }
        if (p->do_sample)  // HF _sample with do_sample: warpers, softmax, one draw per row (patch_hf.py:606-624)
            for (int r = 0; r < na; ++r)
                h->tok_host[r] = warp_and_sample(h->samp_host + (size_t)r * h->vocab_pad, c.vocab, p->temperature, p->top_k, p->top_p, p->epsilon_cutoff,
                                                 sample_uniform(p->seed, stream_ids[active[r]], h->streams[stream_ids[active[r]]].chunks, gen_count[active[r]]));
        std::vector<int> next_active;
        for (int r = 0; r < na; ++r) {
            const int i = active[r];
            int tok = h->tok_host[r];
            if (forced_tokens && forced_tokens[i] && n_forced && gen_count[i] < n_forced[i]) tok = forced_tokens[i][gen_count[i]];
            if (tok < 0 || tok >= c.vocab) return h->fail(ISST_ERR_ARG, "token %d out of range", tok);
            out_ids[i][gen_count[i]++] = tok;
            bool stop = gen_count[i] >= p->max_new_tokens;
            for (int e = 0; e < c.n_eos; ++e) stop = stop || tok == c.eos_ids[e];
            if (forced_tokens && forced_tokens[i] && n_forced && gen_count[i] >= n_forced[i]) stop = true;
            if (!stop) next_active.push_back(i);
            else done[i] = 1;
        }
        active.swap(next_active);
        if (active.empty()) break;

        // next decode step: one row per active stream, the token just sampled at the next position
        const int nr = (int)active.size();
        for (int r = 0; r < nr; ++r) {
            const int i = active[r];
            const int tok = out_ids[i][gen_count[i] - 1];
            mh.row_stream[r] = i;
            mh.row_pos[r] = total0[i] + rows_len[i] + gen_count[i] - 1;
            mh.views[i].new_start = mh.row_pos[r];
            mh.views[i].row0 = r;
            mh.groups[r].x = r;
            mh.groups[r].y = 1;
            mh.ids[r] = tok;
            mh.last_rows[r] = r;
            mh.ids_pool[(size_t)i * h->max_ids + prompt_lens[i] + gen_count[i] - 1] = tok;
            mh.samp[r].n_ids = prompt_lens[i] + gen_count[i];
            mh.samp[r].n_enc = n_prev ? n_prev[i] : 0;
            mh.samp[r].ids_off = i * h->max_ids; mh.samp[r].enc_off = i * h->max_enc_ids; mh.samp[r].logits_row = r;
        }
        if (g_ht.on) { ht_a = std::chrono::steady_clock::now(); g_ht.between += HostTrace::us(ht_b, ht_a); }
        if (const int rc = decode_step(nr)) return rc;  // (the failing call has already recorded its message)
        if (g_ht.on) g_ht.enq_pass += HostTrace::us(ht_a, std::chrono::steady_clock::now());
    }
    // (every call still ENDS with the stream drained -- the side stream of the next call and the host-side eviction rely on it; with the fused tail the last
    //  wait above returned on the published tokens, a few microseconds before the kernel's own completion)
    if (tail_fused) HIPCHK(hipStreamSynchronize(st));
    // ---- 4. state: the cache holds the prompt and every generated token except the last one ----
    for (int i = 0; i < n; ++i) {
        StreamState& s = h->streams[stream_ids[i]];
        s.llm_total = total0[i] + rows_len[i] + gen_count[i] - 1;
        s.chunks++;
        out_lens[i] = gen_count[i];
    }
    if (g_ht.on) {
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[isst host] n=%d since_prev_return %.1f  entry->prefill_enqueued %.1f  waits %d total %.1f  token->next_enqueue_start avg %.1f  enqueue_pass avg %.1f  call %.1f us\n", n,
                     g_ht.have_prev ? HostTrace::us(g_ht.t_prev_return, g_ht.t_entry) : -1.0, g_ht.enq_first, g_ht.waits, g_ht.wait,
                     g_ht.waits > 1 ? g_ht.between / (g_ht.waits - 1) : 0.0, g_ht.waits > 1 ? g_ht.enq_pass / (g_ht.waits - 1) : 0.0, HostTrace::us(g_ht.t_entry, now));
        g_ht.t_prev_return = now;
        g_ht.have_prev = true;
    }
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// per-kernel entry points
// --------------------------------------------------------------------------------------------
extern "C" int64_t isst_op_packed_elems(int n_rows, int K) { return (int64_t)round_up(n_rows, 16) * K; }

extern "C" int isst_op_pack_weight(const uint16_t* w, uint16_t* packed, int n_rows, int K, int conv_k, void* hip_stream) {
    return launch_pack_weight(w, packed, n_rows, K, 0, 1, 0, conv_k, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_gemm(const uint16_t* A, int64_t lda, const uint16_t* packed, const uint16_t* bias, const uint16_t* res, int64_t ldres,
                            void* out, int64_t ldo, int M, int N, int K, int n_valid, int epi, const uint16_t* norm_w, float norm_eps,
                            void* hip_stream) {
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = packed; g.bias = bias; g.res = res; g.ldres = ldres; g.out = out; g.ldo = ldo;
    g.M = M; g.N = round_up(N, 16); g.K = K; g.batch = 1; g.epi = epi; g.n_valid = n_valid;
    g.norm_w = norm_w; g.norm_eps = norm_eps;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_gemm_splitk_rmsnorm(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* x, const uint16_t* norm_w, uint16_t* out,
                                           float* slabs, int M, int N, int K, int ksplit, float norm_eps, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    if (N % 16 != 0) return ISST_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = packed; g.out = slabs; g.ldo = N; g.out_batch = (long)M * N;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.epi = EPI_PARTIAL; g.n_valid = N; g.ksplit = ksplit;
    if (!gemm_mid_supported(g) && !gemm_tiled_supported(g)) return ISST_ERR_ARG;
    const int rc = launch_gemm(g, st);
    if (rc != ISST_OK) return rc;
    return launch_rmsnorm_reduce(slabs, (long)M * N, ksplit, x, N, norm_w, out, N, M, N, norm_eps, st);
}
extern "C" int isst_op_gemm_splitk_fused(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* x, float* slabs, float* ssq, int* tickets,
                                         int M, int N, int K, int ksplit, void* hip_stream) {
    if (N % 32 != 0 || !x || !slabs || !tickets) return ISST_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = packed; g.out = slabs; g.ldo = N; g.out_batch = (long)M * N;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.epi = EPI_PARTIAL; g.n_valid = N; g.ksplit = ksplit;
    g.res = x; g.ldres = N; g.ssq = ssq; g.ssq_n = N / 32; g.tickets = tickets;
    if (!gemm_mid_supported(g)) return ISST_ERR_ARG;
    return launch_gemm_mid(g, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_gemm_splitk_plain(const uint16_t* A, int64_t lda, const uint16_t* packed, uint16_t* out, int64_t ldo, float* slabs, int* tickets,
                                         int M, int N, int K, int ksplit, const uint16_t* norm_w, float norm_eps, float* ssq_in, void* hip_stream) {
    if (N % 32 != 0 || !out || !slabs || !tickets || (norm_w && !ssq_in)) return ISST_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = packed; g.out = slabs; g.ldo = N; g.out_batch = (long)M * N;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.epi = EPI_PARTIAL; g.n_valid = N; g.ksplit = ksplit;
    g.res = out; g.ldres = ldo; g.reduce_plain = 1; g.tickets = tickets;
    g.norm_w = norm_w; g.norm_eps = norm_eps; g.ssq = norm_w ? ssq_in : nullptr; g.ssq_n = norm_w ? K / 32 : 0;
    if (!gemm_mid_supported(g)) return ISST_ERR_ARG;
    return launch_gemm_mid(g, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_gemm_norm_ssq(const uint16_t* x, int64_t ldx, const uint16_t* packed, void* out, int64_t ldo, int M, int N, int K, int n_valid,
                                     int epi, const uint16_t* norm_w, float norm_eps, float* ssq, void* hip_stream) {
    if (!norm_w || !ssq) return ISST_ERR_ARG;
    GemmArgs g{};
    g.A = x; g.lda = ldx; g.Wp = packed; g.out = out; g.ldo = ldo;
    g.M = M; g.N = round_up(N, 16); g.K = K; g.batch = 1; g.epi = epi; g.n_valid = n_valid;
    g.norm_w = norm_w; g.norm_eps = norm_eps; g.ssq = ssq; g.ssq_n = K / 32;
    if (!gemm_mid_supported(g) && !gemm_wide_supported(g)) return ISST_ERR_ARG;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(hip_stream));  // (gemm_mid.hip, or gemm_wide.hip for the widest projections from 33 rows on)
}
extern "C" int isst_op_gemm_splitk_layernorm(const uint16_t* A, int64_t lda, const uint16_t* packed, const uint16_t* bias, uint16_t* x,
                                             const uint16_t* ln_w, const uint16_t* ln_b, uint16_t* out, float* slabs, int M, int N, int K, int ksplit,
                                             float eps, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    if (N % 16 != 0 || !bias) return ISST_ERR_ARG;
    GemmArgs g{};
    g.A = A; g.lda = lda; g.Wp = packed; g.out = slabs; g.ldo = N; g.out_batch = (long)M * N;
    g.M = M; g.N = N; g.K = K; g.batch = 1; g.epi = EPI_PARTIAL; g.n_valid = N; g.ksplit = ksplit;
    if (!gemm_mid_supported(g) && !gemm_tiled_supported(g)) return ISST_ERR_ARG;
    const int rc = launch_gemm(g, st);
    if (rc != ISST_OK) return rc;
    return launch_layernorm_reduce(slabs, (long)M * N, ksplit, bias, x, N, ln_w, ln_b, out, N, M, N, eps, st);
}
extern "C" int isst_op_set_gemm_tuning(int waves_per_block, int ntiles_per_block) {
    gemm_set_tuning(waves_per_block, ntiles_per_block);
    return ISST_OK;
}
extern "C" int isst_profile_begin_rows(isst_handle* h, int rows_lo, int rows_hi) {
    if (!h || rows_lo < 1 || rows_hi < rows_lo) return h ? h->fail(ISST_ERR_ARG, "isst_profile_begin_rows: bad row range %d..%d", rows_lo, rows_hi) : ISST_ERR_ARG;
    h->prof_on = true;
    h->prof_used = 0;
    h->prof_rows_lo = rows_lo;
    h->prof_rows_hi = rows_hi;
    return ISST_OK;
}
extern "C" int isst_profile_begin(isst_handle* h) { return isst_profile_begin_rows(h, 1, 1); }
extern "C" int isst_profile_end(isst_handle* h, void* hip_stream, double* avg_us, int64_t* launches) {
    if (!h || !avg_us || !launches) return ISST_ERR_ARG;
    h->prof_on = false;
    HIPCHK(hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream)));
    double sum = 0.0;
    for (size_t i = 0; i + 1 < h->prof_used; i += 2) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, h->prof_ev[i], h->prof_ev[i + 1]));
        sum += ms * 1e3;
    }
    *launches = (int64_t)(h->prof_used / 2);
    *avg_us = *launches ? sum / *launches : 0.0;
    return ISST_OK;
}
extern "C" int isst_op_set_attn_tuning(int target_workgroups) {
    llm_attn_set_tuning(target_workgroups);
    return ISST_OK;
}
extern "C" int isst_op_layernorm(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* out, int rows, int C, float eps, int gelu,
                                 void* hip_stream) {
    return launch_layernorm(x, C, w, b, out, C, rows, C, eps, gelu, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_rmsnorm(const uint16_t* x, const uint16_t* w, uint16_t* out, int rows, int D, float eps, void* hip_stream) {
    return launch_rmsnorm(x, D, nullptr, w, out, D, rows, D, eps, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_conv0(const uint16_t* audio, const uint16_t* w, const uint16_t* bias, const uint16_t* ln_w, const uint16_t* ln_b,
                             uint16_t* out, int T, int C, int k, int stride, void* hip_stream) {
    return launch_conv0(audio, 0, w, bias, ln_w, ln_b, out, 0, T, C, k, stride, 1, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_sample(float* logits, int vocab, const int* ids, int n_ids, const int* enc_ids, int n_enc, const int* suppress,
                              int n_suppress, float repetition_penalty, int ngram, int enc_ngram, int* out_token, void* hip_stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    SampleStream ss{n_ids, n_enc, 0, 0, 0};
    unsigned char* scratch = nullptr;  // [SampleStream | 64 floats | 64 ints]
    if (hipMalloc(reinterpret_cast<void**>(&scratch), 1024) != hipSuccess) return ISST_ERR_NOMEM;
    SampleStream* dss = reinterpret_cast<SampleStream*>(scratch);
    int rc = ISST_ERR_HIP;
    if (hipMemcpyAsync(dss, &ss, sizeof ss, hipMemcpyHostToDevice, st) == hipSuccess)
        rc = launch_sample(logits, vocab, vocab, dss, ids, enc_ids, suppress, n_suppress, repetition_penalty, ngram, enc_ngram, out_token,
                           reinterpret_cast<float*>(scratch + 256), reinterpret_cast<int*>(scratch + 512), 1, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(scratch);
    return rc;
}

extern "C" int isst_debug_read_kv(isst_handle* h, int id, int beam, int layer, int kv_head, int pos, uint16_t* k_out, uint16_t* v_out) {
    if (!h || !k_out || !v_out) return ISST_ERR_ARG;
    if (id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", id);
    const StreamState& s = h->streams[id];
    if (beam < 0 || beam >= h->max_beams || layer < 0 || layer >= h->cfg.llm_layers || kv_head < 0 || kv_head >= h->cfg.llm_kv_heads || pos < 0 ||
        pos >= s.llm_total)
        return h->fail(ISST_ERR_ARG, "isst_debug_read_kv: index out of range");
    const int slots = h->sys_cap + h->ring_cap;
    long slot = pos;
    if (pos >= s.llm_sys) slot = (long)h->sys_cap + (s.llm_ring_start + (pos - s.llm_sys)) % h->ring_cap;
    const long base = h->arena_off(id, beam) + (long)layer * h->adims.layer_stride + (long)kv_head * slots * 128;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(k_out, h->llm_k + base + slot * 128, 256, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(v_out, h->llm_v + base + slot * 128, 256, hipMemcpyDeviceToHost));
    return ISST_OK;
}

extern "C" int isst_debug_tap(isst_handle* h, const char* name, uint16_t* dst, int64_t max_elems, int64_t* got_elems) {
    if (!h || !name || !got_elems) return ISST_ERR_ARG;
    auto it = h->taps.find(name);
    if (it == h->taps.end() || !it->second.dev) return h->fail(ISST_ERR_NOTFOUND, "no tap named %s (cfg.debug_taps set?)", name);
    *got_elems = it->second.elems;
    if (dst) {
        const int64_t nel = it->second.elems < max_elems ? it->second.elems : max_elems;
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(dst, it->second.dev, (size_t)nel * 2, hipMemcpyDeviceToHost));
    }
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// stream state import: resume a stream from saved caches (states.speech_cache / states.past_key_values of the reference,
// agents/infinisst.py:50-67); also how tests and bench.py put a stream into its steady state without running 40 chunks first
// --------------------------------------------------------------------------------------------
extern "C" int isst_stream_import_llm_kv(isst_handle* h, int id, int layer, const uint16_t* k, const uint16_t* v, int total, int sys_len, int ring_start) {
    if (!h) return ISST_ERR_ARG;
    if (id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", id);
    const isst_config& c = h->cfg;
    if (!k || !v || layer < 0 || layer >= c.llm_layers) return h->fail(ISST_ERR_ARG, "isst_stream_import_llm_kv: bad argument");
    if (total < 0 || sys_len < 0 || sys_len > total || sys_len > h->sys_cap || total - sys_len > h->ring_cap || ring_start < 0 || ring_start >= h->ring_cap)
        return h->fail(ISST_ERR_ARG, "isst_stream_import_llm_kv: total %d / sys_len %d / ring_start %d do not fit the arena (sys %d, ring %d slots)", total, sys_len,
                       ring_start, h->sys_cap, h->ring_cap);
    const int KV = c.llm_kv_heads, slots = h->sys_cap + h->ring_cap;
    std::vector<bf16_t> ks((size_t)KV * slots * 128, 0), vs((size_t)KV * slots * 128, 0);
    for (int kvh = 0; kvh < KV; ++kvh)
        for (int p = 0; p < total; ++p) {
            const long slot = p < sys_len ? p : (long)h->sys_cap + (ring_start + (p - sys_len)) % h->ring_cap;
            std::memcpy(&ks[((size_t)kvh * slots + slot) * 128], k + ((size_t)kvh * total + p) * 128, 256);
            std::memcpy(&vs[((size_t)kvh * slots + slot) * 128], v + ((size_t)kvh * total + p) * 128, 256);
        }
    HIPCHK(hipDeviceSynchronize());
    for (int b = 0; b < h->max_beams; ++b) {  // the arenas of a stream's beams are identical between chunks
        const long base = h->arena_off(id, b) + (long)layer * h->adims.layer_stride;
        HIPCHK(hipMemcpy(h->llm_k + base, ks.data(), ks.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->llm_v + base, vs.data(), vs.size() * 2, hipMemcpyHostToDevice));
    }
    StreamState& s = h->streams[id];
    s.llm_total = total; s.llm_sys = sys_len; s.llm_ring_start = ring_start;
    if (s.chunks == 0) s.chunks = 1;
    return ISST_OK;
}

extern "C" int isst_stream_import_enc_kv(isst_handle* h, int id, int layer, const uint16_t* k, const uint16_t* v, int len, int n_steps, int ring_start) {
    if (!h) return ISST_ERR_ARG;
    if (id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", id);
    const isst_config& c = h->cfg;
    if (!k || !v || layer < 0 || layer >= c.enc_layers) return h->fail(ISST_ERR_ARG, "isst_stream_import_enc_kv: bad argument");
    const int cap = h->enc_cap, H = c.enc_heads;
    if (len < 0 || len > cap || n_steps < len || ring_start < 0 || ring_start >= cap)
        return h->fail(ISST_ERR_ARG, "isst_stream_import_enc_kv: len %d / n_steps %d / ring_start %d do not fit a ring of %d slots", len, n_steps, ring_start, cap);
    std::vector<bf16_t> ks((size_t)H * cap * 64, 0), vs((size_t)H * 64 * cap, 0);  // K [heads][cap][64], V transposed [heads][64][cap]
    for (int hd = 0; hd < H; ++hd)
        for (int j = 0; j < len; ++j) {
            const int slot = (ring_start + j) % cap;
            std::memcpy(&ks[((size_t)hd * cap + slot) * 64], k + ((size_t)hd * len + j) * 64, 128);
            for (int d = 0; d < 64; ++d) vs[((size_t)hd * 64 + d) * cap + slot] = v[((size_t)hd * len + j) * 64 + d];
        }
    HIPCHK(hipDeviceSynchronize());
    const size_t base = (size_t)id * h->enc_stream_stride + (size_t)layer * h->enc_layer_stride;
    HIPCHK(hipMemcpy(h->enc_k + base, ks.data(), ks.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->enc_v + base, vs.data(), vs.size() * 2, hipMemcpyHostToDevice));
    StreamState& s = h->streams[id];
    s.enc_start = ring_start; s.enc_len = len; s.enc_steps = n_steps;
    if (s.chunks == 0) s.chunks = 1;
    return ISST_OK;
}

extern "C" int isst_stream_import_audio_history(isst_handle* h, int id, const uint16_t* samples, int n) {
    if (!h) return ISST_ERR_ARG;
    if (id < 0 || id >= (int)h->streams.size() || !h->streams[id].open) return h->fail(ISST_ERR_ARG, "bad stream id %d", id);
    if (!samples || n != h->hist) return h->fail(ISST_ERR_ARG, "isst_stream_import_audio_history: exactly %d samples (receptive field - 1) are kept", h->hist);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(h->audio_hist + (size_t)id * round_up(h->hist, 8), samples, (size_t)n * 2, hipMemcpyHostToDevice));
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// beam-search test aid: candidate trace + teacher forcing of a one-stream call
// --------------------------------------------------------------------------------------------
extern "C" int isst_debug_beam_trace_begin(isst_handle* h, int num_beams, const int* forced_tokens, const int* forced_parents, int n_steps) {
    if (!h || num_beams < 1 || n_steps < 0 || (n_steps > 0 && (!forced_tokens || !forced_parents))) return h ? h->fail(ISST_ERR_ARG, "isst_debug_beam_trace_begin: bad argument") : ISST_ERR_ARG;
    h->btrace_on = true;
    h->btrace_beams = num_beams;
    h->btrace.clear();
    h->bforce_tok.assign(forced_tokens, forced_tokens + (size_t)n_steps * num_beams);
    h->bforce_par.assign(forced_parents, forced_parents + (size_t)n_steps * num_beams);
    return ISST_OK;
}
extern "C" int isst_debug_beam_trace_step(isst_handle* h, int step, int* rows, int* n_keep, float* top_val, int* top_idx, float* beam_scores, int max_elems) {
    if (!h || !rows || !n_keep) return ISST_ERR_ARG;
    if (step < 0 || step >= (int)h->btrace.size()) return h->fail(ISST_ERR_NOTFOUND, "no beam trace for step %d (%d recorded)", step, (int)h->btrace.size());
    const auto& t = h->btrace[step];
    *rows = t.rows; *n_keep = t.n_keep;
    const int ne = t.rows * t.n_keep;
    if (top_val && top_idx && max_elems >= ne) {
        std::memcpy(top_val, t.val.data(), sizeof(float) * ne);
        std::memcpy(top_idx, t.idx.data(), sizeof(int) * ne);
    }
    if (beam_scores && max_elems >= t.rows) std::memcpy(beam_scores, t.score.data(), sizeof(float) * t.rows);
    return ISST_OK;
}
extern "C" int isst_debug_beam_trace_end(isst_handle* h, int* n_steps) {
    if (!h) return ISST_ERR_ARG;
    if (n_steps) *n_steps = (int)h->btrace.size();
    h->btrace_on = false;
    h->bforce_tok.clear();
    h->bforce_par.clear();
    return ISST_OK;
}

// --------------------------------------------------------------------------------------------
// kernel-level entry points of the splice and the two attention kernels (parity tests replay the reference-generated fixtures
// tests/golden/{splice,encoder,llm_attention}.npz through them); all pointers are DEVICE pointers
// --------------------------------------------------------------------------------------------
extern "C" int isst_op_embed_splice(const int* ids, const int* speech_row, const uint16_t* table, const uint16_t* speech, uint16_t* out, int rows, int D,
                                    void* hip_stream) {
    if (!ids || !table || !out) return ISST_ERR_ARG;
    return launch_embed_splice(ids, speech_row, table, speech, out, rows, D, reinterpret_cast<hipStream_t>(hip_stream));
}

extern "C" int isst_op_enc_attention(const uint16_t* qkv, uint16_t* kring, uint16_t* vring, int ring_start, int prefix, const float* rope_cos,
                                     const float* rope_sin, int rope_round_each, uint16_t* out, int Q, int heads, int cap, int max_cache, int blocksize,
                                     void* hip_stream) {
    if (!qkv || !kring || !vring || !rope_cos || !rope_sin || !out || ring_start < 0 || ring_start >= cap || prefix < 0) return ISST_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    EncStreamView hv{ring_start, prefix};
    EncStreamView* dv = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&dv), sizeof hv) != hipSuccess) return ISST_ERR_NOMEM;
    int rc = ISST_ERR_HIP;
    if (hipMemcpy(dv, &hv, sizeof hv, hipMemcpyHostToDevice) == hipSuccess)
        rc = launch_enc_attention(qkv, kring, vring, 0, dv, rope_cos, rope_sin, rope_round_each, out, 1, Q, heads, cap, max_cache, blocksize, st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == ISST_OK) rc = ISST_ERR_HIP;
    (void)hipFree(dv);
    return rc;
}

extern "C" int isst_op_llm_attention(const uint16_t* qkv, int rows, int pos0, uint16_t* kpool, uint16_t* krpool, uint16_t* vpool, int heads, int kv_heads,
                                     int sys_cap, int ring_cap, int sys_len, int ring_start, const uint16_t* rope_cos, const uint16_t* rope_sin, int rot_keys,
                                     uint16_t* out, void* hip_stream) {
    if (!qkv || !kpool || !krpool || !vpool || !rope_cos || !rope_sin || !out || rows < 1 || pos0 < 0 || heads < 1 || kv_heads < 1 || heads % kv_heads)
        return ISST_ERR_ARG;
    const int G = heads / kv_heads, slots = sys_cap + ring_cap;
    if ((G != 1 && G != 2 && G != 4) || slots % 64 || sys_cap % 16 || sys_len < 0 || sys_len > sys_cap || ring_start < 0 || ring_start >= ring_cap ||
        pos0 + rows - sys_len > ring_cap)
        return ISST_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    LlmAttnDims d{};
    d.heads = heads; d.kv_heads = kv_heads; d.sys_cap = sys_cap; d.ring_cap = ring_cap; d.layer_stride = (long)kv_heads * slots * 128;
    LlmStreamView v{};
    v.sys_len = sys_len; v.ring_start = ring_start; v.kv_offset = 0; v.new_start = pos0; v.row0 = 0; v.rot_keys = rot_keys ? 1 : 0;
    const int gmax = LLM_ATTN_GROUP_ROWS(G);
    std::vector<int> row_stream(rows, 0), row_pos(rows);
    for (int r = 0; r < rows; ++r) row_pos[r] = pos0 + r;
    std::vector<int2> groups, units;
    for (int t = 0; t < rows; t += gmax) groups.push_back(make_int2(t, std::min(gmax, rows - t)));
    int max_unit_groups = 0;
    for (int g0 = 0; g0 < (int)groups.size(); g0 += 8) {
        units.push_back(make_int2(g0, std::min(8, (int)groups.size() - g0)));
        max_unit_groups = std::max(max_unit_groups, units.back().y);
    }
    const size_t off_pos = sizeof(int) * rows, off_view = off_pos + sizeof(int) * rows, off_groups = (off_view + sizeof v + 15) / 16 * 16,
                 off_units = off_groups + sizeof(int2) * groups.size(), meta_bytes = off_units + sizeof(int2) * units.size();
    std::vector<unsigned char> hostm(meta_bytes);
    std::memcpy(hostm.data(), row_stream.data(), sizeof(int) * rows);
    std::memcpy(hostm.data() + off_pos, row_pos.data(), sizeof(int) * rows);
    std::memcpy(hostm.data() + off_view, &v, sizeof v);
    std::memcpy(hostm.data() + off_groups, groups.data(), sizeof(int2) * groups.size());
    std::memcpy(hostm.data() + off_units, units.data(), sizeof(int2) * units.size());
    unsigned char* meta = nullptr;
    float* partial = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&meta), meta_bytes) != hipSuccess) return ISST_ERR_NOMEM;
    if (hipMalloc(reinterpret_cast<void**>(&partial), sizeof(float) * (size_t)rows * heads * (slots / 64) * ATTN_SLAB) != hipSuccess) { (void)hipFree(meta); return ISST_ERR_NOMEM; }
    int rc = hipMemcpy(meta, hostm.data(), meta_bytes, hipMemcpyHostToDevice) == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    const LlmStreamView* dv = reinterpret_cast<const LlmStreamView*>(meta + off_view);
    if (rc == ISST_OK && rot_keys && pos0 > 0) rc = launch_llm_rope_cache(dv, 1, rope_cos, rope_sin, kpool, krpool, d, 1, st);
    if (rc == ISST_OK) {
        LlmAttnOne one{};
        if (groups.size() == 1) { one.enabled = 1; one.grp = groups[0]; one.pos0 = pos0; one.v = v; }
        rc = launch_llm_attention(qkv, reinterpret_cast<const int*>(meta), reinterpret_cast<const int*>(meta + off_pos), dv,
                                  reinterpret_cast<const int2*>(meta + off_groups), (int)groups.size(), gmax, rope_cos, rope_sin, kpool, krpool, vpool, partial, out, d,
                                  0, rows, st, &one, reinterpret_cast<const int2*>(meta + off_units), (int)units.size(), max_unit_groups, 0);
    }
    if (hipStreamSynchronize(st) != hipSuccess && rc == ISST_OK) rc = ISST_ERR_HIP;
    (void)hipFree(meta);
    (void)hipFree(partial);
    return rc;
}

extern "C" int isst_op_splice_map(const int* ids, int len, int user_id, int assistant_id, int start_header_id, int n_features, int* row_src, int* n_rows) {
    if (!ids || len < 1 || !row_src || !n_rows || n_features < 0) return ISST_ERR_ARG;
    std::vector<int> desc;
    const int rc = splice_rows(ids, len, user_id, assistant_id, start_header_id, n_features, desc);
    if (rc != ISST_OK) return rc;
    *n_rows = (int)desc.size();  // <= len
    std::memcpy(row_src, desc.data(), sizeof(int) * desc.size());
    return ISST_OK;
}

// split-KV merge, as the combine launch and as the o_proj GEMV's A-staging prologue (parity test: the two must give the same bits)
extern "C" int isst_op_attn_combine(const float* partial, uint16_t* out, int heads, int rows, int n_splits, void* hip_stream) {
    if (!partial || !out) return ISST_ERR_ARG;
    return launch_llm_attn_combine(partial, out, heads, rows, n_splits, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_gemm_attn_merge(const float* partial, int n_splits, const uint16_t* packed, const uint16_t* res, int64_t ldres, uint16_t* out,
                                       int64_t ldo, int M, int N, int K, void* hip_stream) {
    if (!partial || !packed || !out) return ISST_ERR_ARG;
    GemmArgs g{};
    g.lda = K; g.Wp = packed; g.res = res; g.ldres = ldres; g.out = out; g.ldo = ldo;
    if (!res) return ISST_ERR_ARG;
    g.M = M; g.N = round_up(N, 16); g.K = K; g.batch = 1; g.epi = EPI_RES; g.n_valid = N;
    g.attn_partial = partial; g.attn_splits = n_splits;
    return launch_gemm(g, reinterpret_cast<hipStream_t>(hip_stream));
}
