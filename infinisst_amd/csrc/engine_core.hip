// Host side of libinfinisst_hip.so, part 1 of 4 (engine_internal.h): the handle -- configuration, device weights (re-laid out for the MFMA GEMM),
// per-stream state (audio history, encoder KV rings, LLM KV arenas), eviction, imports and debug reads.  Reference: agents/infinisst.py:50-67,115-123,
// 179-180, 354-361.  No CPU fallback exists.
#include "engine_internal.h"

namespace isst_impl {
std::string g_create_error;
}

namespace {

const char* ENC = "model.speech_encoder.speech_encoder.";
const char* SHR = "model.speech_encoder.length_shrink.";
const char* PRJ = "model.speech_encoder.proj.";


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
    if (h->meta_host2) (void)hipHostFree(h->meta_host2);
    if (h->bst_host) (void)hipHostFree(h->bst_host);
    if (h->bpow_host) (void)hipHostFree(h->bpow_host);
    if (h->bforce_host) (void)hipHostFree(h->bforce_host);
    if (h->blog) (void)hipHostFree(h->blog);
    for (hipEvent_t e : h->prof_ev) (void)hipEventDestroy(e);
    if (h->side_ev) (void)hipEventDestroy(h->side_ev);
    if (h->side_ev2) (void)hipEventDestroy(h->side_ev2);
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
    if (const char* e = getenv("ISST_GEMM_TUNING")) {  // A/B aid: comma-separated gemm_set_tuning codes (e.g. 800011 = gemm_dense with 256-row tiles only)
        for (const char* q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr) gemm_set_tuning(atoi(q), 0);
    }
    if (const char* e = getenv("ISST_ROPE_SIDE")) h->rope_side = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_ROPE_FUSE")) h->rope_fuse = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSED_SAMPLE")) h->fused_sample = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_TAIL_ADVANCE")) h->tail_advance = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_SYNC_AT_END")) h->sync_at_end = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_QKV_SLICES")) h->qkv_slices = atoi(e) >= 1 && atoi(e) <= 8 ? atoi(e) : 1;
    if (const char* e = getenv("ISST_INLINE_COMBINE")) h->inline_combine = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_BEAM_SHARED")) h->beam_shared = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSE_ATTN_OPROJ")) { h->fuse_attn_oproj = e[0] && e[0] != '0'; h->fuse_ao_mode = e[0] == '2' ? 1 : 0; h->fuse_ao_beams = e[0] != '1' && e[0] != '2'; }  // 0: three launches; 1: one row only; 2: bisecting aid; 3 (= default): beam groups too
    if (const char* e = getenv("ISST_FUSE_AO_TEST_TIMEOUT")) h->fuse_ao_test_timeout = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_FUSE_AO_DELAY")) h->fuse_ao_delay = atoi(e) >= 0 && atoi(e) <= 64 ? atoi(e) : 0;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) h->n_cus = prop.multiProcessorCount;
    }
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
        // (+ 2 ffn x dim bf16 per layer -- 7.5 GB at Llama-3.1-8B size, of 288 -- for the copy the one-row passes stream: engine_llm.hip llm_forward)
        static const bool gateup8_on = !(getenv("ISST_GATEUP8") && atoi(getenv("ISST_GATEUP8")) == 0);
        if (gateup8_on && c.llm_ffn % 8 == 0) {
            ok = ok && alloc_linear(h, L.gateup8, 2 * c.llm_ffn, DL, false);
            L.gateup8.n_valid = c.llm_ffn;
        }
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
    h->enc_cs = h->dalloc<bf16_t>((size_t)h->enc_rope_rows * 64, true);
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
    h->fuse_bar = h->dalloc<unsigned>(40 * 32, true);
    h->fuse_row = h->dalloc<unsigned>((size_t)4 * H * 64 * 2, true);
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
        h->bseq[0] = h->dalloc<int>(NB * h->max_ids);
        h->bseq[1] = h->dalloc<int>(NB * h->max_ids);
        h->bst_dev = h->dalloc<BeamDevStream>(ns);
        h->bpow_dev = h->dalloc<double>((size_t)c.max_new_tokens + 2);
        h->bops_dev[0] = h->dalloc<KvCopyOp>(NB * 2);
        h->bops_dev[1] = h->dalloc<KvCopyOp>(NB);
        h->bop_counts_dev = h->dalloc<int>(((size_t)c.max_new_tokens + 1) * 2, true);
        h->breorder_dev = h->dalloc<BeamReorder>(ns, true);
        h->bticket_dev = h->dalloc<int>(4, true);
        h->bforce_dev = h->dalloc<int>((size_t)2 * c.max_new_tokens * h->max_beams);
        h->bview.side_cap = h->max_ids;
        h->bview.logz = h->dalloc<float>(NB);
        h->bview.side_tok = h->dalloc<int>(NB * h->max_ids);
        h->bview.side_val = h->dalloc<float>(NB * h->max_ids);
        h->bview.side_n = h->dalloc<int>(NB, true);
        if (!h->tbuf_k || !h->tbuf_v || !h->tbuf_kr || !h->lse_max || !h->lse_sum || !h->cand_val || !h->cand_idx || !h->top_val || !h->top_idx || !h->bseq[0] || !h->bseq[1] ||
            !h->bst_dev || !h->bpow_dev || !h->bops_dev[0] || !h->bops_dev[1] || !h->bop_counts_dev || !h->breorder_dev || !h->bticket_dev || !h->bforce_dev || !h->bview.logz || !h->bview.side_tok ||
            !h->bview.side_val || !h->bview.side_n) {
            h->fail(ISST_ERR_NOMEM, "beam search allocation failed"); return die(ISST_ERR_NOMEM);
        }
    }
    h->meta_bytes = (size_t)LR * 8 * sizeof(int) + NB * (sizeof(int) + sizeof(LlmStreamView) + sizeof(SampleStream) + sizeof(EncStreamView)) +
                    NB * (h->max_ids + h->max_enc_ids) * sizeof(int) + NB * 4 * sizeof(KvCopyOp) * KV_OPS_SLOTS + 65536 * sizeof(int) + 8192;
    h->meta_dev = h->dalloc<unsigned char>(h->meta_bytes);
    if (h->max_beams > 1) {
        h->meta_dev2 = h->dalloc<unsigned char>(h->meta_bytes);
        if (!h->meta_dev2) { h->fail(ISST_ERR_NOMEM, "beam search allocation failed"); return die(ISST_ERR_NOMEM); }
    }
    const void* must[] = {h->audio_hist, h->enc_k, h->enc_v, h->llm_k, h->llm_v, h->llm_kr, h->enc_cos, h->enc_sin, h->enc_cs, h->llm_cos, h->llm_sin, h->pcm_f32,
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
    if (h->max_beams > 1) {
        // per-step slot of the pinned beam log: candidates [NB][BEAM_TOPK] (values, ids), decisions [NB], status per stream; behind the slots one 64-byte line
        // with the sequence number of the last beam_select launch that has published
        h->blog_steps = c.max_new_tokens + 1;
        h->blog_slot_bytes = (NB * BEAM_TOPK * 8 + NB * sizeof(BeamDecision) + (size_t)ns * sizeof(int) + 63) / 64 * 64;
        if (hipHostMalloc(reinterpret_cast<void**>(&h->meta_host2), h->meta_bytes) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&h->bst_host), sizeof(BeamDevStream) * ns) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&h->bpow_host), sizeof(double) * ((size_t)c.max_new_tokens + 2)) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&h->bforce_host), sizeof(int) * 2 * c.max_new_tokens * h->max_beams) != hipSuccess ||
            hipHostMalloc(reinterpret_cast<void**>(&h->blog), h->blog_slot_bytes * h->blog_steps + 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
            h->fail(ISST_ERR_NOMEM, "pinned host allocation failed (beam search)"); return die(ISST_ERR_NOMEM);
        }
        std::memset(h->blog, 0, h->blog_slot_bytes * h->blog_steps + 64);
    }
    if (const char* e = getenv("ISST_BEAM_DEVICE")) h->beam_device = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_BEAM_ONE_COPY")) h->beam_one_copy = e[0] && e[0] != '0';
    if (const char* e = getenv("ISST_BEAM_LEAN_TAIL")) h->beam_lean_tail = e[0] && e[0] != '0';
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
        if (!strcmp(suf, "mlp.gate_proj.weight")) {
            if (!shape_is(ndim, shape, {c.llm_ffn, DL})) return bad_shape();
            rc = pack_into(h, L.gateup, src, c.llm_ffn, 0, 2, 0, 0);
            if (rc == ISST_OK && L.gateup8.wp) rc = launch_pack_weight_half(src, L.gateup8.wp, c.llm_ffn, DL, 0, 0);
        }
        if (!strcmp(suf, "mlp.up_proj.weight")) {
            if (!shape_is(ndim, shape, {c.llm_ffn, DL})) return bad_shape();
            rc = pack_into(h, L.gateup, src, c.llm_ffn, 0, 2, 1, 0);
            if (rc == ISST_OK && L.gateup8.wp) rc = launch_pack_weight_half(src, L.gateup8.wp, c.llm_ffn, DL, 1, 0);
        }
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
    // The reference's production setting casts the encoder to bf16 (agents/infinisst.py:173), so its rotary module's cos / sin ARE bf16 numbers (rope.py "bf16" mode).
    // Then the attention kernel reads them from one packed bf16 table [position][k-step][8-dim group][cos x 4 | sin x 4] -- same values, half the bytes, one 16-byte
    // load per group instead of two (enc_attn.hip EncTab); tables with any other fp32 value (rope.py "fp32" mode) are read as handed over.  ISST_ENC_TAB_BF16=0: always fp32.
    {
        const size_t n = (size_t)h->enc_rope_rows * 32;
        auto is_bf16 = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (u & 0xffffu) == 0; };
        bool exact = !(getenv("ISST_ENC_TAB_BF16") && atoi(getenv("ISST_ENC_TAB_BF16")) == 0);
        for (size_t i = 0; exact && i < n; ++i) exact = is_bf16(enc_cos[i]) && is_bf16(enc_sin[i]);
        h->enc_cs_valid = false;
        if (exact) {
            auto top = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return (uint16_t)(u >> 16); };
            std::vector<uint16_t> cs((size_t)h->enc_rope_rows * 64);
            for (int pos = 0; pos < h->enc_rope_rows; ++pos)
                for (int g = 0; g < 8; ++g)  // g = k-step * 4 + group: pairs 4 g .. 4 g + 3
                    for (int i = 0; i < 4; ++i) {
                        cs[((size_t)pos * 8 + g) * 8 + i] = top(enc_cos[(size_t)pos * 32 + g * 4 + i]);
                        cs[((size_t)pos * 8 + g) * 8 + 4 + i] = top(enc_sin[(size_t)pos * 32 + g * 4 + i]);
                    }
            HIPCHK(hipMemcpy(h->enc_cs, cs.data(), cs.size() * 2, hipMemcpyHostToDevice));
            h->enc_cs_valid = true;
        }
    }
    HIPCHK(hipMemcpy(h->llm_cos, llm_cos, (size_t)h->llm_rope_rows * 64 * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->llm_sin, llm_sin, (size_t)h->llm_rope_rows * 64 * 2, hipMemcpyHostToDevice));
    h->rope_set = true;
    return ISST_OK;
}

// position held by row `row` of the --rope 0 table (rows above 255 are consecutive bf16 bit patterns from 256.0 = 0x4380)
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
    HIPCHK(hipDeviceSynchronize());  // (isst_generate may return with the tail of its last kernel still running on the caller's stream)
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
namespace isst_impl {

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

int check_ready(isst_handle* h) {
    if (!h->finalized) return h->fail(ISST_ERR_STATE, "weights not finalized (isst_finalize_weights)");
    return ISST_OK;
}

}  // namespace isst_impl

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
