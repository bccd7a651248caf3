// Host side of libinfinisst_hip.so, part 4 of 4 (engine_internal.h): the per-kernel entry points of the C ABI (isst_op_*: parity tests drive every kernel
// through them) and the profiling hooks of bench.py.
#include "engine_internal.h"

extern "C" int64_t isst_op_packed_elems(int n_rows, int K) { return (int64_t)round_up(n_rows, 16) * K; }

extern "C" int isst_op_pack_weight(const uint16_t* w, uint16_t* packed, int n_rows, int K, int conv_k, void* hip_stream) {
    return launch_pack_weight(w, packed, n_rows, K, 0, 1, 0, conv_k, reinterpret_cast<hipStream_t>(hip_stream));
}
extern "C" int isst_op_pack_gateup8(const uint16_t* gate, const uint16_t* up, uint16_t* packed, int ffn, int K, void* hip_stream) {
    if (!gate || !up || !packed) return ISST_ERR_ARG;
    const int rc = launch_pack_weight_half(gate, packed, ffn, K, 0, reinterpret_cast<hipStream_t>(hip_stream));
    return rc != ISST_OK ? rc : launch_pack_weight_half(up, packed, ffn, K, 1, reinterpret_cast<hipStream_t>(hip_stream));
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
extern "C" int isst_op_set_reduce_tuning(int rms_lf_rows, int ln_lf_rows) {
    reduce_set_tuning(rms_lf_rows, ln_lf_rows);
    return ISST_OK;
}
extern "C" int isst_op_set_attn_tuning(int target_workgroups) {
    llm_attn_set_tuning(target_workgroups);
    return ISST_OK;
}
extern "C" int isst_op_topk_rows(const float* scores, long ld, int vocab, int k, int rows, float* out_val, int* out_idx, void* hip_stream) {
    // test entry of beam.hip's per-row top-k (ties -> lowest index): out_val / out_idx are [rows][32] device arrays, the first k entries of a row are filled
    if (!scores || !out_val || !out_idx || rows < 1 || vocab < 1 || k < 1 || k > BEAM_TOPK || ld < vocab) return ISST_ERR_ARG;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    float* cv = nullptr;
    int* ci = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&cv), sizeof(float) * (size_t)rows * 64 * BEAM_TOPK) != hipSuccess) return ISST_ERR_NOMEM;
    if (hipMalloc(reinterpret_cast<void**>(&ci), sizeof(int) * (size_t)rows * 64 * BEAM_TOPK) != hipSuccess) { (void)hipFree(cv); return ISST_ERR_NOMEM; }
    int rc = launch_topk_rows(scores, ld, vocab, k, cv, ci, out_val, out_idx, rows, st);
    if (hipStreamSynchronize(st) != hipSuccess && rc == ISST_OK) rc = ISST_ERR_HIP;
    (void)hipFree(cv);
    (void)hipFree(ci);
    return rc;
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
    for (int g0 = 0; g0 < (int)groups.size(); g0 += LLM_PREFILL_UNIT_GROUPS) {
        units.push_back(make_int2(g0, std::min(LLM_PREFILL_UNIT_GROUPS, (int)groups.size() - g0)));
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
        if (groups.size() == 1) { one.enabled = 1; one.grp = groups[0]; one.pos0 = pos0; one.pos_step = 1; one.v = v; }
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

