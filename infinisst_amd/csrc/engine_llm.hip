// Host side of libinfinisst_hip.so, part 3 of 4 (engine_internal.h): the Llama side of one chunk -- splice, prefill, the greedy / sample loop, beam search.
// It is the MI355X replacement of `self.model.generate(...)` at reference agents/infinisst.py:307-332 (model/llm.py:51-126,192-270,
// model/patches/patch_llm.py:231-336, model/patches/patch_hf.py:43-302,586-624,687-967).  Every compute step is a HIP kernel launch on the caller's stream.
#include "engine_internal.h"

namespace isst_impl {

int gemm(isst_handle* h, const bf16_t* A, long lda, const PackedLinear& L, int epi, const bf16_t* res, long ldres, void* out, long ldo,
         int M, hipStream_t st, int batch, long a_batch, long out_batch, long res_batch,
         const bf16_t* norm_w, float norm_eps, float* ssq) {
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
    if (rows > 64) {  // A/B aid: ISST_KSPLIT="K:N:s,..." forces s slices for the projection K -> N at many rows (in-situ comparisons of slice counts)
        if (const char* e = getenv("ISST_KSPLIT")) {
            int k = 0, n = 0, sv = 0;
            for (const char* q = e; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : nullptr)
                if (sscanf(q, "%d:%d:%d", &k, &n, &sv) == 3 && k == K && n == N && sv >= 1 && (long)sv * rows * N <= slab_cap) return sv;
        }
    }
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
        // (Round 6 tried the unsplit launch on 128-row tiles instead -- 176 workgroups that walk all of K, the residual in the GEMM's epilogue and a plain norm
        //  launch behind it.  A probe with the weights resident in the Infinity Cache liked it (o_proj 49.5 + 8 us against 67 in two slices); INSIDE a step,
        //  weights from HBM, the half tiles -- which stream the same weight bytes for half the MFMAs and are bound by the L2 -> LDS path -- took 63.7 + 6.0 us
        //  against ~50 + 16.5, and the encoder's out_proj / fc2 at 3072 rows lost the same way: +0.44 ms per 64-stream step, same box, kernel traces of both
        //  libraries (profiles/r06/trace_same_box_*.txt).  The slices stay.)
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
                 bf16_t* x, long ldx, float* ssq, bf16_t* plain_out, long ld_plain,
                 const bf16_t* norm_w, float norm_eps, float* ssq_in) {
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

}  // namespace isst_impl

namespace {

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
                hipStream_t st, const StepMeta* hm = nullptr, int n_units = 0, int max_unit_groups = 0, int n_beam_wgs = 0, bool rows_embedded = false) {
    const isst_config& c = h->cfg;
    const int DL = c.llm_dim, H = c.llm_heads, KV = c.llm_kv_heads;
    // one group (one stream's decode step): its metadata travels in the kernel arguments (llm_attn.hip LlmAttnOne)
    LlmAttnOne one{};
    if (hm && n_groups == 1) {
        const int2 g0 = hm->groups[0];
        bool consecutive = true, same = true;  // (a prompt's rows / one decode row; the beams of a stream all sit at one position)
        for (int k = 1; k < g0.y; ++k) {
            consecutive = consecutive && hm->row_pos[g0.x + k] == hm->row_pos[g0.x] + k;
            same = same && hm->row_pos[g0.x + k] == hm->row_pos[g0.x];
        }
        if (consecutive || same) {
            one.enabled = 1;
            one.grp = g0;
            one.pos0 = hm->row_pos[g0.x];
            one.pos_step = consecutive ? 1 : 0;
            one.v = hm->views[hm->row_stream[g0.x]];
        }
    }
    // (rows_embedded: h->lx already holds the rows -- one stream's decode step, whose sampling tail copied the new token's embedding there)
    if (!rows_embedded) CHK(launch_embed_splice(d.ids, splice ? d.speech_row : nullptr, h->embed, h->speech, h->lx, rows, DL, st));
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
        // one stream's decode step: attention, combine and o_proj (+ residual) are ONE launch (llm_attn.hip llm_attn_oproj_kernel)
        const bool fused_ao = h->fuse_attn_oproj && (n_beam_wgs == 0 || h->fuse_ao_beams) && !h->fuse_combine && !h->inline_combine && so == 1 && n_units == 0 && !tap_prefix && L.o.n_valid == L.o.N &&
                              llm_attn_oproj_supported(h->adims, rows, n_groups, &one, L.o.N, L.o.K, h->n_cus, n_beam_wgs) > 0;
        // one or two decode rows: no combine launch -- o_proj merges the split-KV partials while it stages its A row (gemm.hip AMODE 3)
        int merge_splits = 0;
        const bool merge_in_oproj = h->fuse_combine && !h->inline_combine && so == 1 && rows <= ATTN_MERGE_MAX_ROWS && n_units == 0 && n_beam_wgs == 0;
        if (fused_ao) {
            h->fuse_ao_used = true;
            CHK(launch_llm_attn_oproj(h->lqkv, h->llm_cos, h->llm_sin, h->llm_k, h->llm_kr, h->llm_v, h->lpartial, h->adims, l, one, n_beam_wgs, L.o.wp, L.o.N, L.o.K,
                                      L.o.n_valid, h->lx + (long)one.grp.x * DL, h->lx + (long)one.grp.x * DL, DL, h->lattn + (long)one.grp.x * H * 128, h->fuse_row, h->fuse_bar,
                                      h->tok_host + h->tok_cap + 8, h->n_cus, st, &h->fuse_arrive_total, &h->fuse_merge_total, h->fuse_ao_mode, h->fuse_ao_delay, h->fuse_ao_test_timeout ? 1000000u : 0u));
        } else
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
            if (fused_ao && h->fuse_ao_mode == 0) {
            } else if (merge_splits > 0) {
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
                // One or two rows (the decode step of one or two greedy streams): the self-paired copy of gate / up -- 1792 one-tile workgroups, 7 on every CU, where
                // the tile pairs are 896 = 3.5 per CU: 36.2 against 38.0 us for the same bytes at one row, 37.9 against 39.7 at two (profiles/r06/gemv_balance_probe.txt,
                // gemv_balance_rows_probe.txt); from three rows on the pairs win (every workgroup normalises the rows it stages: twice the workgroups, twice that work)
                if (rows <= 2 && L.gateup8.wp)
                    CHK(gemm(h, h->lx, DL, L.gateup8, EPI_SWIGLU8, nullptr, 0, h->lact, c.llm_ffn, rows, st, 1, 0, 0, 0, L.post_norm, c.rms_eps));
                else
                CHK(gemm(h, h->lx, DL, L.gateup, EPI_SWIGLU, nullptr, 0, h->lact, c.llm_ffn, rows, st, 1, 0, 0, 0, L.post_norm, c.rms_eps));
                if (pe) HIPCHK(hipEventRecord(pe, st));
            } else {
                CHK(launch_rmsnorm(h->lx, DL, nullptr, L.post_norm, h->lxn, DL, rows, DL, c.rms_eps, st));
                CHK(prof_open(&pe));
                // (5 and 6 rows -- the register-A GEMV behind the norm launch -- also stream the self-paired copy: 39.0 / 39.5 against 40.7 / 40.8 us; at 8 rows the
                //  pairs are level again, every workgroup reads all rows: profiles/r06/gemv_balance_rows_probe.txt)
                if (rows <= 6 && L.gateup8.wp)
                    CHK(gemm(h, h->lxn, DL, L.gateup8, EPI_SWIGLU8, nullptr, 0, h->lact, c.llm_ffn, rows, st));
                else
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


}  // namespace

namespace isst_impl {
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
}  // namespace isst_impl



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
    const double* powtab;  // powtab[l] = pow((double)l, length_penalty): ONE table per call for this scorer and the device's (beam.hip beam_select_kernel)
    std::vector<BeamHyp> beams;
    double worst = 1e9;
    // returns the buffer slot freed by an evicted (or rejected) hypothesis, or -1
    int add(BeamHyp hyp, double sum_logprobs, int generated_len) {
        hyp.score = sum_logprobs / powtab[generated_len];
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
        const double highest = best_sum_logprobs / powtab[cur_len - prompt_len];
        return worst >= highest;
    }
};

struct BeamStream {  // host state of one stream during a beam call
    std::vector<std::vector<int>> seq;  // per beam: prompt + generated
    std::vector<std::vector<int>> seq_next;  // (scratch of the scorer, kept across steps: 64 streams x 4 beams made the scorer's allocations a 270 us GPU-idle gap per step)
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

// One step of the scorer for stream i of a beam call (beam_search_process, patch_hf.py:43-157): merges the rows' top-k candidates, walks them in rank order
// (EOS candidates inside the top B ranks become hypotheses that keep a copy of their beam's tail), decides (token, parent) of the B next beams, updates the
// stream's sequences / scores / done flag.  val / idx: the candidates of the stream's rows_per rows, [rows_per][BEAM_TOPK].  `ops` != null: the tail copies of new
// hypotheses are appended to it (the host drives the search); null: bookkeeping only -- the device scorer issues the copies (beam_decode_device) and this run
// follows it.  forced_lp(b, parent, token, &lp): the processed log-prob of a teacher-forced choice (test aid).
template <typename ForcedLp>
int scorer_step(isst_handle* h, const isst_gen_params* p, BeamStream& S, int i, int sid, int rows_per, int n_keep, int B, int step, int prompt_len, int P0,
                const float* val, const int* idx, std::vector<int>& ntok, std::vector<int>& npar, std::vector<KvCopyOp>* ops, ForcedLp forced_lp) {
    const isst_config& c = h->cfg;
    const int V = c.vocab;
    ntok.clear();
    npar.clear();
    if (S.done) {
        // a finished batch entry is skipped by the scorer (patch_hf.py:83-92: pad tokens, zero scores, its hypotheses untouched)
        // while the other streams of the call go on; its rows still ride through the forward pass (their KV beyond the
        // winner's tail is never read: llm_total is set from the winning hypothesis)
        const int pad = c.n_eos ? c.eos_ids[0] : 0;
        for (int b = 0; b < B; ++b) { ntok.push_back(pad); npar.push_back(b); S.seq[b].push_back(pad); }
        S.score.assign(B, 0.f);
        return ISST_OK;
    }
    struct Cand { float val; long flat; };
    static thread_local std::vector<Cand> cands;
    static thread_local std::vector<float> nscore;
    cands.clear();
    nscore.clear();
    ntok.reserve(B);
    npar.reserve(B);
    for (int b = 0; b < rows_per; ++b)
        for (int j = 0; j < n_keep; ++j) {
            const int id = idx[b * BEAM_TOPK + j];
            if (id < 0 || id >= V) continue;
            cands.push_back({val[b * BEAM_TOPK + j] + S.score[b], (long)b * V + id});
        }
    if (h->btrace_on && i == 0) {
        isst_handle::BeamTraceStep ts;
        ts.rows = rows_per; ts.n_keep = n_keep;
        for (int b = 0; b < rows_per; ++b) {
            for (int j = 0; j < n_keep; ++j) {
                ts.val.push_back(val[b * BEAM_TOPK + j]);
                ts.idx.push_back(idx[b * BEAM_TOPK + j]);
            }
            ts.score.push_back(S.score[b]);
        }
        h->btrace.push_back(std::move(ts));
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b2) { return a.val > b2.val || (a.val == b2.val && a.flat < b2.flat); });  // (a strict total order: flat indices are distinct -- no stable_sort, which allocates a buffer per call)
    if ((int)cands.size() > n_keep) cands.resize(n_keep);
    const int cur_len = (int)S.seq[0].size() + 1;
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
                if (ops) push_copy(h, *ops, sid, b, hyp.buf, P0, step, false);
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
            if (const int rc = forced_lp(b, par, tok, &lp)) return rc;
            ntok[b] = tok; npar[b] = par; nscore[b] = S.score[par] + lp;
        }
    }
    // input_ids = cat(input_ids[beam_idx], tokens)  (:899)
    std::vector<std::vector<int>>& nseq = S.seq_next;  // (capacity survives the swap below: no allocation after the first steps)
    nseq.resize(B);
    for (int b = 0; b < B; ++b) {
        const std::vector<int>& src = S.seq[npar[b]];
        nseq[b].reserve(src.size() + 1 + p->max_new_tokens);
        nseq[b].assign(src.begin(), src.end());
        nseq[b].push_back(ntok[b]);
    }
    S.seq.swap(nseq);
    S.score = nscore;
    return ISST_OK;
}

// the per-call table both scorers divide by: powtab[l] = l ^ length_penalty (host libm)
void fill_powtab(std::vector<double>& t, int max_new, double lp) {
    t.resize((size_t)max_new + 2);
    for (size_t l = 0; l < t.size(); ++l) t[l] = std::pow((double)l, lp);
}

// the fused attention + o_proj launch timed out (another process on the GPU, a CU mask: its workgroups were not all resident): from here on this handle
// runs the three launches -- said once, on stderr -- and the caller re-issues the pass (nothing of the step was committed)
void latch_three_launch_path(isst_handle* h) {
    volatile int* ferr = h->tok_host + h->tok_cap + 8;
    *ferr = 0;
    h->fuse_ao_used = false;
    h->fuse_attn_oproj = false;
    if (h->dgraph.exec) { (void)hipGraphExecDestroy(h->dgraph.exec); h->dgraph.exec = nullptr; }
    std::fprintf(stderr, "[isst] the fused attention + o_proj launch timed out waiting for its own workgroups (is another process using this GPU, or a CU mask set?): "
                         "this handle runs attention, combine and o_proj as three launches from now on (ISST_FUSE_ATTN_OPROJ=0 selects that from the start)\n");
}

// finalize (:159-275): open beams become hypotheses, the best one wins; every arena of a stream ends up holding its winner's tail; outputs + stream lengths
int beam_finalize(isst_handle* h, const isst_gen_params* p, int n, int B, const int* stream_ids, const int* prompt_lens, const std::vector<int>& rows_len,
                  const std::vector<int>& total0, std::vector<BeamStream>& bs, int* const* out_ids, int* out_lens, StepMeta& mh, StepMeta& md, hipStream_t st,
                  bool drain = true) {
    const isst_config& c = h->cfg;
    std::vector<KvCopyOp> ops;
    std::vector<size_t> best_of(n);
    for (int i = 0; i < n; ++i) {
        BeamStream& S = bs[i];
        const int prompt_len = prompt_lens[i];
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
        best_of[i] = best;
    }
    // make every arena of a stream hold its winner's tail -- TWO copy launches for the whole call (all streams' "winner still in an arena: stage it
    // through temporary 0", then all streams' "temporary / hypothesis buffer -> the arenas"), not two per stream: at 64 streams those were 128 launches of
    // 27 us + a 13 us gap each per chunk, 4.5 % of the step (profiles/r04/trace_busy_prof64x4.txt)
    for (int i = 0; i < n; ++i) {
        const BeamHyp& win = bs[i].hyps.beams[best_of[i]];
        if (win.fed > 0 && win.buf < 0) push_copy(h, ops, stream_ids[i], -1 - win.buf, 0, total0[i] + rows_len[i], win.fed, false);
    }
    CHK(flush_copies(h, ops, mh, md, st));
    for (int i = 0; i < n; ++i) {
        const BeamHyp& win = bs[i].hyps.beams[best_of[i]];
        if (win.fed <= 0) continue;
        const int skip_beam = win.buf < 0 ? -1 - win.buf : -1, src_buf = win.buf < 0 ? 0 : win.buf;
        for (int b = 0; b < B; ++b)
            if (b != skip_beam) push_copy(h, ops, stream_ids[i], b, src_buf, total0[i] + rows_len[i], win.fed, true);
    }
    CHK(flush_copies(h, ops, mh, md, st));
    for (int i = 0; i < n; ++i) {
        const BeamHyp& win = bs[i].hyps.beams[best_of[i]];
        const int prompt_len = prompt_lens[i];
        const int P0 = total0[i] + rows_len[i];
        StreamState& ss = h->streams[stream_ids[i]];
        ss.llm_total = P0 + win.fed;
        ss.chunks++;
        const int max_length = prompt_len + p->max_new_tokens;
        std::vector<int> outv(win.tokens.begin() + prompt_len, win.tokens.end());
        if ((int)win.tokens.size() < std::min((int)win.tokens.size() + 1, max_length)) outv.push_back(c.n_eos ? c.eos_ids[0] : 0);
        for (size_t q = 0; q < outv.size(); ++q) out_ids[i][q] = outv[q];
        out_lens[i] = (int)outv.size();
    }
    // (drain == false: the caller has every token on the host already and nothing of the host's state depends on the copies above having run -- the next
    //  call's work is ordered behind them on the stream, and the pinned op slots they were uploaded from are handed out in turn: flush_copies)
    if (drain) {
        HIPCHK(hipStreamSynchronize(st));
        h->kv_ops_used = 0;
    }
    return ISST_OK;
}

// the metadata of one decode pass of a beam call: one row per (stream, beam).  Shared-prefix form (llm_attn.hip, LlmStreamView::n_beams): the B rows of a
// stream are ONE attention group over arena 0 for everything older than this chunk's generated tokens (identical in all arenas) plus one workgroup per beam
// for the <= 4 tiles that differ; otherwise every beam is its own group over its arena.  `fed`: generated tokens already in the caches (the rows sit at P0 + fed)
void beam_pass_rows(isst_handle* h, StepMeta& m, int n, int B, bool shared, const int* stream_ids, const std::vector<int>& total0, const std::vector<int>& rows_len, int fed) {
    for (int i = 0; i < n; ++i) {
        const StreamState& ss = h->streams[stream_ids[i]];
        for (int b = 0; b < B; ++b) {
            const int r = i * B + b;
            m.row_stream[r] = shared ? i * B : r;  // view index
            m.row_pos[r] = total0[i] + rows_len[i] + fed;
            m.last_rows[r] = r;
            m.views[r].sys_len = ss.llm_sys;
            m.views[r].ring_start = ss.llm_ring_start;
            m.views[r].kv_offset = h->arena_off(stream_ids[i], b);
            m.views[r].new_start = m.row_pos[r];
            m.views[r].row0 = shared ? i * B : r;
            m.views[r].rot_keys = h->rot_keys ? 1 : 0;  // every beam's arena carries its rotated keys (pre-pass over all arenas + position copies)
            m.views[r].n_beams = shared ? B : 0;
            m.views[r].tail_start = total0[i] + rows_len[i];
            m.views[r].beam_stride = h->llm_stream_stride;
            if (shared) {
                m.groups[i].x = i * B;
                m.groups[i].y = B;
            } else {
                m.groups[r].x = r;
                m.groups[r].y = 1;
            }
        }
    }
}

// decode phase of a beam call, the HOST deciding every step (do_sample: the warpers and the draws without replacement run on the host; ISST_BEAM_DEVICE=0);
// the prefill (on arena 0 of every stream) has already produced h->logits rows 0..n-1
int beam_decode(isst_handle* h, const isst_gen_params* p, int n, const int* stream_ids, const int* const* prompt_ids, const int* prompt_lens,
                const std::vector<int>& rows_len /* KV entries the prompt wrote (<= prompt_lens after a short splice) */, const int* const* prev_target_ids, const int* n_prev, const std::vector<int>& total0, int* const* out_ids, int* out_lens,
                StepMeta& mh, StepMeta& md, hipStream_t st) {
    const isst_config& c = h->cfg;
    const int B = p->num_beams, V = c.vocab;
    const int n_keep = std::max(2, 1 + c.n_eos) * B;
    if (n_keep > BEAM_TOPK) return h->fail(ISST_ERR_ARG, "beam search keeps %d candidates per step, at most %d are supported", n_keep, BEAM_TOPK);
    const double lp = p->length_penalty == 0.f ? 1.0 : (double)p->length_penalty;
    std::vector<double> powtab;
    fill_powtab(powtab, p->max_new_tokens, lp);
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
        bs[i].hyps.powtab = powtab.data();
        for (int b = B; b < h->nbuf; ++b) bs[i].free_bufs.push_back(b);  // slots 0..B-1 are reorder temporaries
        // the prompt's KV was written to arena 0: replicate it into the other beams' arenas (they are identical before it)
        push_copy(h, ops, stream_ids[i], 0, 0, total0[i], rows_len[i], false);
    }
    CHK(flush_copies(h, ops, mh, md, st));
    for (int i = 0; i < n; ++i)
        for (int b = 1; b < B; ++b) push_copy(h, ops, stream_ids[i], b, 0, total0[i], rows_len[i], true);
    CHK(flush_copies(h, ops, mh, md, st));

    // ISST_HOST_TRACE=2: where the host's time between two forward passes of a beam step goes (one line per call)
    static const bool bt_on = std::getenv("ISST_HOST_TRACE") && std::atoi(std::getenv("ISST_HOST_TRACE")) == 2;
    using bclock = std::chrono::steady_clock;
    double bt_tail = 0, bt_sync = 0, bt_score = 0, bt_copies = 0, bt_meta = 0, bt_enq = 0;
    auto bt_us = [](bclock::time_point a, bclock::time_point b2) { return std::chrono::duration<double, std::micro>(b2 - a).count(); };
    bclock::time_point bt0 = bclock::now(), bt1;
    auto bt_lap = [&](double& acc) { if (bt_on) { bt1 = bclock::now(); acc += bt_us(bt0, bt1); bt0 = bt1; } };
    int step = 0;  // tokens already chosen per beam
    bool relaunched = false;  // the pass whose logits are being scored was re-issued on the three-launch path
    const bool shared = h->beam_shared && B * (c.llm_heads / c.llm_kv_heads) <= 16 && (p->max_new_tokens + 15) / 16 + 1 <= 4;  // (isst_generate's `beams_share_prefix`: the pre-pass relies on it)
    while (true) {
        if (bt_on) bt0 = bclock::now();
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
        } else {
            // (the final selection stores its <= 32 candidates per row straight into the pinned host arrays: no device copy, no two D2H launches per step)
            CHK(launch_topk_rows(h->logits, h->vocab_pad, V, n_keep, h->cand_val, h->cand_idx, h->top_val_host, h->top_idx_host, rows, st));
            bt_lap(bt_tail);
            HIPCHK(hipStreamSynchronize(st));
            bt_lap(bt_sync);
            h->kv_ops_used = 0;  // every earlier copy batch has run
        }
        if (h->fuse_ao_used) {  // the fused launches of the pass that produced these logits have run: did one of them give up waiting?
            h->fuse_ao_used = false;
            if (*reinterpret_cast<volatile int*>(h->tok_host + h->tok_cap + 8) != 0) {
                if (relaunched || step == 0) return h->fail(ISST_ERR_HIP, "the fused attention + o_proj launch timed out and the pass could not be re-issued");
                latch_three_launch_path(h);
                relaunched = true;  // nothing of this step is committed (the host's state moves below): the same pass once more, attention / combine / o_proj as three launches
                beam_pass_rows(h, mh, n, B, shared, stream_ids, total0, rows_len, step - 1);
                for (int i = 0; i < n; ++i)
                    for (int b = 0; b < B; ++b) mh.ids[i * B + b] = bs[i].seq[b].back();
                HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
                CHK(llm_forward(h, md, n * B, n * B, shared ? n : n * B, shared ? B : 1, false, nullptr, st, &mh, 0, 0, shared ? B : 0));
                continue;
            }
        }
        relaunched = false;
        if (p->do_sample) {
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
        }

        // ---- scorer (beam_search_process, :43-157) ----
        bool all_done = true;
        std::vector<std::vector<int>> parents(n), next_tok(n);
        for (int i = 0; i < n; ++i) {
            const int P0 = total0[i] + rows_len[i];  // first position written by the decode phase
            auto forced_lp = [&](int, int par, int tok, float* out) -> int {
                if (p->do_sample) *out = h->samp_host[(size_t)(i * rows_per + par) * h->vocab_pad + tok];  // (the warped row is on the host)
                else HIPCHK(hipMemcpy(out, h->logits + (size_t)(i * rows_per + par) * h->vocab_pad + tok, sizeof(float), hipMemcpyDeviceToHost));
                return ISST_OK;
            };
            if (const int rc = scorer_step(h, p, bs[i], i, stream_ids[i], rows_per, n_keep, B, step, prompt_lens[i], P0, h->top_val_host + (size_t)i * rows_per * BEAM_TOPK,
                                           h->top_idx_host + (size_t)i * rows_per * BEAM_TOPK, next_tok[i], parents[i], &ops, forced_lp))
                return rc;
            all_done = all_done && bs[i].done;
        }
        bt_lap(bt_score);
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
        bt_lap(bt_copies);
        if (all_done || step >= p->max_new_tokens) break;  // :920
        // ---- next forward pass: one row per (stream, beam) ----
        const int nr = n * B;
        beam_pass_rows(h, mh, n, B, shared, stream_ids, total0, rows_len, step - 1);
        for (int i = 0; i < n; ++i)
            for (int b = 0; b < B; ++b) mh.ids[i * B + b] = bs[i].seq[b].back();
        bt_lap(bt_meta);
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
        CHK(llm_forward(h, md, nr, nr, shared ? n : nr, shared ? B : 1, false, nullptr, st, &mh, 0, 0, shared ? B : 0));
        bt_lap(bt_enq);
    }
    if (bt_on)
        std::fprintf(stderr, "[isst beam host] %d streams x %d beams, %d steps: us per step -- sampling tail enqueue %.0f, wait for the candidates %.0f, scorer %.0f, KV copies %.0f, metadata %.0f, forward enqueue %.0f\n",
                     n, B, step, bt_tail / step, bt_sync / step, bt_score / step, bt_copies / step, bt_meta / step, bt_enq / step);
    return beam_finalize(h, p, n, B, stream_ids, prompt_lens, rows_len, total0, bs, out_ids, out_lens, mh, md, st);
}

// decode phase of a beam call with the SCORER ON THE DEVICE (the default; beam.hip beam_select_kernel).  Per step the stream carries
//     log-softmax -> processors -> per-row top-k -> beam_select (decides, reorders the beams' sequences, writes the next pass's token ids / positions and the
//     position-copy lists, logs candidates + choices to pinned memory, publishes a sequence number) -> the two copy launches -> the next forward pass
// and never waits for the host: the host waits for the sequence number (it arrives while the copies still run), reads the done flags, enqueues the next pass --
// whose one-group attention metadata it knows without the tokens -- and only then re-derives the step from the logged candidates with the scorer above
// (the state finalize needs: sequences, hypotheses, their buffers), checking every choice against the device's.  The reference's beam_search_process is a
// host round trip per step by construction (patch_hf.py:43-157 runs in Python between two forward passes).
int beam_decode_device(isst_handle* h, const isst_gen_params* p, int n, const int* stream_ids, const int* const* prompt_ids, const int* prompt_lens,
                       const std::vector<int>& rows_len, const int* const* prev_target_ids, const int* n_prev, const std::vector<int>& total0, int* const* out_ids,
                       int* out_lens, StepMeta& mh, StepMeta& md, hipStream_t st) {
    const isst_config& c = h->cfg;
    const int B = p->num_beams, V = c.vocab;
    const int n_keep = std::max(2, 1 + c.n_eos) * B;
    if (n_keep > BEAM_TOPK) return h->fail(ISST_ERR_ARG, "beam search keeps %d candidates per step, at most %d are supported", n_keep, BEAM_TOPK);
    if (B > BEAM_MAX_B || p->max_new_tokens + 1 > h->blog_steps) return h->fail(ISST_ERR_ARG, "beam search: %d beams / %d new tokens exceed the configured capacity", B, p->max_new_tokens);
    const double lp = p->length_penalty == 0.f ? 1.0 : (double)p->length_penalty;
    std::vector<double> powtab;
    fill_powtab(powtab, p->max_new_tokens, lp);
    std::vector<BeamStream> bs(n);
    std::vector<KvCopyOp> ops;
    // (copy batches of the previous call's finalize may still be in flight -- it does not drain the stream; flush_copies hands the pinned op slots out in turn
    //  and synchronises by itself should they run out)
    for (int i = 0; i < n; ++i) {
        bs[i].seq.assign(B, std::vector<int>(prompt_ids[i], prompt_ids[i] + prompt_lens[i]));
        bs[i].score.assign(B, -1e9f);
        bs[i].score[0] = 0.f;
        bs[i].hyps.num_beams = B;
        bs[i].hyps.powtab = powtab.data();
        for (int b = B; b < h->nbuf; ++b) bs[i].free_bufs.push_back(b);  // slots 0..B-1 are reorder temporaries
        // the prompt's KV was written to arena 0: replicate it into the other beams' arenas (they are identical before it)
        push_copy(h, ops, stream_ids[i], 0, 0, total0[i], rows_len[i], false);
    }
    CHK(flush_copies(h, ops, mh, md, st));
    for (int i = 0; i < n; ++i)
        for (int b = 1; b < B; ++b) push_copy(h, ops, stream_ids[i], b, 0, total0[i], rows_len[i], true);
    CHK(flush_copies(h, ops, mh, md, st));

    // ---- the device's copy of the search state.  The pinned staging blocks written below (bpow_host, bst_host, bforce_host, meta_host2) are free: calls no longer end
    //      with the stream drained, but every upload out of these blocks is enqueued BEFORE the first beam step of its call, and a call only returns after the
    //      host has seen the sequence number of its LAST step (published by beam_select_kernel, which runs behind every one of those uploads on the stream).
    //      Invariant for future edits: no upload from one of these blocks may be enqueued after the last published step of a call. ----
    const bool shared = h->beam_shared && B * (c.llm_heads / c.llm_kv_heads) <= 16 && (p->max_new_tokens + 15) / 16 + 1 <= 4;  // (isst_generate's `beams_share_prefix`: the pre-pass relies on it)
    const int nr = n * B;
    std::memcpy(h->bpow_host, powtab.data(), sizeof(double) * powtab.size());
    HIPCHK(hipMemcpyAsync(h->bpow_dev, h->bpow_host, sizeof(double) * powtab.size(), hipMemcpyHostToDevice, st));
    for (int i = 0; i < n; ++i) {
        BeamDevStream& D = h->bst_host[i];
        std::memset(&D, 0, sizeof D);
        for (int b = 0; b < BEAM_MAX_B; ++b) D.score[b] = b == 0 ? 0.f : -1e9f;
        D.worst = 1e9;
        for (int b = B; b < h->nbuf; ++b) D.free_bufs[D.n_free++] = b;  // (the order of bs[i].free_bufs: both scorers hand out the same slots)
        const StreamState& ss = h->streams[stream_ids[i]];
        D.prompt_len = prompt_lens[i];
        D.P0 = total0[i] + rows_len[i];
        D.sid = stream_ids[i];
        D.sys_len = ss.llm_sys;
        D.ring_start = ss.llm_ring_start;
        D.n_enc = n_prev ? n_prev[i] : 0;
    }
    HIPCHK(hipMemcpyAsync(h->bst_dev, h->bst_host, sizeof(BeamDevStream) * n, hipMemcpyHostToDevice, st));
    const int force_steps = h->btrace_on ? (int)(h->bforce_tok.size() / (size_t)B) : 0;
    if (force_steps > 0) {
        if (force_steps > c.max_new_tokens) return h->fail(ISST_ERR_ARG, "more forced beam steps (%d) than max_new_tokens", force_steps);
        std::memcpy(h->bforce_host, h->bforce_tok.data(), sizeof(int) * force_steps * B);
        std::memcpy(h->bforce_host + (size_t)c.max_new_tokens * h->max_beams, h->bforce_par.data(), sizeof(int) * force_steps * B);
        HIPCHK(hipMemcpyAsync(h->bforce_dev, h->bforce_host, sizeof(int) * 2 * c.max_new_tokens * h->max_beams, hipMemcpyHostToDevice, st));
    }
    HIPCHK(hipMemsetAsync(h->bop_counts_dev, 0, sizeof(int) * 2 * ((size_t)c.max_new_tokens + 1), st));
    // the decode passes' rows (static part; token ids, positions and sequence lengths are written by beam_select step by step)
    StepMeta bh = carve(h, h->meta_host2), bd = carve(h, h->meta_dev2);
    beam_pass_rows(h, bh, n, B, shared, stream_ids, total0, rows_len, 0);
    for (int i = 0; i < n; ++i) {
        const int ne = n_prev ? n_prev[i] : 0;
        if (ne) std::memcpy(bh.enc_pool + (size_t)i * h->max_enc_ids, prev_target_ids[i], (size_t)ne * 4);
        for (int b = 0; b < B; ++b) {
            const int r = i * B + b;
            bh.ids[r] = 0;
            bh.speech_row[r] = -1;
            bh.samp[r].n_ids = 0; bh.samp[r].n_enc = ne;
            bh.samp[r].ids_off = r * h->max_ids; bh.samp[r].enc_off = i * h->max_enc_ids; bh.samp[r].logits_row = r;
        }
    }
    HIPCHK(hipMemcpyAsync(h->meta_dev2 + 4096, h->meta_host2 + 4096, bh.step_bytes - 4096, hipMemcpyHostToDevice, st));

    const size_t NB = (size_t)c.max_streams * h->max_beams;
    auto slot_val = [&](int s) { return reinterpret_cast<float*>(h->blog + (size_t)s * h->blog_slot_bytes); };
    auto slot_idx = [&](int s) { return reinterpret_cast<int*>(h->blog + (size_t)s * h->blog_slot_bytes + NB * BEAM_TOPK * 4); };
    auto slot_dec = [&](int s) { return reinterpret_cast<BeamDecision*>(h->blog + (size_t)s * h->blog_slot_bytes + NB * BEAM_TOPK * 8); };
    auto slot_done = [&](int s) { return reinterpret_cast<int*>(h->blog + (size_t)s * h->blog_slot_bytes + NB * BEAM_TOPK * 8 + NB * sizeof(BeamDecision)); };
    volatile int* seq_word = reinterpret_cast<volatile int*>(h->blog + (size_t)h->blog_steps * h->blog_slot_bytes);
    volatile int* ferr = h->tok_host + h->tok_cap + 8;

    // the host's re-derivation of step s from the log (after the next pass is enqueued: it overlaps the GPU's work)
    std::vector<int> f_tok, f_par;
    auto follow = [&](int s) -> int {
        const int rows_per = s == 0 ? 1 : B;
        const BeamDecision* dec = slot_dec(s);
        const int* dn = slot_done(s);
        for (int i = 0; i < n; ++i) {
            auto forced_lp = [&](int b, int, int, float* out) -> int { *out = dec[i * B + b].forced_lp; return ISST_OK; };
            if (const int rc = scorer_step(h, p, bs[i], i, stream_ids[i], rows_per, n_keep, B, s, prompt_lens[i], total0[i] + rows_len[i], slot_val(s) + (size_t)i * B * BEAM_TOPK,
                                           slot_idx(s) + (size_t)i * B * BEAM_TOPK, f_tok, f_par, nullptr, forced_lp))
                return rc;
            if (dn[i] < 0) return h->fail(ISST_ERR_STATE, "beam search: the device scorer failed at step %d of stream %d (status %d) where the host scorer did not", s, stream_ids[i], dn[i]);
            {   // the hypothesis bookkeeping, which both sides keep independently: kept hypotheses' tail buffers in insertion order + the free list
                int hb[BEAM_MAX_B + 1], nh = 0;
                for (const BeamHyp& hp : bs[i].hyps.beams) hb[nh++] = hp.buf >= 0 ? hp.buf : -1;
                const unsigned want = beam_book_digest(nh, hb, (int)bs[i].free_bufs.size(), bs[i].free_bufs.data());
                if ((unsigned)(dn[i] >> 1) != want)
                    return h->fail(ISST_ERR_STATE, "beam search: device and host scorers keep different hypothesis buffers at step %d of stream %d (digest %x against %x: %d hypotheses, %d free buffers on the host)",
                                   s, stream_ids[i], (unsigned)(dn[i] >> 1), want, nh, (int)bs[i].free_bufs.size());
            }
            for (int b = 0; b < B; ++b) {
                const BeamDecision& d = dec[i * B + b];
                if (d.tok != f_tok[b] || d.par != f_par[b] || std::memcmp(&d.score, &bs[i].score[b], sizeof(float)) != 0 || ((dn[i] & 1) != 0) != bs[i].done)
                    return h->fail(ISST_ERR_STATE, "beam search: device and host scorers disagree at step %d, stream %d, beam %d (device token %d parent %d score %.9g done %d, host token %d parent %d score %.9g done %d)",
                                   s, stream_ids[i], b, d.tok, d.par, (double)d.score, dn[i] & 1, f_tok[b], f_par[b], (double)bs[i].score[b], (int)bs[i].done);
            }
        }
        return ISST_OK;
    };

    static const bool bt_on = std::getenv("ISST_HOST_TRACE") && std::atoi(std::getenv("ISST_HOST_TRACE")) == 2;
    using bclock = std::chrono::steady_clock;
    double bt_tail = 0, bt_wait = 0, bt_enq = 0, bt_follow = 0;
    bclock::time_point bt0 = bclock::now(), bt1;
    auto bt_lap = [&](double& acc) { if (bt_on) { bt1 = bclock::now(); acc += std::chrono::duration<double, std::micro>(bt1 - bt0).count(); bt0 = bt1; } };
    int step = 0;            // tokens already chosen per beam
    bool relaunched = false; // the pass whose logits are being scored was re-issued on the three-launch path
    while (true) {
        if (bt_on) bt0 = bclock::now();
        const int rows_per = step == 0 ? 1 : B;  // step 0: only beam 0 carries a finite score (:767-771)
        const int rows = n * rows_per;
        // ---- log_softmax -> processors (on log-probs) -> per-row top-k, all on device-resident sequences ----
        const SampleStream* samp = step == 0 ? md.samp : bd.samp;
        const int* ids_pool = step == 0 ? md.ids_pool : h->bseq[step & 1];
        const int* enc_pool = step == 0 ? md.enc_pool : bd.enc_pool;
        if (h->beam_lean_tail) {  // two sweeps over the rows' fp32 scores: lse_part, then the top-k scan reading `raw - log Z` through beam_process_kernel's encoding
            CHK(launch_beam_scores(h->logits, h->vocab_pad, V, samp, ids_pool, enc_pool, md.suppress, p->n_suppress, p->repetition_penalty, p->no_repeat_ngram_size,
                                   p->encoder_no_repeat_ngram_size, h->lse_max, h->lse_sum, h->bview, n_keep, h->cand_val, h->cand_idx, h->top_val, h->top_idx, rows, st));
        } else {
            CHK(launch_log_softmax(h->logits, h->vocab_pad, V, h->lse_max, h->lse_sum, rows, st));
            CHK(launch_sample_process(h->logits, h->vocab_pad, samp, ids_pool, enc_pool, md.suppress, p->n_suppress, p->repetition_penalty, p->no_repeat_ngram_size,
                                      p->encoder_no_repeat_ngram_size, rows, st));
            CHK(launch_topk_rows(h->logits, h->vocab_pad, V, n_keep, h->cand_val, h->cand_idx, h->top_val, h->top_idx, rows, st));
        }
        // ---- the scorer + the reorder ----
        BeamSelArgs a{};
        a.n = n; a.B = B; a.n_keep = n_keep; a.V = V; a.step = step; a.rows_per = rows_per; a.max_ids = h->max_ids; a.max_enc_ids = h->max_enc_ids;
        a.n_eos = c.n_eos; a.pad_tok = c.n_eos ? c.eos_ids[0] : 0;
        for (int e = 0; e < 8; ++e) a.eos[e] = e < c.n_eos ? c.eos_ids[e] : -1;
        a.top_val = h->top_val; a.top_idx = h->top_idx; a.st = h->bst_dev; a.powtab = h->bpow_dev;
        a.seq_in = ids_pool; a.seq_in_rows_per = step == 0 ? 1 : B; a.seq_out = h->bseq[(step + 1) & 1];
        a.ids = bd.ids; a.row_pos = bd.row_pos; a.views = bd.views; a.samp = bd.samp;
        a.ops1 = h->bops_dev[0]; a.ops2 = h->bops_dev[1]; a.op_counts = h->bop_counts_dev + 2 * step;
        a.reorder = h->beam_one_copy ? h->breorder_dev : nullptr;
        a.stream_stride = h->llm_stream_stride; a.tbuf_stride = h->tbuf_stride; a.max_beams = h->max_beams; a.nbuf = h->nbuf;
        a.log_val = slot_val(step); a.log_idx = slot_idx(step); a.log_dec = slot_dec(step); a.log_done = slot_done(step);
        a.log_seq = const_cast<int*>(seq_word); a.seq_value = ++h->bsel_seq; a.ticket = h->bticket_dev;
        a.err_word = const_cast<const int*>(ferr);
        a.force_tok = h->bforce_dev; a.force_par = h->bforce_dev + (size_t)c.max_new_tokens * h->max_beams; a.force_steps = force_steps;
        a.logits = h->logits; a.ld_logits = h->vocab_pad;
        if (h->beam_lean_tail) a.view = h->bview;
        CHK(launch_beam_select(a, st));
        if (step > 0) {  // (step 0: every beam descends from beam 0 and no tail exists yet -- nothing to copy)
            bf16_t* kr = h->rot_keys ? h->llm_kr : nullptr;
            if (a.reorder) {  // hypothesis tails + the reorder in ONE launch, no temporaries (beam.hip kv_beam_reorder_kernel)
                CHK(launch_kv_beam_reorder(h->llm_k, h->llm_v, kr, h->tbuf_k, h->tbuf_v, h->tbuf_kr, a.reorder, n, B, h->llm_stream_stride, h->tbuf_stride, h->adims, c.llm_layers, h->tcap, st));
            } else {
                CHK(launch_kv_positions_copy_list(h->llm_k, h->llm_v, kr, h->tbuf_k, h->tbuf_v, h->tbuf_kr, a.ops1, a.op_counts, 2 * nr, h->adims, c.llm_layers, h->tcap, st));
                CHK(launch_kv_positions_copy_list(h->llm_k, h->llm_v, kr, h->tbuf_k, h->tbuf_v, h->tbuf_kr, a.ops2, a.op_counts + 1, nr, h->adims, c.llm_layers, h->tcap, st));
            }
        }
        bt_lap(bt_tail);
        // ---- the step's sequence number (published by the scorer while the copies still run) ----
        for (unsigned long spins = 1; *seq_word != a.seq_value; ++spins) {
            if ((spins & 0x3FFFF) == 0) {  // every ~quarter million polls: has the stream drained without the number arriving?
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {
                    if (*seq_word == a.seq_value) break;
                    return h->fail(ISST_ERR_HIP, "beam step %d did not publish its choices", step);
                }
                if (q != hipErrorNotReady) return h->fail(ISST_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
            }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        bt_lap(bt_wait);
        h->fuse_ao_used = false;
        h->kv_ops_used = 0;  // (every copy batch enqueued before this step's scorer has run)
        bool all_done = true, upstream_failed = false;
        const int* dn = slot_done(step);
        for (int i = 0; i < n; ++i) { all_done = all_done && dn[i] >= 0 && (dn[i] & 1); upstream_failed = upstream_failed || dn[i] == -1; }  // (status >= 0: bit 0 = done, above it the bookkeeping digest)
        if (upstream_failed || *ferr != 0) {
            // a fused attention + o_proj launch of the pass behind these logits gave up waiting: the scorer left its state alone.  Latch the three-launch
            // path and run the same pass and the same step again (the KV appends and the hidden state are rewritten from the token ids)
            if (relaunched || step == 0 || !upstream_failed) return h->fail(ISST_ERR_HIP, "the fused attention + o_proj launch timed out and the pass could not be re-issued");
            latch_three_launch_path(h);
            relaunched = true;
            beam_pass_rows(h, bh, n, B, shared, stream_ids, total0, rows_len, step - 1);
            CHK(llm_forward(h, bd, nr, nr, shared ? n : nr, shared ? B : 1, false, nullptr, st, &bh, 0, 0, shared ? B : 0));
            continue;
        }
        relaunched = false;
        ++step;
        if (all_done || step >= p->max_new_tokens) {  // :920
            if (const int rc = follow(step - 1)) return rc;
            break;
        }
        // ---- the next forward pass (its rows' ids / positions are on the device already; the host needs the positions for the one-group launch forms) ----
        beam_pass_rows(h, bh, n, B, shared, stream_ids, total0, rows_len, step - 1);
        CHK(llm_forward(h, bd, nr, nr, shared ? n : nr, shared ? B : 1, false, nullptr, st, &bh, 0, 0, shared ? B : 0));
        bt_lap(bt_enq);
        if (const int rc = follow(step - 1)) return rc;
        bt_lap(bt_follow);
    }
    if (bt_on)
        std::fprintf(stderr, "[isst beam host] %d streams x %d beams, %d steps, scorer on the device: us per step -- tail + scorer + copies enqueue %.0f, wait for the step's number %.0f, forward enqueue %.0f, host follower (beside the GPU) %.0f\n",
                     n, B, step, bt_tail / step, bt_wait / step, bt_enq / step, bt_follow / step);
    return beam_finalize(h, p, n, B, stream_ids, prompt_lens, rows_len, total0, bs, out_ids, out_lens, mh, md, st, h->sync_at_end);
}

}  // namespace

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
        for (int g0 = g_first; g0 < n_groups; g0 += LLM_PREFILL_UNIT_GROUPS) {  // units: runs of <= 6 groups of this stream share their key tiles (llm_attn_prefill_kernel's consumer waves)
            mh.units[n_units].x = g0;
            mh.units[n_units].y = std::min(LLM_PREFILL_UNIT_GROUPS, n_groups - g0);
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
    // The prefill attention fills the rotated-key arena itself (LlmStreamView::rot_keys == 2: its loader waves rotate every cached tile they stage and store
    // it) wherever the prefill runs on llm_attn_prefill_kernel and only arena 0 of a stream is read through the arena below this chunk's tokens -- greedy, and
    // beam search in the shared-prefix form.  ISST_ROPE_FUSE=0 (or one beam group per arena) keeps the pre-pass over all layers.
    const bool fuse_rope = h->rot_keys && h->rope_fuse && gmax > 1 && (B == 1 || beams_share_prefix) && !h->rope_side;
    if (fuse_rope)
        for (int i = 0; i < n; ++i) mh.views[i].rot_keys = 2;
    if (fuse_rope) {
        HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
    } else if (h->rot_keys && any_cached && h->rope_side && st != nullptr) {
        // The host is far ahead of the GPU here (the encoder above is milliseconds of queued work), so the metadata upload and the pre-pass, issued on
        // the side stream now, run BESIDE the encoder; the prefill waits for both.  (Calls no longer end with the caller's stream drained: the event recorded on it
        // below orders the side stream behind everything the previous call left there, so the pre-pass cannot overtake an earlier writer of the caches.)
        if (!h->side) {
            int lo = 0, hi = 0;
            HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
            HIPCHK(hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo));
            HIPCHK(hipEventCreateWithFlags(&h->side_ev, hipEventDisableTiming));
        }
        if (!h->side_ev2) HIPCHK(hipEventCreateWithFlags(&h->side_ev2, hipEventDisableTiming));
        HIPCHK(hipEventRecord(h->side_ev2, st));  // (behind whatever the previous call left on the caller's stream: calls no longer end drained)
        HIPCHK(hipStreamWaitEvent(h->side, h->side_ev2, 0));
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
    // (the views keep rot_keys == 2 for the decode passes: every kernel but the prefill one reads "not 0" as "the arena is valid" -- and the pinned block must
    //  not be touched here anyway: its upload above executes when the stream gets there, not when it was enqueued)
    if (B > 1) {
        if (h->beam_device && !p->do_sample)
            return beam_decode_device(h, p, n, stream_ids, prompt_ids, prompt_lens, rows_len, prev_target_ids, n_prev, total0, out_ids, out_lens, mh, md, st);
        return beam_decode(h, p, n, stream_ids, prompt_ids, prompt_lens, rows_len, prev_target_ids, n_prev, total0, out_ids, out_lens, mh, md, st);
    }

    // ---- 3. greedy loop (patch_hf.py:606-624 -> HF _sample) ----
    std::vector<int> active(n);
    for (int i = 0; i < n; ++i) active[i] = i;

    // One stream (BASELINE.json configs[1]): the fused sampling tail also prepares the next pass on the device -- the sampled id appended to the id list the
    // processors read, its embedding row copied to the decoder's input row -- so a decode step is neither preceded by a metadata upload (a 3 us copy kernel
    // with a 13 us dependency gap behind it) nor opened by the embedding launch: 31 + 3 + 13 + 5 + 6 us between the last kernel of a pass and the first
    // projection of the next became the token poll + one launch (profiles/r04/trace_busy_prof1.txt).  The attention metadata of a one-group launch travels
    // in the kernel arguments already.  (Forced tokens -- a test aid -- replace the sampled id on the host, the sample branch draws on the host, a captured
    // graph replays frozen arguments: those keep the upload.)
    const bool advance_on_device = n == 1 && h->fused_sample && h->tail_advance && !any_forced && !p->do_sample && !h->use_graphs && !c.debug_taps && !h->prof_on;
    const SampleAdvance adv{advance_on_device ? 1 : 0, md.ids, md.ids_pool, md.samp, h->embed, h->lx, c.llm_dim};
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
                                    h->tok_host + h->tok_cap, na, st, &adv));
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
    auto decode_step = [&](int nr, bool restore = false) -> int {
        const bool graph_ok = h->use_graphs && st != nullptr && !logits_out && !c.debug_taps && !h->prof_on && !p->do_sample;  // (the NULL stream cannot be captured)
        if (!graph_ok) {
            // (restore: the pass is being re-issued after a failed one whose tail advanced the device's copies on garbage -- the host's block is the truth)
            const bool on_device = advance_on_device && !restore;
            if (!on_device) HIPCHK(hipMemcpyAsync(h->meta_dev + 4096, h->meta_host + 4096, mh.step_bytes - 4096, hipMemcpyHostToDevice, st));
            CHK(llm_forward(h, md, nr, nr, nr, 1, false, nullptr, st, &mh, 0, 0, 0, on_device));
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
    int last_nr = 0;          // rows of the decode pass whose tokens are being waited for (0: the prefill, which never takes the fused launch)
    bool relaunched = false;  // ... and that pass is already the re-issue on the three-launch path
    std::chrono::steady_clock::time_point ht_a, ht_b;
    if (g_ht.on) { ht_a = std::chrono::steady_clock::now(); g_ht.enq_first = HostTrace::us(g_ht.t_entry, ht_a); }
    while (true) {
        const int na = (int)active.size();
        if (g_ht.on) ht_a = std::chrono::steady_clock::now();
        CHK(wait_tokens());
        if (h->fuse_ao_used) {  // the fused attention + o_proj launches of the pass behind these tokens have run: did one of them give up waiting for its own workgroups?
            h->fuse_ao_used = false;
            if (*reinterpret_cast<volatile int*>(h->tok_host + h->tok_cap + 8) != 0) {
                // nothing of the step is committed yet (the tokens above are discarded, the stream lengths move at the end of the call): latch the three-launch
                // path and run the SAME pass again -- its metadata is still in the pinned block, its KV appends and hidden state are rewritten from the token ids
                if (relaunched || last_nr == 0) return h->fail(ISST_ERR_HIP, "the fused attention + o_proj launch timed out and the pass could not be re-issued on the three-launch path");
                latch_three_launch_path(h);
                relaunched = true;
                if (const int rc = decode_step(last_nr, true)) return rc;
                continue;
            }
        }
        relaunched = false;
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
        last_nr = nr;
        if (const int rc = decode_step(nr)) return rc;  // (the failing call has already recorded its message)
        if (g_ht.on) g_ht.enq_pass += HostTrace::us(ht_a, std::chrono::steady_clock::now());
    }
    // (With the fused tail the last wait above returned on the published tokens, a few microseconds before the kernel's own completion -- every write of that
    //  kernel to device state precedes the publication.  Nothing on the host depends on the stream being idle: the eviction that follows is a ring-start
    //  advance in host state, the next call's work is ordered behind on the stream, the pinned staging blocks were consumed passes ago, and the entry points
    //  that read device memory directly (debug reads, imports) synchronise the device themselves.  So the call returns without draining the stream -- the
    //  completion round trip of hipStreamSynchronize was GPU-idle time between two chunks.  ISST_SYNC_AT_END=1 restores the drain; the side-stream
    //  pre-pass (ISST_ROPE_SIDE, off by default) orders itself behind the caller's stream with an event.)
    if (tail_fused && h->sync_at_end) HIPCHK(hipStreamSynchronize(st));
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
