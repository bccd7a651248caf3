// Host side of libinfinisst_hip.so, part 2 of 4 (engine_internal.h): the speech side of one chunk -- conv feature extractor, streaming wav2vec2 encoder
// (exact window, no 2x recompute), length shrink, projector.  Reference: model/speech_encoder.py:219-236, model/patches/patch_speech_encoder.py:228-933.
#include "engine_internal.h"

namespace isst_impl {

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
                                     c.enc_heads, h->enc_cap, c.max_cache_size, bs, st, h->enc_cs_valid ? h->enc_cs : nullptr));
        } else {
            for (int i = 0; i < n; ++i) {
                bf16_t* kb = h->enc_k + (size_t)sids[i] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
                bf16_t* vb = h->enc_v + (size_t)sids[i] * h->enc_stream_stride + (size_t)l * h->enc_layer_stride;
                CHK(launch_enc_attention(h->eqkv + (size_t)i * Q * 3 * D, kb, vb, 0, ev + i, h->enc_cos, h->enc_sin, c.enc_rope_round_each,
                                         h->eattn + (size_t)i * Q * D, 1, Q, c.enc_heads, h->enc_cap, c.max_cache_size, bs, st, h->enc_cs_valid ? h->enc_cs : nullptr));
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

}  // namespace isst_impl

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

