// Device pieces of beam search (reference model/patches/patch_hf.py:687-967, the production decoding mode):
//   * the beams of a stream are B lock-stepped KV arenas that differ only in the positions written during the current
//     chunk; "reordering the cache" (:910-913, a full index_select per step in the reference) and the per-hypothesis KV
//     copies (:113-120, :193-200) become copies of a few POSITIONS between arenas and small buffers;
//   * log_softmax over the vocabulary in fp32 (:833-837), then the logits processors act on log-probs (:839);
//   * top-k over beams x vocab (:878): per-row top-k on the device, the host merges the B rows.
#include "common.h"
#include "kernels.h"

#define HD 128

// ---- copy `count` logical positions starting at p0 between an arena (K [slots][128], V [slots][128]) and a buffer
//      (K [layers][kv][tcap][128], V [layers][kv][tcap][128], both row-major).  grid = (kv_heads, layers * n_ops): one workgroup moves all positions of
//      its (kv head, layer, op) as 16-byte chunks, the three pools' loads in flight together.  (One 128-thread workgroup per (position, kv head, layer, op)
//      moving 2 bytes per thread was 414 k workgroups per launch at 64 streams x 4 beams: 152 us per launch, 1.2 TB/s -- profiles/r04/trace_busy_prof64x4.txt.)
__global__ __launch_bounds__(256) void kv_positions_copy_kernel(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf,
                                                               const KvCopyOp* __restrict__ ops, LlmAttnDims d, int layers, int tcap) {
    const int kvh = blockIdx.x;
    const int layer = blockIdx.y % layers;
    const KvCopyOp op = ops[blockIdx.y / layers];
    const int slots = d.sys_cap + d.ring_cap;
    const long abase = op.arena_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;
    const long bbase = op.buf_offset + ((long)layer * d.kv_heads + kvh) * tcap * HD;
    // (krpool: the keys rotated at their position of this chunk, LlmStreamView::rot_keys -- travels with K so that the beams' arenas
    //  can be read through it like a greedy stream's)
    for (int e = threadIdx.x; e < op.count * (HD / 8); e += 256) {
        const int t = e / (HD / 8), ch = e % (HD / 8);
        const int p = op.p0 + t;
        long slot;
        if (p < op.sys_len) slot = p;
        else { int x = op.ring_start + (p - op.sys_len); x %= d.ring_cap; slot = (long)d.sys_cap + x; }
        const long ai = abase + slot * HD + ch * 8, bi = bbase + (long)t * HD + ch * 8;
        if (op.to_arena) {
            const u32x4_t k = *reinterpret_cast<const u32x4_t*>(kbuf + bi), v = *reinterpret_cast<const u32x4_t*>(vbuf + bi);
            u32x4_t kr = {0u, 0u, 0u, 0u};
            if (krpool) kr = *reinterpret_cast<const u32x4_t*>(krbuf + bi);
            *reinterpret_cast<u32x4_t*>(kpool + ai) = k;
            *reinterpret_cast<u32x4_t*>(vtpool + ai) = v;
            if (krpool) *reinterpret_cast<u32x4_t*>(krpool + ai) = kr;
        } else {
            const u32x4_t k = *reinterpret_cast<const u32x4_t*>(kpool + ai), v = *reinterpret_cast<const u32x4_t*>(vtpool + ai);
            u32x4_t kr = {0u, 0u, 0u, 0u};
            if (krpool) kr = *reinterpret_cast<const u32x4_t*>(krpool + ai);
            *reinterpret_cast<u32x4_t*>(kbuf + bi) = k;
            *reinterpret_cast<u32x4_t*>(vbuf + bi) = v;
            if (krpool) *reinterpret_cast<u32x4_t*>(krbuf + bi) = kr;
        }
    }
}

int launch_kv_positions_copy(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const KvCopyOp* ops, int n_ops, int max_count,
                             LlmAttnDims d, int layers, int tcap, hipStream_t s) {
    if (n_ops <= 0 || max_count <= 0) return ISST_OK;
    hipLaunchKernelGGL(kv_positions_copy_kernel, dim3(d.kv_heads, layers * n_ops), dim3(256), 0, s, kpool, vtpool, krpool, kbuf, vbuf, krbuf, ops, d, layers, tcap);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ---- the same copies from an op list written on the device (beam_select_kernel): the host knows only an upper bound of the op count, so a workgroup
//      (kv head, layer, lane z) walks ops z, z + Z, ... of the *count listed ----
__global__ __launch_bounds__(256) void kv_positions_copy_list_kernel(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf,
                                                                    const KvCopyOp* __restrict__ ops, const int* __restrict__ count, LlmAttnDims d, int tcap) {
    const int kvh = blockIdx.x, layer = blockIdx.y;
    const int n_ops = *count;
    const int slots = d.sys_cap + d.ring_cap;
    for (int o = blockIdx.z; o < n_ops; o += gridDim.z) {
        const KvCopyOp op = ops[o];
        const long abase = op.arena_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;
        const long bbase = op.buf_offset + ((long)layer * d.kv_heads + kvh) * tcap * HD;
        for (int e = threadIdx.x; e < op.count * (HD / 8); e += 256) {
            const int t = e / (HD / 8), ch = e % (HD / 8);
            const int p = op.p0 + t;
            long slot;
            if (p < op.sys_len) slot = p;
            else { int x = op.ring_start + (p - op.sys_len); x %= d.ring_cap; slot = (long)d.sys_cap + x; }
            const long ai = abase + slot * HD + ch * 8, bi = bbase + (long)t * HD + ch * 8;
            if (op.to_arena) {
                const u32x4_t k = *reinterpret_cast<const u32x4_t*>(kbuf + bi), v = *reinterpret_cast<const u32x4_t*>(vbuf + bi);
                u32x4_t kr = {0u, 0u, 0u, 0u};
                if (krpool) kr = *reinterpret_cast<const u32x4_t*>(krbuf + bi);
                *reinterpret_cast<u32x4_t*>(kpool + ai) = k;
                *reinterpret_cast<u32x4_t*>(vtpool + ai) = v;
                if (krpool) *reinterpret_cast<u32x4_t*>(krpool + ai) = kr;
            } else {
                const u32x4_t k = *reinterpret_cast<const u32x4_t*>(kpool + ai), v = *reinterpret_cast<const u32x4_t*>(vtpool + ai);
                u32x4_t kr = {0u, 0u, 0u, 0u};
                if (krpool) kr = *reinterpret_cast<const u32x4_t*>(krpool + ai);
                *reinterpret_cast<u32x4_t*>(kbuf + bi) = k;
                *reinterpret_cast<u32x4_t*>(vbuf + bi) = v;
                if (krpool) *reinterpret_cast<u32x4_t*>(krbuf + bi) = kr;
            }
        }
    }
}
int launch_kv_positions_copy_list(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const KvCopyOp* ops, const int* count,
                                  int max_ops, LlmAttnDims d, int layers, int tcap, hipStream_t s) {
    if (max_ops <= 0) return ISST_OK;
    // op lanes: enough workgroups to fill the chip (kv heads x layers x Z >= ~2048), never more lanes than ops
    int z = 2048 / (d.kv_heads * layers) + 1;
    z = z > max_ops ? max_ops : z;
    hipLaunchKernelGGL(kv_positions_copy_list_kernel, dim3(d.kv_heads, layers, z), dim3(256), 0, s, kpool, vtpool, krpool, kbuf, vbuf, krbuf, ops, count, d, tcap);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ---- every copy of a beam step in ONE launch (kernels.h BeamReorder).  grid (kv heads, layers, streams) ----
template <int NB>
__global__ __launch_bounds__(128) void kv_beam_reorder_kernel(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf,
                                                             const BeamReorder* __restrict__ recs, int B, long stream_stride, long tbuf_stride, LlmAttnDims d, int tcap) {
    const BeamReorder& rec = recs[blockIdx.z];
    const int n_move = rec.n_move, n_save = rec.n_save;
    if (n_move == 0 && n_save == 0) return;
    const int kvh = blockIdx.x, layer = blockIdx.y;
    const int slots = d.sys_cap + d.ring_cap;
    const long abase = rec.arena0 + (long)layer * d.layer_stride + (long)kvh * slots * HD;
    const long bbase = rec.buf0 + ((long)layer * d.kv_heads + kvh) * tcap * HD;
    int par[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) par[b] = b < B ? rec.par[b] : b;
    for (int e = threadIdx.x; e < rec.count * (HD / 8); e += 128) {
        const int t = e / (HD / 8), ch = e % (HD / 8);
        const int p = rec.p0 + t;
        long slot;
        if (p < rec.sys_len) slot = p;
        else { int x = rec.ring_start + (p - rec.sys_len); x %= d.ring_cap; slot = (long)d.sys_cap + x; }
        const long ai = abase + slot * HD + ch * 8, bi = bbase + (long)t * HD + ch * 8;
        // hypothesis tails: (old) beam -> buffer.  Their loads are issued before any arena is written below
        for (int k = 0; k < n_save; ++k) {
            const long src = ai + (long)rec.save_beam[k] * stream_stride, dst = bi + (long)rec.save_buf[k] * tbuf_stride;
            const u32x4_t kk = *reinterpret_cast<const u32x4_t*>(kpool + src), vv = *reinterpret_cast<const u32x4_t*>(vtpool + src);
            u32x4_t rr = {0u, 0u, 0u, 0u};
            if (krpool) rr = *reinterpret_cast<const u32x4_t*>(krpool + src);
            *reinterpret_cast<u32x4_t*>(kbuf + dst) = kk;
            *reinterpret_cast<u32x4_t*>(vbuf + dst) = vv;
            if (krpool) *reinterpret_cast<u32x4_t*>(krbuf + dst) = rr;
        }
        // the reorder: every moving beam's chunk comes from its parent's arena ...
        u32x4_t mk[NB], mv[NB], mr[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            mk[b] = mv[b] = mr[b] = (u32x4_t){0u, 0u, 0u, 0u};
            if (par[b] != b) {  // (wave-uniform)
                const long src = ai + (long)par[b] * stream_stride;
                mk[b] = *reinterpret_cast<const u32x4_t*>(kpool + src);
                mv[b] = *reinterpret_cast<const u32x4_t*>(vtpool + src);
                if (krpool) mr[b] = *reinterpret_cast<const u32x4_t*>(krpool + src);
            }
        }
        // ... and no arena is written before every load above -- of this thread, the only one that touches this chunk -- has returned
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (par[b] != b) {
                const long dst = ai + (long)b * stream_stride;
                *reinterpret_cast<u32x4_t*>(kpool + dst) = mk[b];
                *reinterpret_cast<u32x4_t*>(vtpool + dst) = mv[b];
                if (krpool) *reinterpret_cast<u32x4_t*>(krpool + dst) = mr[b];
            }
        }
    }
}
int launch_kv_beam_reorder(bf16_t* kpool, bf16_t* vtpool, bf16_t* krpool, bf16_t* kbuf, bf16_t* vbuf, bf16_t* krbuf, const BeamReorder* recs, int n_streams, int B,
                           long stream_stride, long tbuf_stride, LlmAttnDims d, int layers, int tcap, hipStream_t s) {
    if (n_streams <= 0) return ISST_OK;
    if (B < 1 || B > BEAM_MAX_B) return ISST_ERR_ARG;
    const dim3 grid(d.kv_heads, layers, n_streams);
    if (B <= 4) hipLaunchKernelGGL(kv_beam_reorder_kernel<4>, grid, dim3(128), 0, s, kpool, vtpool, krpool, kbuf, vbuf, krbuf, recs, B, stream_stride, tbuf_stride, d, tcap);
    else hipLaunchKernelGGL(kv_beam_reorder_kernel<BEAM_MAX_B>, grid, dim3(128), 0, s, kpool, vtpool, krpool, kbuf, vbuf, krbuf, recs, B, stream_stride, tbuf_stride, d, tcap);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

__device__ __forceinline__ float beam_score_of(float x, int v, float lz, const int* __restrict__ side_tok, const float* __restrict__ side_val, int n_side);  // (below, with beam_process_kernel)
__device__ __noinline__ float beam_score_side(int v, const int* __restrict__ side_tok, const float* __restrict__ side_val, int n_side);

// ---- the beam scorer of one step, on the device: one wave per stream.  It restates engine_llm.hip's host scorer (the `follower` there, which is pinned to
//      the reference's beam_search_process / BeamHypotheses.add through beam_scorer.npz / beam_loop.npz) operation for operation -- fp32 `candidate + beam
//      score`, the (value desc, flat index asc) order, double `sum / len^penalty` with len^penalty from a HOST-filled table -- so that both reach the same
//      decisions bit for bit; the host re-derives every step from the logged candidates and fails the call if the two ever differ.  What it buys: the next
//      forward pass's token ids, the reordered tails and the sequences the processors read exist on the device the moment the candidates do; the host
//      scorer (172 us of GPU-idle time per step at 64 streams x 4 beams, 40 + 37 us at one stream: profiles/r04/trace_busy_prof64x4.txt, _beam4.txt) is off
//      the critical path. ----
__global__ __launch_bounds__(64) void beam_select_kernel(BeamSelArgs a) {
    __shared__ float cval[BEAM_MAX_B * BEAM_TOPK];
    __shared__ int cflat[BEAM_MAX_B * BEAM_TOPK];
    __shared__ float sval[BEAM_TOPK];
    __shared__ int sflat[BEAM_TOPK];
    __shared__ BeamDevStream S;
    __shared__ int ntok[BEAM_MAX_B], npar[BEAM_MAX_B];
    __shared__ float nscore[BEAM_MAX_B], nflp[BEAM_MAX_B];
    __shared__ KvCopyOp l1[2 * BEAM_MAX_B], l2[BEAM_MAX_B];
    __shared__ BeamReorder R;
    __shared__ int n1, n2, base1, base2, status;
    const int i = blockIdx.x, lane = threadIdx.x, B = a.B;
    BeamDevStream* gS = a.st + i;
    for (int w = lane; w < (int)(sizeof(BeamDevStream) / 4); w += 64) reinterpret_cast<int*>(&S)[w] = reinterpret_cast<const int*>(gS)[w];
    if (lane == 0) { n1 = 0; n2 = 0; status = 0; R.n_save = 0; R.n_move = 0; R.count = 0; }
    __syncthreads();
    const bool upstream_failed = a.err_word && *reinterpret_cast<const volatile int*>(a.err_word) != 0;
    const int len = S.prompt_len + a.step;  // tokens of every beam's sequence before this step
    auto op = [&](int beam, int buf, int count, int to_arena) {
        KvCopyOp o;
        o.arena_offset = ((long)S.sid * a.max_beams + beam) * a.stream_stride;
        o.buf_offset = ((long)S.sid * a.nbuf + buf) * a.tbuf_stride;
        o.p0 = S.P0; o.count = count; o.sys_len = S.sys_len; o.ring_start = S.ring_start; o.to_arena = to_arena; o.pad = 0;
        return o;
    };
    if (upstream_failed) {
        if (lane == 0) status = -1;  // the state stays as it is: the host re-issues the pass on the three-launch path and this step runs again
    } else if (S.done) {
        // a finished batch entry (patch_hf.py:83-92): pad tokens, zero scores, hypotheses untouched; its rows still ride through the forward pass
        if (lane < B) { ntok[lane] = a.pad_tok; npar[lane] = lane; nscore[lane] = 0.f; nflp[lane] = 0.f; }
        if (lane == 0) status = 1;
    } else {
        // ---- candidates: processed log-prob + beam score (fp32 add), flat index = beam * V + token; invalid slots are left out ----
        const int N = a.rows_per * a.n_keep;
        for (int c = lane; c < N; c += 64) {
            const int b = c / a.n_keep, j = c - b * a.n_keep;
            const int r = i * a.rows_per + b;
            const int idx = a.top_idx[r * BEAM_TOPK + j];
            const bool valid = idx >= 0 && idx < a.V;
            cval[c] = valid ? a.top_val[r * BEAM_TOPK + j] + S.score[b] : 0.f;
            cflat[c] = valid ? b * a.V + idx : -1;
        }
        __syncthreads();
        // ---- rank = number of valid candidates that come first under (value desc, flat asc): a strict total order (flat indices are distinct) ----
        int valid_total = 0;
        for (int c = lane; c < N; c += 64) {
            const int fc = cflat[c];
            if (fc < 0) continue;
            const float vc = cval[c];
            int rank = 0;
            for (int q = 0; q < N; ++q) {
                const int fq = cflat[q];
                rank += (fq >= 0 && (cval[q] > vc || (cval[q] == vc && fq < fc))) ? 1 : 0;
            }
            if (rank < a.n_keep && rank < BEAM_TOPK) { sval[rank] = vc; sflat[rank] = fc; }
        }
        for (int c = lane; c < N; c += 64) valid_total += cflat[c] >= 0 ? 1 : 0;
        valid_total = (int)wave_sum((float)valid_total);  // (<= 256: exact in fp32)
        __syncthreads();
        if (lane == 0) {
            const int ns = valid_total < a.n_keep ? valid_total : a.n_keep;
            const int gen_len = a.step + 1;  // cur_len - prompt_len: the hypothesis length an EOS candidate of this step closes
            int chosen = 0, st = 0;
            for (int rank = 0; rank < ns && chosen < B; ++rank) {
                const int b = sflat[rank] / a.V, tok = sflat[rank] - b * a.V;
                bool is_eos = false;
                for (int e = 0; e < a.n_eos; ++e) is_eos = is_eos || tok == a.eos[e];
                if (is_eos) {
                    if (rank >= B) continue;
                    int buf = -1;
                    if (a.step > 0) {  // the hypothesis keeps a copy of that beam's tail (the reference clones the whole KV cache, :113-120)
                        if (S.n_free == 0) { st = -2; break; }
                        buf = S.free_bufs[--S.n_free];
                        if (a.reorder) { R.save_beam[R.n_save] = b; R.save_buf[R.n_save] = buf; ++R.n_save; }
                        else l1[n1++] = op(b, buf, a.step, 0);
                    }
                    // BeamHypotheses.add (:278-302)
                    const double hs = (double)sval[rank] / a.powtab[gen_len];
                    int freed = -1;
                    if (S.hyp_n < B || hs > S.worst) {
                        S.hyp_score[S.hyp_n] = hs; S.hyp_buf[S.hyp_n] = buf; ++S.hyp_n;
                        if (S.hyp_n > B) {  // drop the worst: sorted((score, index))[0]; worst_score = the next one's
                            int w0 = 0;
                            for (int q = 1; q < S.hyp_n; ++q) if (S.hyp_score[q] < S.hyp_score[w0]) w0 = q;
                            freed = S.hyp_buf[w0];
                            for (int q = w0; q + 1 < S.hyp_n; ++q) { S.hyp_score[q] = S.hyp_score[q + 1]; S.hyp_buf[q] = S.hyp_buf[q + 1]; }
                            --S.hyp_n;
                            int w1 = 0;
                            for (int q = 1; q < S.hyp_n; ++q) if (S.hyp_score[q] < S.hyp_score[w1]) w1 = q;
                            S.worst = S.hyp_score[w1];
                        } else {
                            S.worst = hs < S.worst ? hs : S.worst;
                        }
                    } else {
                        freed = buf;
                    }
                    if (freed >= B) S.free_bufs[S.n_free++] = freed;
                } else {
                    nscore[chosen] = sval[rank]; ntok[chosen] = tok; npar[chosen] = b; nflp[chosen] = 0.f;
                    ++chosen;
                }
            }
            if (st == 0 && chosen < B) st = -3;
            if (st == 0 && ns > 0 && !S.done && S.hyp_n >= B) {  // BeamHypotheses.is_done, early_stopping = False [3P]
                const double highest = (double)sval[0] / a.powtab[gen_len];
                if (S.worst >= highest) S.done = 1;
            }
            if (st == 0 && i == 0 && a.step < a.force_steps) {  // teacher forcing (test aid): the caller's (token, parent) choices; score = parent's + the token's processed log-prob
                for (int b = 0; b < B; ++b) {
                    const int tok = a.force_tok[a.step * B + b], par = a.force_par[a.step * B + b];
                    if (tok < 0 || tok >= a.V || par < 0 || par >= a.rows_per) { st = -4; break; }
                    const int fr_ = i * a.rows_per + par;
                    const float raw = a.logits[(long)fr_ * a.ld_logits + tok];
                    const float lp = a.view.logz ? beam_score_of(raw, tok, a.view.logz[fr_], a.view.side_tok + (long)fr_ * a.view.side_cap, a.view.side_val + (long)fr_ * a.view.side_cap,
                                                                 a.view.side_n[fr_]) : raw;
                    ntok[b] = tok; npar[b] = par; nscore[b] = S.score[par] + lp; nflp[b] = lp;
                }
            }
            if (st == 0 && a.step > 0 && a.reorder) {  // reorder the tails (:910-913) as one record: kv_beam_reorder_kernel moves them in one launch
                for (int b = 0; b < B; ++b) { R.par[b] = npar[b]; R.n_move += npar[b] != b ? 1 : 0; }
                for (int b = B; b < BEAM_MAX_B; ++b) R.par[b] = b;
                R.arena0 = (long)S.sid * a.max_beams * a.stream_stride;
                R.buf0 = (long)S.sid * a.nbuf * a.tbuf_stride;
                R.p0 = S.P0; R.count = a.step; R.sys_len = S.sys_len; R.ring_start = S.ring_start;
            } else if (st == 0 && a.step > 0) {  // ... or as two op lists: new beam b continues parent npar[b] -- parents staged through temporaries 0..B-1
                unsigned need = 0;
                for (int b = 0; b < B; ++b) if (npar[b] != b) need |= 1u << npar[b];
                for (int src = 0; src < B; ++src) if (need >> src & 1u) l1[n1++] = op(src, src, a.step, 0);
                for (int b = 0; b < B; ++b) if (npar[b] != b) l2[n2++] = op(b, npar[b], a.step, 1);
            }
            if (st != 0) { n1 = 0; n2 = 0; R.n_save = 0; R.n_move = 0; }
            status = st != 0 ? st : (S.done ? 1 : 0);
            if (st == 0) for (int b = 0; b < B; ++b) S.score[b] = nscore[b];
        }
    }
    __syncthreads();
    const int st = status;
    if (st >= 0) {
        // ---- sequences (the processors' ids of the next step): row i * B + b = parent's tokens + the chosen one ----
        for (int b = 0; b < B; ++b) {
            const int* src = a.seq_in + (long)(i * a.seq_in_rows_per + (a.seq_in_rows_per == 1 ? 0 : npar[b])) * a.max_ids;
            int* dst = a.seq_out + (long)(i * B + b) * a.max_ids;
            for (int t = lane; t < len; t += 64) dst[t] = src[t];
            if (lane == 0) dst[len] = ntok[b];
        }
        // ---- the next forward pass's rows ----
        if (lane < B) {
            const int r = i * B + lane;
            a.ids[r] = ntok[lane];
            a.row_pos[r] = S.P0 + a.step;
            a.views[r].new_start = S.P0 + a.step;
            a.samp[r].n_ids = len + 1;
        }
        // ---- position copies: reserve this stream's share of the two lists ----
        if (lane == 0) {
            base1 = n1 ? atomicAdd(a.op_counts, n1) : 0;
            base2 = n2 ? atomicAdd(a.op_counts + 1, n2) : 0;
        }
        __syncthreads();
        for (int q = lane; q < n1; q += 64) a.ops1[base1 + q] = l1[q];
        for (int q = lane; q < n2; q += 64) a.ops2[base2 + q] = l2[q];
        // ---- state back to memory ----
        for (int w = lane; w < (int)(sizeof(BeamDevStream) / 4); w += 64) reinterpret_cast<int*>(gS)[w] = reinterpret_cast<const int*>(&S)[w];
    }
    // ---- the step's copy record: ALWAYS written (a failed or finished stream's says "move nothing": the copy launch must not find the previous step's) ----
    if (a.reorder)
        for (int w = lane; w < (int)(sizeof(BeamReorder) / 4); w += 64) reinterpret_cast<int*>(a.reorder + i)[w] = reinterpret_cast<const int*>(&R)[w];
    // ---- log to the host: candidates as the top-k left them, the choices, the stream's status ----
    for (int e = lane; e < a.rows_per * BEAM_TOPK; e += 64) {
        const int b = e / BEAM_TOPK, j = e - b * BEAM_TOPK;
        a.log_val[(long)(i * B + b) * BEAM_TOPK + j] = a.top_val[(long)(i * a.rows_per + b) * BEAM_TOPK + j];
        a.log_idx[(long)(i * B + b) * BEAM_TOPK + j] = a.top_idx[(long)(i * a.rows_per + b) * BEAM_TOPK + j];
    }
    if (lane < B && st >= 0) {
        BeamDecision dcs;
        dcs.tok = ntok[lane]; dcs.par = npar[lane]; dcs.score = nscore[lane]; dcs.forced_lp = nflp[lane];
        a.log_dec[i * B + lane] = dcs;
    }
    // the status word also carries a digest of the hypothesis bookkeeping: the host follower re-derives the step with its own scorer and compares
    // tokens / parents / scores, but a divergence in WHICH BUFFER a closed hypothesis's tail was saved to -- or in the free list's order -- would only show
    // when finalize copies a winner's tail out of the wrong buffer.  st >= 0: bit 0 = done, bits 1.. = beam_book_digest (kernels.h) of hyp_n, hyp_buf[], n_free, free_bufs[]
    if (lane == 0) a.log_done[i] = st < 0 ? st : (st | (int)(beam_book_digest(S.hyp_n, S.hyp_buf, S.n_free, S.free_bufs) << 1));
    __threadfence_system();  // every lane: its part of the log is in host memory before this stream's ticket is drawn
    __syncthreads();         // (the workgroup IS one wave -- launch_beam_select, static_assert above -- so this costs nothing; it states that lane 0's ticket follows every lane's fence)
    if (lane == 0) {
        bool last = true;  // (one stream: this wave is the launch -- no ticket, no second fence on the step's critical path)
        if (a.n > 1) {
            last = atomicAdd(a.ticket, 1) == a.n - 1;
            if (last) {
                *a.ticket = 0;  // re-armed (launch boundary = visibility)
                __threadfence_system();
            }
        }
        if (last) *reinterpret_cast<volatile int*>(a.log_seq) = a.seq_value;
    }
}
static_assert(WAVE == 64, "beam_select_kernel is ONE 64-lane wave per stream: wave_sum / wave_argmax_all are its block reductions and its log writes precede lane 0's ticket in program order");
int launch_beam_select(const BeamSelArgs& a, hipStream_t s) {
    if (a.n <= 0) return ISST_OK;
    if (a.B < 1 || a.B > BEAM_MAX_B || a.n_keep < 1 || a.n_keep > BEAM_TOPK || a.rows_per < 1 || a.rows_per > a.B || a.n_eos < 0 || a.n_eos > 8) return ISST_ERR_ARG;
    hipLaunchKernelGGL(beam_select_kernel, dim3(a.n), dim3(64), 0, s, a);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ---- log_softmax, stage 1: per (part, row) max and sum of exp.  One pass over memory: a thread's (at most LSE_PT) 16-byte pieces of the slice are all in
//      flight before the first use and stay in registers for the sum (the first form read the slice twice with 4-byte loads: 51 us at 256 rows, 2.5 TB/s).
//      Slices start at multiples of 4 elements (rows are 64-byte aligned, ld % 4 == 0 is checked by the launcher). ----
#define LSE_PARTS 64
#define LSE_PT 4
__device__ __forceinline__ int lse_per(int vocab) { return (((vocab + LSE_PARTS - 1) / LSE_PARTS) + 3) & ~3; }
__global__ __launch_bounds__(256) void lse_part_kernel(const float* __restrict__ logits, long ld, int vocab, float* __restrict__ pmax,
                                                       float* __restrict__ psum) {
    __shared__ float sm[4], ss[4];
    const float* L = logits + (long)blockIdx.y * ld;
    const int per = lse_per(vocab);
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    float m = -INFINITY, s = 0.f;
    for (int v0 = lo; v0 < hi; v0 += 1024 * LSE_PT) {  // (one trip for every vocabulary up to 64 x 4096 entries)
        f32x4_t q[LSE_PT];
#pragma unroll
        for (int u = 0; u < LSE_PT; ++u) {
            const int v4 = v0 + (u * 256 + (int)threadIdx.x) * 4;
            q[u] = *reinterpret_cast<const f32x4_t*>(L + (v4 < hi ? v4 : lo));  // UNCONDITIONAL load (a piece past the slice re-reads its first one and is masked below): a branch around a load makes hipcc drain the queue per load
#pragma unroll
            for (int e = 0; e < 4; ++e) q[u][e] = (v4 + e < hi) ? q[u][e] : -INFINITY;  // (the row's padding up to ld is readable; entries at or past `hi` do not count)
        }
        float m2 = m;
#pragma unroll
        for (int u = 0; u < LSE_PT; ++u) m2 = fmaxf(fmaxf(fmaxf(q[u][0], q[u][1]), fmaxf(q[u][2], q[u][3])), m2);
        if (m2 > -INFINITY) {
            s = s * expf(m - m2);  // (m == -inf: s is 0 and exp(-inf) = 0)
#pragma unroll
            for (int u = 0; u < LSE_PT; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += expf(q[u][e] - m2);
            m = m2;
        }
    }
    const float wm = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = wm;
    __syncthreads();
    const float M = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    s = (m > -INFINITY) ? s * expf(m - M) : 0.f;
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        pmax[blockIdx.y * LSE_PARTS + blockIdx.x] = M;
        psum[blockIdx.y * LSE_PARTS + blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
    }
}
// log Z of a row from its LSE_PARTS partial results: one wave, every lane ends up with the value
__device__ __forceinline__ float lse_logz(const float* __restrict__ pmax, const float* __restrict__ psum, int row, int lane) {
    const float m = pmax[row * LSE_PARTS + lane];
    const float M = wave_max(m);
    const float s = (m == -INFINITY) ? 0.f : psum[row * LSE_PARTS + lane] * expf(m - M);
    const float S = wave_sum(s);
    return M + logf(S);
}
// ---- stage 2: logits <- logits - logZ (in place) ----
__global__ __launch_bounds__(256) void lse_apply_kernel(float* __restrict__ logits, long ld, int vocab, const float* __restrict__ pmax,
                                                        const float* __restrict__ psum) {
    __shared__ float logz;
    if (threadIdx.x < 64) {
        const float z = lse_logz(pmax, psum, blockIdx.y, threadIdx.x);
        if (threadIdx.x == 0) logz = z;
    }
    __syncthreads();
    float* L = logits + (long)blockIdx.y * ld;
    const int per = lse_per(vocab);
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    for (int v = lo + threadIdx.x; v < hi; v += blockDim.x) L[v] -= logz;
}

// ---- the beam step's processors WITHOUT the in-place log-softmax sweep (beam_decode_device): one block per row.  The row keeps its RAW logits; what the
//      processors change is recorded so that a reader can reproduce `processors(log_softmax(row))` element by element, bit for bit:
//        * log Z of the row -> logz[row] (the same arithmetic as lse_apply_kernel's): an untouched entry is  raw - logz;
//        * repetition penalty (first occurrence of every distinct id): the penalised log-prob goes to the row's SIDE list (token, value) and the entry
//          becomes NaN -- "look me up" (logits of a bf16 path are never NaN; a reader that finds no side entry treats the NaN as -inf);
//        * n-gram bans, suppressed tokens: the entry becomes -inf (-inf - logz = -inf), over a NaN too: penalised AND banned is banned.
//      The order is the reference's (RepetitionPenalty -> NoRepeatNGram -> EncoderNoRepeatNGram -> SuppressTokens, sample.hip).  Two sweeps over the
//      rows x vocab fp32 scores of a step remain (lse_part, the top-k scan) instead of four; 256 rows x 128 263: 524 MB -> 262 MB. ----
__global__ __launch_bounds__(256) void beam_process_kernel(float* __restrict__ logits, long ld, const SampleStream* __restrict__ ss, const int* __restrict__ ids_pool,
                                                           const int* __restrict__ enc_pool, const int* __restrict__ suppress, int n_suppress, float pen, int ngram,
                                                           int enc_ngram, const float* __restrict__ pmax, const float* __restrict__ psum, float* __restrict__ logz_out,
                                                           int* __restrict__ side_tok, float* __restrict__ side_val, int* __restrict__ side_n, int side_cap) {
    __shared__ float s_logz;
    __shared__ int s_cnt;
    const SampleStream st = ss[blockIdx.x];
    const int row = st.logits_row;
    float* L = logits + (long)row * ld;
    const int* ids = ids_pool + st.ids_off;
    const int* enc = enc_pool + st.enc_off;
    const int n = st.n_ids, tid = threadIdx.x;
    if (tid < 64) {
        const float z = lse_logz(pmax, psum, row, tid);
        if (tid == 0) { s_logz = z; s_cnt = 0; logz_out[row] = z; }
    }
    __syncthreads();
    const float lz = s_logz;
    if (pen != 1.0f) {
        for (int i = tid; i < n; i += blockDim.x) {
            const int t = ids[i];
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && (ids[j] != t);
            if (first) {
                const float v = L[t] - lz;
                const int k = atomicAdd(&s_cnt, 1);
                if (k < side_cap) {
                    side_tok[(long)row * side_cap + k] = t;
                    side_val[(long)row * side_cap + k] = v < 0.f ? v * pen : v / pen;
                    L[t] = __int_as_float(0x7fc00000);
                } else {
                    L[t] = v < 0.f ? v * pen : v / pen;  // (cannot happen: the list holds one entry per id of the sequence) -- degrade to an in-place value
                }
            }
        }
    }
    __syncthreads();
    if (tid == 0) side_n[row] = s_cnt < side_cap ? s_cnt : side_cap;
    if (ngram > 0 && n + 1 >= ngram) {
        const int* key = ids + n - (ngram - 1);
        for (int p = tid; p + ngram <= n; p += blockDim.x) {
            bool eq = true;
            for (int q = 0; q < ngram - 1; ++q) eq = eq && (ids[p + q] == key[q]);
            if (eq) L[ids[p + ngram - 1]] = -INFINITY;
        }
    }
    if (enc_ngram > 0 && n + 1 >= enc_ngram) {
        const int* key = ids + n - (enc_ngram - 1);
        for (int p = tid; p + enc_ngram <= st.n_enc; p += blockDim.x) {
            bool eq = true;
            for (int q = 0; q < enc_ngram - 1; ++q) eq = eq && (enc[p + q] == key[q]);
            if (eq) L[enc[p + enc_ngram - 1]] = -INFINITY;
        }
    }
    for (int i = tid; i < n_suppress; i += blockDim.x) L[suppress[i]] = -INFINITY;
}
// the processed log-prob of entry v of a row prepared by beam_process_kernel (x = its stored value).  The NaN path -- a handful of entries per row -- is a CALL:
// inlined into the top-k kernels' unrolled element loops (16-64 copies of a search loop) it made them several times their size.
__device__ __noinline__ float beam_score_side(int v, const int* __restrict__ side_tok, const float* __restrict__ side_val, int n_side) {
    for (int k = 0; k < n_side; ++k)
        if (side_tok[k] == v) return side_val[k];
    return -INFINITY;
}
__device__ __forceinline__ float beam_score_of(float x, int v, float lz, const int* __restrict__ side_tok, const float* __restrict__ side_val, int n_side) {
    if (x == x) return x - lz;
    return beam_score_side(v, side_tok, side_val, n_side);
}
struct TopkView {  // how a top-k kernel reads a score row: as it stands (lz == null), or through beam_process_kernel's encoding
    const float* logz;
    const int* side_tok;
    const float* side_val;
    const int* side_n;
    int side_cap;
};
__device__ __forceinline__ float topk_view(const TopkView& tv, int row, float x, int v) {
    if (!tv.logz) return x;
    return beam_score_of(x, v, tv.logz[row], tv.side_tok + (long)row * tv.side_cap, tv.side_val + (long)row * tv.side_cap, tv.side_n[row]);
}
struct TopkRowView {  // one row's view with its per-row values loaded ONCE (the element loops call this per entry)
    bool on;
    float lz;
    const int* side_tok;
    const float* side_val;
    int n_side;
    __device__ __forceinline__ TopkRowView(const TopkView& tv, int row) : on(tv.logz != nullptr), lz(on ? tv.logz[row] : 0.f),
        side_tok(on ? tv.side_tok + (long)row * tv.side_cap : nullptr), side_val(on ? tv.side_val + (long)row * tv.side_cap : nullptr), n_side(on ? tv.side_n[row] : 0) {}
    __device__ __forceinline__ float operator()(float x, int v) const { return on ? beam_score_of(x, v, lz, side_tok, side_val, n_side) : x; }
};

// ---- top-k (k <= BEAM_TOPK), ties -> lowest index.  Stage 1: the slice of a (part, row) is copied to LDS and the
//      block argmax is taken k times; stage 2: the same over the LSE_PARTS * k survivors of a row. ----
__device__ __forceinline__ void amax_merge(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}
__device__ void block_topk(float* vals, const int* idx, int n, int k, float* out_val, int* out_idx) {
    // Two levels, one barrier: every WAVE finds the top k of its own threads' elements with wave-level reductions (a thread keeps the best of ITS elements
    // e = tid, tid + blockDim, .. in registers; a round is one shuffle reduction, and only the winner's owner rescans its handful of elements), then wave 0
    // takes the top k of the waves' k-lists.  The top k of a union of per-wave top-k lists is the global top k under the same total order (value, then
    // lowest index), so the results are those of k block-wide argmax rounds -- which cost 2 barriers and n LDS reads per round (the 64-part stage at 256 rows:
    // 16 384 workgroups, 195 us).
    __shared__ float wv[4 * BEAM_TOPK];
    __shared__ int wi[4 * BEAM_TOPK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float mv;
    int mi, mpos;
    auto rescan = [&]() {
        mv = -INFINITY; mi = 0x7fffffff; mpos = -1;
        for (int e = threadIdx.x; e < n; e += blockDim.x) {
            const float x = vals[e];
            const int id = idx ? idx[e] : e;
            if (x > mv || (x == mv && id < mi)) { mv = x; mi = id; mpos = e; }
        }
    };
    rescan();
    for (int it = 0; it < k; ++it) {
        float bv = mv;
        int bi = mi;
        wave_argmax_all(bv, bi);  // (DPP: common.h)
        if (lane == 0) { wv[wave * BEAM_TOPK + it] = bv; wi[wave * BEAM_TOPK + it] = bi; }
        if (mpos >= 0 && mv == bv && mi == bi) {  // the owner retires the winner (indices are distinct but for the -inf sentinels: every holder of one retires it) and finds its next best
            vals[mpos] = -INFINITY;
            rescan();
        }
    }
    __syncthreads();
    if (wave != 0) return;
    // wave 0: the same procedure over the nw * k candidates (candidate c = entry c % k of wave c / k; lane l owns c = l, l + 64, ..)
    const int nc = nw * k;
    float cv;
    int ci, cpos;
    auto rescan2 = [&]() {
        cv = -INFINITY; ci = 0x7fffffff; cpos = -1;
        for (int c = lane; c < nc; c += 64) {
            const int q = (c / k) * BEAM_TOPK + c % k;
            if (wv[q] > cv || (wv[q] == cv && wi[q] < ci)) { cv = wv[q]; ci = wi[q]; cpos = q; }
        }
    };
    rescan2();
    for (int it = 0; it < k; ++it) {
        float bv = cv;
        int bi = ci;
        wave_argmax_all(bv, bi);
        if (lane == 0) { out_val[it] = bv; out_idx[it] = bi; }
        if (cpos >= 0 && cv == bv && ci == bi) {  // the owner of the winning entry retires it
            wv[cpos] = -INFINITY;
            wi[cpos] = 0x7fffffff;
            rescan2();
        }
    }
}
#define TOPK_SLICE 2048
__global__ __launch_bounds__(256) void topk_part_kernel(const float* __restrict__ scores, long ld, int vocab, int k,
                                                        float* __restrict__ cval, int* __restrict__ cidx, TopkView view) {
    __shared__ float vals[TOPK_SLICE];
    __shared__ int ids[TOPK_SLICE];
    const float* L = scores + (long)blockIdx.y * ld;
    const int per = (vocab + LSE_PARTS - 1) / LSE_PARTS;  // <= TOPK_SLICE (checked by the launcher)
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    const int n = max(hi - lo, 0);
    for (int e = threadIdx.x; e < n; e += blockDim.x) { vals[e] = topk_view(view, blockIdx.y, L[lo + e], lo + e); ids[e] = lo + e; }
    __syncthreads();
    block_topk(vals, ids, n, k, cval + ((long)blockIdx.y * LSE_PARTS + blockIdx.x) * BEAM_TOPK,
               cidx + ((long)blockIdx.y * LSE_PARTS + blockIdx.x) * BEAM_TOPK);
}
__global__ __launch_bounds__(256) void topk_final_kernel(const float* __restrict__ cval, const int* __restrict__ cidx, int k,
                                                         float* __restrict__ out_val, int* __restrict__ out_idx) {
    __shared__ float vals[LSE_PARTS * BEAM_TOPK];
    __shared__ int ids[LSE_PARTS * BEAM_TOPK];
    int n = 0;
    for (int e = threadIdx.x; e < LSE_PARTS * k; e += blockDim.x) {
        const int part = e / k, j = e % k;
        vals[e] = cval[((long)blockIdx.x * LSE_PARTS + part) * BEAM_TOPK + j];
        ids[e] = cidx[((long)blockIdx.x * LSE_PARTS + part) * BEAM_TOPK + j];
    }
    n = LSE_PARTS * k;
    __syncthreads();
    block_topk(vals, ids, n, k, out_val + (long)blockIdx.x * BEAM_TOPK, out_idx + (long)blockIdx.x * BEAM_TOPK);
}

int launch_log_softmax(float* logits, long ld, int vocab, float* pmax, float* psum, int rows, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(logits) & 15) != 0 || vocab > LSE_PARTS * 4096) return ISST_ERR_ARG;  // (lse_part_kernel: 16-byte loads, one trip)
    hipLaunchKernelGGL(lse_part_kernel, dim3(LSE_PARTS, rows), dim3(256), 0, s, logits, ld, vocab, pmax, psum);
    hipLaunchKernelGGL(lse_apply_kernel, dim3(LSE_PARTS, rows), dim3(256), 0, s, logits, ld, vocab, pmax, psum);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
// ---- stage 1, many rows (beam search over many streams): a (part, row) workgroup SCANS its slice once -- every thread keeps the best KR of its elements in
//      registers (an unsorted list and its worst entry: a new element replaces the worst one) -- and the lists are then drained like block_topk's: a wave round is
//      one shuffle reduction over the lanes' current bests, the winner's owner drops that entry; wave 0 merges the waves' lists.  One pass over the scores, no
//      per-slice LDS image: 256 rows x 64 slices of 2 005 elements through block_topk took 195-212 us, whatever its rounds cost (profiles/r04/topk_scan_ab.txt). ----
__device__ __forceinline__ bool topk_better(float x, int i, float y, int j) { return x > y || (x == y && i < j); }
template <int KR>
__device__ void topk_scan_body(const float* __restrict__ scores, long ld, int vocab, int k, int parts, float* __restrict__ cval, int* __restrict__ cidx, const TopkView& view) {
    __shared__ float wv[4 * BEAM_TOPK];
    __shared__ int wi[4 * BEAM_TOPK];
    const float* L = scores + (long)blockIdx.y * ld;
    const int per = ((vocab + parts - 1) / parts + 3) & ~3;  // (slices start at multiples of 4 elements: 16-byte loads; rows are 64-byte aligned, ld % 16 == 0)
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const TopkRowView rv(view, blockIdx.y);
    float tv[KR];
    int ti[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    float wv_ = -INFINITY;  // the list's worst entry (the one a better element replaces) ...
    int wi_ = 0x7fffffff;
    // (four 16-byte loads in flight per thread: with one, a (slice, row) workgroup's scan was a chain of ~30 dependent round trips -- 110 us for 256 rows x 128 263)
    for (int vb = lo + (int)threadIdx.x * 4; vb < hi; vb += 4096) {
        f32x4_t qq[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int v4 = vb + w * 1024;
            qq[w] = *reinterpret_cast<const f32x4_t*>(L + (v4 < hi ? v4 : lo));  // UNCONDITIONAL (a piece past the slice re-reads its first one; its elements fail `v0 < hi` below): no branch around a load
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
        const int v4 = vb + w * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const int v0 = v4 + u;
        const float x = rv(qq[w][u], v0);
        if (v0 < hi && topk_better(x, v0, wv_, wi_)) {
            bool done = false;
#pragma unroll
            for (int j = 0; j < KR; ++j) {  // replace the (first) worst entry
                const bool here = !done && tv[j] == wv_ && ti[j] == wi_;
                tv[j] = here ? x : tv[j];
                ti[j] = here ? v0 : ti[j];
                done = done || here;
            }
            wv_ = tv[0]; wi_ = ti[0];
#pragma unroll
            for (int j = 1; j < KR; ++j)
                if (topk_better(wv_, wi_, tv[j], ti[j])) { wv_ = tv[j]; wi_ = ti[j]; }
        }
        }
        }
    }
    // drain: lane's current best of its list
    float mv;
    int mi;
    auto rescan = [&]() {
        mv = tv[0]; mi = ti[0];
#pragma unroll
        for (int j = 1; j < KR; ++j)
            if (topk_better(tv[j], ti[j], mv, mi)) { mv = tv[j]; mi = ti[j]; }
    };
    rescan();
    for (int it = 0; it < k; ++it) {
        float bv = mv;
        int bi = mi;
        wave_argmax_all(bv, bi);  // (DPP: common.h)
        if (lane == 0) { wv[wave * BEAM_TOPK + it] = bv; wi[wave * BEAM_TOPK + it] = bi; }
        if (bi == mi && bi != 0x7fffffff) {  // (real indices are distinct: exactly one lane owns the winner)
#pragma unroll
            for (int j = 0; j < KR; ++j) {
                const bool here = ti[j] == bi;
                tv[j] = here ? -INFINITY : tv[j];
                ti[j] = here ? 0x7fffffff : ti[j];
            }
            rescan();
        }
    }
    __syncthreads();
    if (wave != 0) return;
    const int nc = 4 * k;
    float cv;
    int ci, cpos;
    auto rescan2 = [&]() {
        cv = -INFINITY; ci = 0x7fffffff; cpos = -1;
        for (int c = lane; c < nc; c += 64) {
            const int q = (c / k) * BEAM_TOPK + c % k;
            if (topk_better(wv[q], wi[q], cv, ci) || cpos < 0) { cv = wv[q]; ci = wi[q]; cpos = q; }
        }
    };
    rescan2();
    float* oc = cval + ((long)blockIdx.y * parts + blockIdx.x) * BEAM_TOPK;
    int* oi_ = cidx + ((long)blockIdx.y * parts + blockIdx.x) * BEAM_TOPK;
    for (int it = 0; it < k; ++it) {
        float bv = cpos >= 0 ? cv : -INFINITY;
        int bi = cpos >= 0 ? ci : 0x7fffffff;
        wave_argmax_all(bv, bi);
        if (lane == 0) { oc[it] = bv; oi_[it] = bi; }
        if (cpos >= 0 && cv == bv && ci == bi) {  // the owner of the winning entry retires it (sentinels: every holder of one)
            wv[cpos] = -INFINITY;
            wi[cpos] = 0x7fffffff;
            rescan2();
        }
    }
}
template <int KR>
__global__ __launch_bounds__(256) void topk_scan_kernel(const float* __restrict__ scores, long ld, int vocab, int k, int parts, float* __restrict__ cval,
                                                        int* __restrict__ cidx, TopkView view) {
    topk_scan_body<KR>(scores, ld, vocab, k, parts, cval, cidx, view);
}
// ---- stage 1 by THRESHOLD (round 5): the scan above keeps a candidate list per thread, and with 64 lanes per wave SOME lane inserts at nearly every element, so
//      the whole wave walks the insertion code every time -- ~64 instructions per element, 125 us for 256 rows x 128 263 (22.8 us for 4 rows), twice the time of
//      the memory pass (profiles/r05/final/trace_busy_prof64x4.txt).  Here a thread keeps its NF4 16-byte pieces of the slice in registers and only their maximum;
//      the k-th largest of the workgroup's 256 thread maxima is a lower bound T of the slice's k-th largest entry (k entries >= T exist), so every entry of the
//      slice's top k is >= T: the threads append their entries >= T (a handful) to an LDS list and block_topk takes the k best of it under the same total order
//      (value desc, index asc).  Ties or bans can make that list long (all entries equal; fewer than k finite ones: T = -inf): past TS_CAP entries the workgroup
//      falls back to the exact scan above. ----
#define TS_CAP 1024
template <int KR, int NF4>
__global__ __launch_bounds__(256) void topk_thresh_kernel(const float* __restrict__ scores, long ld, int vocab, int k, int parts, float* __restrict__ cval,
                                                          int* __restrict__ cidx, TopkView view) {
    __shared__ float tmax[256];
    __shared__ float lv[TS_CAP];
    __shared__ int li[TS_CAP];
    __shared__ int s_cnt;
    __shared__ float s_T;
    const float* L = scores + (long)blockIdx.y * ld;
    const int per = ((vocab + parts - 1) / parts + 3) & ~3;
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    const int tid = threadIdx.x;
    if (tid == 0) { s_cnt = 0; s_T = INFINITY; }
    const TopkRowView rv(view, blockIdx.y);
    f32x4_t q[NF4];
#pragma unroll
    for (int u = 0; u < NF4; ++u) {
        const int v4 = lo + (u * 256 + tid) * 4;
        q[u] = *reinterpret_cast<const f32x4_t*>(L + (v4 < hi ? v4 : lo));  // unconditional load; a piece past the slice is masked below
    }
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < NF4; ++u) {
        const int v4 = lo + (u * 256 + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = v4 + e < hi ? rv(q[u][e], v4 + e) : -INFINITY;
            q[u][e] = x;
            m = fmaxf(m, x);
        }
    }
    tmax[tid] = m;
    __syncthreads();
    {   // rank of this thread's maximum among the 256 (ties by thread id): the one of rank k - 1 is the threshold
        int rank = 0;
        for (int j = 0; j < 256; ++j) {
            const float o = tmax[j];
            rank += (o > m || (o == m && j < tid)) ? 1 : 0;
        }
        if (rank == k - 1) s_T = m;
    }
    __syncthreads();
    const float T = s_T;
#pragma unroll
    for (int u = 0; u < NF4; ++u) {
        const int v4 = lo + (u * 256 + tid) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (q[u][e] >= T && v4 + e < hi) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < TS_CAP) { lv[pos] = q[u][e]; li[pos] = v4 + e; }
            }
        }
    }
    __syncthreads();
    const int n = s_cnt;
    if (n > TS_CAP) {  // (workgroup-uniform) ties / bans: the exact scan
        topk_scan_body<KR>(scores, ld, vocab, k, parts, cval, cidx, view);
        return;
    }
    block_topk(lv, li, n, k, cval + ((long)blockIdx.y * parts + blockIdx.x) * BEAM_TOPK, cidx + ((long)blockIdx.y * parts + blockIdx.x) * BEAM_TOPK);
}
// stage 2, one WAVE per row: the parts x k survivors of a row sit in registers (<= TFW_CPL per lane); a round is one shuffle reduction over the lanes'
// current bests under the same total order (value desc, index asc) and the winner's owner drops that entry.  The block-wide form above (LDS image, two
// levels, barriers) took 17.9 us for 4 rows x 512 candidates inside a one-stream beam step (profiles/r05/trace_busy_profb4_b.txt).
#define TFW_CPL 16
__global__ __launch_bounds__(256) void topk_final_wave_kernel(const float* __restrict__ cval, const int* __restrict__ cidx, int k, int parts, int rows,
                                                              float* __restrict__ out_val, int* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nc = parts * k;
    float tv[TFW_CPL];
    int ti[TFW_CPL];
#pragma unroll
    for (int j = 0; j < TFW_CPL; ++j) {
        const int e = lane + 64 * j;
        const bool in = e < nc;
        const long q = ((long)row * parts + (in ? e / k : 0)) * BEAM_TOPK + (in ? e % k : 0);
        tv[j] = in ? cval[q] : -INFINITY;
        ti[j] = in ? cidx[q] : 0x7fffffff;
    }
    float mv;
    int mi;
    auto rescan = [&]() {
        mv = tv[0]; mi = ti[0];
#pragma unroll
        for (int j = 1; j < TFW_CPL; ++j)
            if (topk_better(tv[j], ti[j], mv, mi)) { mv = tv[j]; mi = ti[j]; }
    };
    rescan();
    for (int it = 0; it < k; ++it) {
        float bv = mv;
        int bi = mi;
        wave_argmax_all(bv, bi);  // (DPP: common.h)
        if (lane == 0) { out_val[(long)row * BEAM_TOPK + it] = bv; out_idx[(long)row * BEAM_TOPK + it] = bi; }
        if (bi == mi && bi != 0x7fffffff) {  // (token indices of a row's survivors are distinct: exactly one lane owns the winner)
#pragma unroll
            for (int j = 0; j < TFW_CPL; ++j) {
                const bool here = ti[j] == bi;
                tv[j] = here ? -INFINITY : tv[j];
                ti[j] = here ? 0x7fffffff : ti[j];
            }
            rescan();
        }
    }
}

static bool g_topk_thresh = true;  // ISST_TOPK_THRESH=0 (read once): the per-thread-list scan for every workgroup (A/B)
static int launch_topk_rows_view(const float* scores, long ld, int vocab, int k, float* cval, int* cidx, float* out_val, int* out_idx, int rows, const TopkView& view,
                                 hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (k < 1 || k > BEAM_TOPK || (vocab + LSE_PARTS - 1) / LSE_PARTS > TOPK_SLICE) return ISST_ERR_ARG;
    static const bool env_read = [] { if (const char* e = getenv("ISST_TOPK_THRESH")) g_topk_thresh = e[0] && e[0] != '0'; return true; }();
    (void)env_read;
    if (k <= 16 && ld % 16 == 0 && (reinterpret_cast<uintptr_t>(scores) & 63) == 0) {  // one scan per (part, row) with the candidates in registers; enough parts for ~1024 workgroups
        int parts = 1024 / rows;
        const int parts_min = (vocab + 16 * 1024 - 1) / (16 * 1024);  // a thread of the threshold kernel holds at most 16 pieces of 4 entries
        parts = parts < parts_min ? parts_min : parts;
        parts = parts < 1 ? 1 : (parts > LSE_PARTS ? LSE_PARTS : parts);
        const int per = ((vocab + parts - 1) / parts + 3) & ~3, nf4 = (per + 1023) / 1024;
        // (by rocprofv3 inside a beam step, round 5, both with the DPP argmax: 256 rows 77.7 us by threshold against 96.5 us; 4 rows 17.5 against 15.6 us --
        //  the few-row launches are their reduction rounds, not their scan: profiles/r05/beam_tail_kernels.txt)
        if (g_topk_thresh && nf4 <= 16 && rows >= 16) {
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(parts, rows), dim3(256), 0, s, scores, ld, vocab, k, parts, cval, cidx, view); };
            if (k <= 8) { if (nf4 <= 2) go(topk_thresh_kernel<8, 2>); else if (nf4 <= 4) go(topk_thresh_kernel<8, 4>); else if (nf4 <= 8) go(topk_thresh_kernel<8, 8>); else go(topk_thresh_kernel<8, 16>); }
            else { if (nf4 <= 2) go(topk_thresh_kernel<16, 2>); else if (nf4 <= 4) go(topk_thresh_kernel<16, 4>); else if (nf4 <= 8) go(topk_thresh_kernel<16, 8>); else go(topk_thresh_kernel<16, 16>); }
        } else if (k <= 8) hipLaunchKernelGGL(topk_scan_kernel<8>, dim3(parts, rows), dim3(256), 0, s, scores, ld, vocab, k, parts, cval, cidx, view);
        else hipLaunchKernelGGL(topk_scan_kernel<16>, dim3(parts, rows), dim3(256), 0, s, scores, ld, vocab, k, parts, cval, cidx, view);
        static_assert(LSE_PARTS * 16 <= 64 * TFW_CPL, "a wave holds every survivor of a row (parts <= LSE_PARTS, k <= 16)");
        hipLaunchKernelGGL(topk_final_wave_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, cval, cidx, k, parts, rows, out_val, out_idx);
        return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    }
    hipLaunchKernelGGL(topk_part_kernel, dim3(LSE_PARTS, rows), dim3(256), 0, s, scores, ld, vocab, k, cval, cidx, view);
    hipLaunchKernelGGL(topk_final_kernel, dim3(rows), dim3(256), 0, s, cval, cidx, k, out_val, out_idx);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
int launch_topk_rows(const float* scores, long ld, int vocab, int k, float* cval, int* cidx, float* out_val, int* out_idx, int rows,
                     hipStream_t s) {
    return launch_topk_rows_view(scores, ld, vocab, k, cval, cidx, out_val, out_idx, rows, TopkView{}, s);
}
// the scoring tail of a beam step without the in-place log-softmax: lse_part -> beam_process (log Z, side lists, bans) -> top-k through the view
int launch_beam_scores(float* logits, long ld, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress, int n_suppress,
                       float rep_penalty, int ngram, int enc_ngram, float* pmax, float* psum, const BeamScoreView& v, int k, float* cval, int* cidx,
                       float* out_val, int* out_idx, int rows, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (ld % 4 != 0 || (reinterpret_cast<uintptr_t>(logits) & 15) != 0 || vocab > LSE_PARTS * 4096 || !v.logz || !v.side_tok || !v.side_val || !v.side_n || v.side_cap < 1) return ISST_ERR_ARG;
    hipLaunchKernelGGL(lse_part_kernel, dim3(LSE_PARTS, rows), dim3(256), 0, s, logits, ld, vocab, pmax, psum);
    hipLaunchKernelGGL(beam_process_kernel, dim3(rows), dim3(256), 0, s, logits, ld, ss, ids_pool, enc_pool, suppress, n_suppress, rep_penalty, ngram, enc_ngram, pmax, psum,
                       v.logz, v.side_tok, v.side_val, v.side_n, v.side_cap);
    if (hipGetLastError() != hipSuccess) return ISST_ERR_HIP;
    TopkView tv{v.logz, v.side_tok, v.side_val, v.side_n, v.side_cap};
    return launch_topk_rows_view(logits, ld, vocab, k, cval, cidx, out_val, out_idx, rows, tv, s);
}
