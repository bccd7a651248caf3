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

// ---- log_softmax, stage 1: per (part, row) running max and sum of exp ----
#define LSE_PARTS 64
__global__ __launch_bounds__(256) void lse_part_kernel(const float* __restrict__ logits, long ld, int vocab, float* __restrict__ pmax,
                                                       float* __restrict__ psum) {
    __shared__ float sm[4], ss[4];
    const float* L = logits + (long)blockIdx.y * ld;
    const int per = (vocab + LSE_PARTS - 1) / LSE_PARTS;
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    float m = -INFINITY;
    for (int v = lo + threadIdx.x; v < hi; v += blockDim.x) m = fmaxf(m, L[v]);
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    float s = 0.f;
    if (m > -INFINITY)
        for (int v = lo + threadIdx.x; v < hi; v += blockDim.x) s += expf(L[v] - m);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        pmax[blockIdx.y * LSE_PARTS + blockIdx.x] = m;
        psum[blockIdx.y * LSE_PARTS + blockIdx.x] = ss[0] + ss[1] + ss[2] + ss[3];
    }
}
// ---- stage 2: logits <- logits - logZ (in place) ----
__global__ __launch_bounds__(256) void lse_apply_kernel(float* __restrict__ logits, long ld, int vocab, const float* __restrict__ pmax,
                                                        const float* __restrict__ psum) {
    __shared__ float logz;
    if (threadIdx.x < 64) {
        const float m = pmax[blockIdx.y * LSE_PARTS + threadIdx.x];
        const float M = wave_max(m);
        const float s = (m == -INFINITY) ? 0.f : psum[blockIdx.y * LSE_PARTS + threadIdx.x] * expf(m - M);
        const float S = wave_sum(s);
        if (threadIdx.x == 0) logz = M + logf(S);
    }
    __syncthreads();
    float* L = logits + (long)blockIdx.y * ld;
    const int per = (vocab + LSE_PARTS - 1) / LSE_PARTS;
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    for (int v = lo + threadIdx.x; v < hi; v += blockDim.x) L[v] -= logz;
}

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
        int bi = mi, bpos = mpos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const int oi = __shfl_xor(bi, o, WAVE), op = __shfl_xor(bpos, o, WAVE);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; bpos = op; }
        }
        if (lane == 0) { wv[wave * BEAM_TOPK + it] = bv; wi[wave * BEAM_TOPK + it] = bi; }
        if (bpos >= 0 && bpos == mpos) {  // the owner retires the winner (NaN-free inputs: -inf entries never win again) and finds its next best
            vals[bpos] = -INFINITY;
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
        int bi = ci, bpos = cpos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const int oi = __shfl_xor(bi, o, WAVE), op = __shfl_xor(bpos, o, WAVE);
            if (ov > bv || (ov == bv && oi < bi) || (ov == bv && oi == bi && op >= 0 && (bpos < 0 || op < bpos))) { bv = ov; bi = oi; bpos = op; }
        }
        if (lane == 0) { out_val[it] = bv; out_idx[it] = bi; }
        if (bpos >= 0 && bpos == cpos) {  // the one owner of the winning entry retires it
            wv[bpos] = -INFINITY;
            rescan2();
        }
    }
}
#define TOPK_SLICE 2048
__global__ __launch_bounds__(256) void topk_part_kernel(const float* __restrict__ scores, long ld, int vocab, int k,
                                                        float* __restrict__ cval, int* __restrict__ cidx) {
    __shared__ float vals[TOPK_SLICE];
    __shared__ int ids[TOPK_SLICE];
    const float* L = scores + (long)blockIdx.y * ld;
    const int per = (vocab + LSE_PARTS - 1) / LSE_PARTS;  // <= TOPK_SLICE (checked by the launcher)
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    const int n = max(hi - lo, 0);
    for (int e = threadIdx.x; e < n; e += blockDim.x) { vals[e] = L[lo + e]; ids[e] = lo + e; }
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
__global__ __launch_bounds__(256) void topk_scan_kernel(const float* __restrict__ scores, long ld, int vocab, int k, int parts, float* __restrict__ cval,
                                                        int* __restrict__ cidx) {
    __shared__ float wv[4 * BEAM_TOPK];
    __shared__ int wi[4 * BEAM_TOPK];
    const float* L = scores + (long)blockIdx.y * ld;
    const int per = ((vocab + parts - 1) / parts + 3) & ~3;  // (slices start at multiples of 4 elements: 16-byte loads; rows are 64-byte aligned, ld % 16 == 0)
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float tv[KR];
    int ti[KR];
#pragma unroll
    for (int j = 0; j < KR; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    float wv_ = -INFINITY;  // the list's worst entry (the one a better element replaces) ...
    int wi_ = 0x7fffffff;
    for (int v4 = lo + (int)threadIdx.x * 4; v4 < hi; v4 += 1024) {
        const f32x4_t q = *reinterpret_cast<const f32x4_t*>(L + v4);  // (the row's padding up to ld is readable; elements at or past `hi` are skipped below)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const float x = q[u];
        const int v0 = v4 + u;
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
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const int oi = __shfl_xor(bi, o, WAVE);
            if (topk_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
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
        float bv = cv;
        int bi = ci, bpos = cpos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const int oi = __shfl_xor(bi, o, WAVE), op = __shfl_xor(bpos, o, WAVE);
            if (op >= 0 && (bpos < 0 || topk_better(ov, oi, bv, bi) || (ov == bv && oi == bi && op < bpos))) { bv = ov; bi = oi; bpos = op; }
        }
        if (lane == 0) { oc[it] = bv; oi_[it] = bi; }
        if (bpos >= 0 && bpos == cpos) {
            wv[bpos] = -INFINITY;
            wi[bpos] = 0x7fffffff;
            rescan2();
        }
    }
}
// stage 2 over `parts` lists of k
__global__ __launch_bounds__(256) void topk_final_parts_kernel(const float* __restrict__ cval, const int* __restrict__ cidx, int k, int parts,
                                                               float* __restrict__ out_val, int* __restrict__ out_idx) {
    __shared__ float vals[LSE_PARTS * BEAM_TOPK];
    __shared__ int ids[LSE_PARTS * BEAM_TOPK];
    for (int e = threadIdx.x; e < parts * k; e += blockDim.x) {
        const int part = e / k, j = e % k;
        vals[e] = cval[((long)blockIdx.x * parts + part) * BEAM_TOPK + j];
        ids[e] = cidx[((long)blockIdx.x * parts + part) * BEAM_TOPK + j];
    }
    __syncthreads();
    block_topk(vals, ids, parts * k, k, out_val + (long)blockIdx.x * BEAM_TOPK, out_idx + (long)blockIdx.x * BEAM_TOPK);
}

int launch_topk_rows(const float* scores, long ld, int vocab, int k, float* cval, int* cidx, float* out_val, int* out_idx, int rows,
                     hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (k < 1 || k > BEAM_TOPK || (vocab + LSE_PARTS - 1) / LSE_PARTS > TOPK_SLICE) return ISST_ERR_ARG;
    if (rows >= 16 && k <= 16 && ld % 16 == 0 && (reinterpret_cast<uintptr_t>(scores) & 63) == 0) {  // many rows: one scan per (part, row) with the candidates in registers; enough parts for ~1024 workgroups
        int parts = 1024 / rows;
        parts = parts < 1 ? 1 : (parts > LSE_PARTS ? LSE_PARTS : parts);
        if (k <= 8) hipLaunchKernelGGL(topk_scan_kernel<8>, dim3(parts, rows), dim3(256), 0, s, scores, ld, vocab, k, parts, cval, cidx);
        else hipLaunchKernelGGL(topk_scan_kernel<16>, dim3(parts, rows), dim3(256), 0, s, scores, ld, vocab, k, parts, cval, cidx);
        hipLaunchKernelGGL(topk_final_parts_kernel, dim3(rows), dim3(256), 0, s, cval, cidx, k, parts, out_val, out_idx);
        return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    }
    hipLaunchKernelGGL(topk_part_kernel, dim3(LSE_PARTS, rows), dim3(256), 0, s, scores, ld, vocab, k, cval, cidx);
    hipLaunchKernelGGL(topk_final_kernel, dim3(rows), dim3(256), 0, s, cval, cidx, k, out_val, out_idx);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
