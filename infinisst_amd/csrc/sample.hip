// Logits processors + greedy argmax on the device (gfx950), one 1024-thread block per stream.
//
// Reference: the processors the agent configures (agents/infinisst.py:319-329) as built by the patched generate
// (model/patches/patch_hf.py:586-600) and applied by HF `_sample` [3P transformers 4.47.0] to the fp32 copy of the
// last position's logits, in this order: RepetitionPenalty(over this chunk's input_ids) ->
// NoRepeatNGram(n, same ids) -> EncoderNoRepeatNGram(n, last <=100 previous target ids) -> SuppressTokens -> argmax
// (first index on ties).
#include "common.h"
#include "kernels.h"

// stage 1 (one 256-thread block per stream): processors, in place on the logits row
__global__ __launch_bounds__(256) void sample_process_kernel(float* __restrict__ logits, long ld, const SampleStream* __restrict__ ss,
                                                             const int* __restrict__ ids_pool, const int* __restrict__ enc_pool,
                                                             const int* __restrict__ suppress, int n_suppress, float pen, int ngram,
                                                             int enc_ngram) {
    const SampleStream st = ss[blockIdx.x];
    float* L = logits + (long)st.logits_row * ld;
    const int* ids = ids_pool + st.ids_off;
    const int* enc = enc_pool + st.enc_off;
    const int n = st.n_ids, tid = threadIdx.x;
    // 1. repetition penalty, once per distinct token
    if (pen != 1.0f) {
        for (int i = tid; i < n; i += blockDim.x) {
            const int t = ids[i];
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && (ids[j] != t);
            if (first) { const float v = L[t]; L[t] = v < 0.f ? v * pen : v / pen; }
        }
    }
    __syncthreads();
    // 2. n-gram bans: the last (n-1) ids followed by t already occurs in ids (resp. in enc)
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

__device__ __forceinline__ void argmax_merge(float& bv, int& bi, float ov, int oi) {
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
}

// stage 2: grid (SAMPLE_PARTS, streams): partial argmax (first index on ties) of one slice of the row
#define SAMPLE_PARTS 64
__global__ __launch_bounds__(256) void sample_argmax_part_kernel(const float* __restrict__ logits, long ld, int vocab,
                                                                 const SampleStream* __restrict__ ss, float* __restrict__ pval,
                                                                 int* __restrict__ pidx) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    const float* L = logits + (long)ss[blockIdx.y].logits_row * ld;
    const int per = (vocab + SAMPLE_PARTS - 1) / SAMPLE_PARTS;
    const int lo = blockIdx.x * per, hi = min(lo + per, vocab);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lo + threadIdx.x; v < hi; v += blockDim.x) argmax_merge(bv, bi, L[v], v);
    wave_argmax_all(bv, bi);  // (DPP reductions, common.h: the same total order -- value, then lowest index)
    if ((threadIdx.x & 63) == 0) { sval[threadIdx.x >> 6] = bv; sidx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) argmax_merge(bv, bi, sval[w], sidx[w]);
        pval[blockIdx.y * SAMPLE_PARTS + blockIdx.x] = bv;
        pidx[blockIdx.y * SAMPLE_PARTS + blockIdx.x] = bi;
    }
}
// stage 3: one wave per stream merges the SAMPLE_PARTS partial results
__global__ __launch_bounds__(64) void sample_argmax_final_kernel(const float* __restrict__ pval, const int* __restrict__ pidx,
                                                                 int* __restrict__ out_tokens) {
    float bv = pval[blockIdx.x * SAMPLE_PARTS + threadIdx.x];
    int bi = pidx[blockIdx.x * SAMPLE_PARTS + threadIdx.x];
    wave_argmax_all(bv, bi);  // (DPP reductions, common.h: the same total order -- value, then lowest index)
    if (threadIdx.x == 0) out_tokens[blockIdx.x] = bi;
}


// One launch for the whole greedy sampling tail: grid (SAMPLE_PARTS, streams).  Every block applies the processors to the tokens of ITS
// slice of the row (the same code as sample_process_kernel, writes confined to the slice: no block touches another's entries, and the row ends up processed in
// place exactly as before), takes the slice's argmax, publishes it, and draws a ticket; the block that draws a stream's last ticket merges the SAMPLE_PARTS
// partial results (same merge order as sample_argmax_final_kernel: same token, ties to the lowest index) and writes the token to device memory; the block that
// completes the last stream copies all tokens to the pinned host array, bumps a device-side sequence number and publishes it to the host (the host waits for that number instead
// of a stream synchronisation; the number lives on the device so that a captured graph replays correctly).
__global__ __launch_bounds__(256) void sample_fused_kernel(float* __restrict__ logits, long ld, int vocab, const SampleStream* __restrict__ ss,
                                                           const int* __restrict__ ids_pool, const int* __restrict__ enc_pool,
                                                           const int* __restrict__ suppress, int n_suppress, float pen, int ngram, int enc_ngram,
                                                           float* __restrict__ pval, int* __restrict__ pidx, int* __restrict__ tickets /* [0] streams done, [1] sequence number, [2 + stream] parts done */,
                                                           int* __restrict__ out_tokens, int* host_tokens, int* host_seq, SampleAdvance adv) {
    __shared__ float sval[4];
    __shared__ int sidx[4];
    __shared__ int s_last;
    const int part = blockIdx.x, stream = blockIdx.y, n_streams = gridDim.y, tid = threadIdx.x;
    const SampleStream st = ss[stream];
    float* L = logits + (long)st.logits_row * ld;
    const int* ids = ids_pool + st.ids_off;
    const int* enc = enc_pool + st.enc_off;
    const int n = st.n_ids;
    const int per = (vocab + SAMPLE_PARTS - 1) / SAMPLE_PARTS;
    const int lo = part * per, hi = min(lo + per, vocab);
    auto mine = [&](int t) { return t >= lo && t < hi; };
    // 1. repetition penalty, once per distinct token (of this slice)
    if (pen != 1.0f) {
        for (int i = tid; i < n; i += blockDim.x) {
            const int t = ids[i];
            if (!mine(t)) continue;
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && (ids[j] != t);
            if (first) { const float v = L[t]; L[t] = v < 0.f ? v * pen : v / pen; }
        }
    }
    __syncthreads();
    // 2. n-gram bans, 3. suppressed tokens
    if (ngram > 0 && n + 1 >= ngram) {
        const int* key = ids + n - (ngram - 1);
        for (int p = tid; p + ngram <= n; p += blockDim.x) {
            const int t = ids[p + ngram - 1];
            if (!mine(t)) continue;
            bool eq = true;
            for (int q = 0; q < ngram - 1; ++q) eq = eq && (ids[p + q] == key[q]);
            if (eq) L[t] = -INFINITY;
        }
    }
    if (enc_ngram > 0 && n + 1 >= enc_ngram) {
        const int* key = ids + n - (enc_ngram - 1);
        for (int p = tid; p + enc_ngram <= st.n_enc; p += blockDim.x) {
            const int t = enc[p + enc_ngram - 1];
            if (!mine(t)) continue;
            bool eq = true;
            for (int q = 0; q < enc_ngram - 1; ++q) eq = eq && (enc[p + q] == key[q]);
            if (eq) L[t] = -INFINITY;
        }
    }
    for (int i = tid; i < n_suppress; i += blockDim.x) { const int t = suppress[i]; if (mine(t)) L[t] = -INFINITY; }
    __syncthreads();
    // 4. argmax of the slice (first index on ties)
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = lo + tid; v < hi; v += blockDim.x) argmax_merge(bv, bi, L[v], v);
    wave_argmax_all(bv, bi);  // (DPP reductions, common.h: the same total order -- value, then lowest index)
    if ((tid & 63) == 0) { sval[tid >> 6] = bv; sidx[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_merge(bv, bi, sval[w], sidx[w]);
        pval[stream * SAMPLE_PARTS + part] = bv;
        pidx[stream * SAMPLE_PARTS + part] = bi;
        __threadfence();  // the partial result is visible device-wide before the ticket is
        s_last = atomicAdd(&tickets[2 + stream], 1) == SAMPLE_PARTS - 1;
    }
    __syncthreads();
    if (!s_last || tid >= 64) return;
    // 5. the stream's last block: merge the partial results (one wave, the order of sample_argmax_final_kernel)
    __threadfence();
    bv = __hip_atomic_load(pval + stream * SAMPLE_PARTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (written by other CUs: not through this CU's L1)
    bi = __hip_atomic_load(pidx + stream * SAMPLE_PARTS + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    wave_argmax_all(bv, bi);  // (DPP reductions, common.h: the same total order -- value, then lowest index)
    // (ONE system-scope fence per launch: a fence per stream costs ~1.2 us each and they serialise in L2 -- 64 streams: +0.8 ms per step, measured.  So a stream's
    //  last block only publishes device-wide; the block that completes the LAST stream copies all tokens to the host, fences once, then publishes the number)
    int all_done = 0;
    if (tid == 0) {
        out_tokens[stream] = bi;
        tickets[2 + stream] = 0;  // re-armed for the next launch (launch boundary = visibility)
        if (n_streams > 1) {
            __threadfence();
            all_done = atomicAdd(&tickets[0], 1) == n_streams - 1;
        } else {
            all_done = 1;  // (one stream: its last block is the launch's last -- no second ticket, no fences around it: ~4 us of the per-token critical path)
        }
    }
    all_done = __shfl(all_done, 0, WAVE);
    if (!all_done) return;
    if (n_streams > 1) __threadfence();
    for (int i = tid; i < n_streams; i += 64) host_tokens[i] = n_streams > 1 ? __hip_atomic_load(out_tokens + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : bi;
    if (adv.enabled) {  // one stream: the next pass's inputs, on the device (the launch boundary makes them visible to it)
        const int tok = bi;  // (this wave merged stream 0 itself: the reduction left the argmax in every lane)
        if (tid == 0) {
            const SampleStream s0 = adv.ss[0];
            adv.ids_pool[s0.ids_off + s0.n_ids] = tok;
            adv.ss[0].n_ids = s0.n_ids + 1;
            adv.ids[0] = tok;
        }
        const bf16_t* src = adv.embed + (long)tok * adv.D;
        for (int c = tid * 8; c < adv.D; c += 512) *reinterpret_cast<u32x4_t*>(adv.lx + c) = *reinterpret_cast<const u32x4_t*>(src + c);
    }
    __threadfence_system();  // every lane: its tokens are in host memory before the call's sequence number can be
    if (tid == 0) {
        if (n_streams > 1) tickets[0] = 0;
        const int seq = atomicAdd(&tickets[1], 1) + 1;
        *reinterpret_cast<volatile int*>(host_seq) = seq;
    }
}

int launch_sample_fused(float* logits, long ld_logits, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress,
                        int n_suppress, float rep_penalty, int ngram, int enc_ngram, int* out_tokens, float* scratch_val, int* scratch_idx, int* tickets,
                        int* host_tokens, int* host_seq, int n_streams, hipStream_t s, const SampleAdvance* adv) {
    if (n_streams <= 0) return ISST_OK;
    SampleAdvance a{};
    if (adv && adv->enabled) {
        if (n_streams != 1 || adv->D % 8 != 0) return ISST_ERR_ARG;
        a = *adv;
    }
    hipLaunchKernelGGL(sample_fused_kernel, dim3(SAMPLE_PARTS, n_streams), dim3(256), 0, s, logits, ld_logits, vocab, ss, ids_pool, enc_pool, suppress, n_suppress,
                       rep_penalty, ngram, enc_ngram, scratch_val, scratch_idx, tickets, out_tokens, host_tokens, host_seq, a);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

int launch_sample(float* logits, long ld_logits, int vocab, const SampleStream* ss, const int* ids_pool, const int* enc_pool,
                  const int* suppress, int n_suppress, float rep_penalty, int ngram, int enc_ngram, int* out_tokens,
                  float* scratch_val, int* scratch_idx, int n_streams, hipStream_t s) {
    if (n_streams <= 0) return ISST_OK;
    hipLaunchKernelGGL(sample_process_kernel, dim3(n_streams), dim3(256), 0, s, logits, ld_logits, ss, ids_pool, enc_pool, suppress,
                       n_suppress, rep_penalty, ngram, enc_ngram);
    hipLaunchKernelGGL(sample_argmax_part_kernel, dim3(SAMPLE_PARTS, n_streams), dim3(256), 0, s, logits, ld_logits, vocab, ss,
                       scratch_val, scratch_idx);
    hipLaunchKernelGGL(sample_argmax_final_kernel, dim3(n_streams), dim3(64), 0, s, scratch_val, scratch_idx, out_tokens);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

int launch_sample_process(float* logits, long ld_logits, const SampleStream* ss, const int* ids_pool, const int* enc_pool, const int* suppress,
                          int n_suppress, float rep_penalty, int ngram, int enc_ngram, int n_rows, hipStream_t s) {
    if (n_rows <= 0) return ISST_OK;
    hipLaunchKernelGGL(sample_process_kernel, dim3(n_rows), dim3(256), 0, s, logits, ld_logits, ss, ids_pool, enc_pool, suppress, n_suppress,
                       rep_penalty, ngram, enc_ngram);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
