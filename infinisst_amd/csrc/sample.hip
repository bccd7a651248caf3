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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, o, WAVE), __shfl_xor(bi, o, WAVE));
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_merge(bv, bi, __shfl_xor(bv, o, WAVE), __shfl_xor(bi, o, WAVE));
    if (threadIdx.x == 0) out_tokens[blockIdx.x] = bi;
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
