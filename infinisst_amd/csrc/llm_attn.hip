// Llama attention over an UNROTATED KV arena with RoPE applied on read (gfx950).
//
// Reference: llama_sdpa_attention_new_forward (model/patches/patch_llm.py:231-336): cache.update(unrotated K, V)
// (:280-284); q rotated at positions past..total-1 and the ENTIRE K cache at 0..total-1 (:286-299); repeat_kv
// (:304-305); causal SDPA (:320-329).  Because the cache is unrotated, dropping the oldest chunks re-indexes the
// remaining keys (reference agents/infinisst.py:340-361).  Here the arena is a per-stream
// [pinned system prompt][ring] and eviction is a ring-start advance: a key's rotation angle is its LOGICAL index
// at read time, so nothing is copied and nothing is re-rotated in memory.
//
// Layout in HBM: per stream, layer, kv head: [sys_cap + ring_cap][128] bf16 for K and the same for V.
// Logical position p -> slot p (p < sys_len) or sys_cap + (ring_start + p - sys_len) mod ring_cap.
//
// RoPE [3P HF apply_rotary_pos_emb, half-split]: out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)) with the
// bf16 cos/sin table (cos[d+64] == cos[d]); each lane owns dims {4j..4j+3} U {64+4j..64+4j+3}, so the rotation
// partner is lane-local and a 16-lane group covers one 128-dim row with two 128-byte segments.
#include "common.h"
#include "kernels.h"

#define HD 128

__device__ __forceinline__ long llm_slot(const LlmStreamView& v, const LlmAttnDims& d, int p) {
    if (p < v.sys_len) return p;
    int x = v.ring_start + (p - v.sys_len);
    x %= d.ring_cap;
    return (long)d.sys_cap + x;
}

__device__ __forceinline__ void load4(const bf16_t* p, float* f) {
    const u32x2_t v = *reinterpret_cast<const u32x2_t*>(p);
    f[0] = lo_bf(v.x); f[1] = hi_bf(v.x); f[2] = lo_bf(v.y); f[3] = hi_bf(v.y);
}
__device__ __forceinline__ void store4(bf16_t* p, const float* f) {
    u32x2_t v;
    v.x = pack_bf(f[0], f[1]); v.y = pack_bf(f[2], f[3]);
    *reinterpret_cast<u32x2_t*>(p) = v;
}
// x1 = dims 4j.., x2 = dims 64+4j..; c,s = cos/sin of dims 4j..
__device__ __forceinline__ void rope_half(const float* x1, const float* x2, const float* c, const float* s, float* r1, float* r2) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r1[i] = bfr(bfr(x1[i] * c[i]) - bfr(x2[i] * s[i]));
        r2[i] = bfr(bfr(x2[i] * c[i]) + bfr(x1[i] * s[i]));
    }
}

// ------------------------------------------------------------------------------------------------
// q rotation + KV append.  grid = rows, block = 256; item = (head of q|k|v, 16-lane slice j)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void llm_qkv_post_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ row_stream,
                                                           const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
                                                           const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
                                                           bf16_t* __restrict__ qrot, bf16_t* __restrict__ kpool,
                                                           bf16_t* __restrict__ vpool, LlmAttnDims d, int layer) {
    const int r = blockIdx.x;
    const int H = d.heads, KV = d.kv_heads;
    const LlmStreamView v = sv[row_stream[r]];
    const int p = row_pos[r];
    const long slots = (long)d.sys_cap + d.ring_cap;
    const long slot = llm_slot(v, d, p);
    const bf16_t* src_row = qkv + (long)r * (H + 2 * KV) * HD;
    for (int item = threadIdx.x; item < (H + 2 * KV) * 16; item += blockDim.x) {
        const int hh = item >> 4, j = item & 15;
        float x1[4], x2[4];
        load4(src_row + hh * HD + 4 * j, x1);
        load4(src_row + hh * HD + 64 + 4 * j, x2);
        if (hh < H) {
            float c[4], s[4], r1[4], r2[4];
            load4(rope_cos + (long)p * 64 + 4 * j, c);
            load4(rope_sin + (long)p * 64 + 4 * j, s);
            rope_half(x1, x2, c, s, r1, r2);
            bf16_t* dst = qrot + (long)r * H * HD + hh * HD;
            store4(dst + 4 * j, r1);
            store4(dst + 64 + 4 * j, r2);
        } else {
            const bool isk = hh < H + KV;
            const int kvh = isk ? hh - H : hh - H - KV;
            bf16_t* dst = (isk ? kpool : vpool) + v.kv_offset + (long)layer * d.layer_stride + ((long)kvh * slots + slot) * HD;
            store4(dst + 4 * j, x1);
            store4(dst + 64 + 4 * j, x2);
        }
    }
}

int launch_llm_qkv_post(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv,
                        const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* qrot, bf16_t* kpool, bf16_t* vpool,
                        LlmAttnDims d, int layer, int rows, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    hipLaunchKernelGGL(llm_qkv_post_kernel, dim3(rows), dim3(256), 0, s, qkv, row_stream, row_pos, sv, rope_cos, rope_sin, qrot,
                       kpool, vpool, d, layer);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// split-KV attention partials: grid = (splits, kv_heads, rows), block = 256 (16 groups of 16 lanes, one key per
// group per step).  Each group keeps an online softmax (m, l, o[128]) for the G query heads of the kv head.
// partial layout: [row][head][split][2 + 128] fp32.
// ------------------------------------------------------------------------------------------------
template <int G>
__global__ __launch_bounds__(256) void llm_attn_partial_kernel(const bf16_t* __restrict__ qrot, const int* __restrict__ row_stream,
                                                               const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
                                                               const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
                                                               const bf16_t* __restrict__ kpool, const bf16_t* __restrict__ vpool,
                                                               float* __restrict__ partial, LlmAttnDims d, int layer, int n_splits) {
    __shared__ float red[16][G][2 + HD];
    const int sp = blockIdx.x, kvh = blockIdx.y, r = blockIdx.z;
    const int p = row_pos[r];
    const int k_lo = sp * LLM_ATTN_SPLIT;
    const int k_hi = min(k_lo + LLM_ATTN_SPLIT, p + 1);
    if (k_lo >= k_hi) return;  // block-uniform
    const LlmStreamView v = sv[row_stream[r]];
    const int tid = threadIdx.x;
    const int j = tid & 15, grp = tid >> 4;  // 16 groups per block
    const long slots = (long)d.sys_cap + d.ring_cap;
    const long base = v.kv_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;
    const bf16_t* kb = kpool + base;
    const bf16_t* vb = vpool + base;
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)

    float q1[G][4], q2[G][4], m[G], l[G], a1[G][4], a2[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bf16_t* qh = qrot + ((long)r * d.heads + kvh * G + g) * HD;
        load4(qh + 4 * j, q1[g]);
        load4(qh + 64 + 4 * j, q2[g]);
        m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { a1[g][i] = 0.f; a2[g][i] = 0.f; }
    }
    for (int key = k_lo + grp; key < k_hi; key += 16) {
        const long slot = llm_slot(v, d, key);
        float x1[4], x2[4], c[4], s[4], r1[4], r2[4], v1[4], v2[4];
        load4(kb + slot * HD + 4 * j, x1);
        load4(kb + slot * HD + 64 + 4 * j, x2);
        load4(vb + slot * HD + 4 * j, v1);
        load4(vb + slot * HD + 64 + 4 * j, v2);
        load4(rope_cos + (long)key * 64 + 4 * j, c);
        load4(rope_sin + (long)key * 64 + 4 * j, s);
        rope_half(x1, x2, c, s, r1, r2);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float part = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) part += q1[g][i] * r1[i] + q2[g][i] * r2[i];
            part += __shfl_xor(part, 8, WAVE);
            part += __shfl_xor(part, 4, WAVE);
            part += __shfl_xor(part, 2, WAVE);
            part += __shfl_xor(part, 1, WAVE);
            const float sc = part * scale;
            const float mn = fmaxf(m[g], sc);
            const float alpha = expf(m[g] - mn);  // exp(-inf) = 0 on the first key
            const float pe = expf(sc - mn);
            l[g] = l[g] * alpha + pe;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a1[g][i] = a1[g][i] * alpha + pe * v1[i];
                a2[g][i] = a2[g][i] * alpha + pe * v2[i];
            }
            m[g] = mn;
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (j == 0) { red[grp][g][0] = m[g]; red[grp][g][1] = l[g]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[grp][g][2 + 4 * j + i] = a1[g][i];
            red[grp][g][2 + 64 + 4 * j + i] = a2[g][i];
        }
    }
    __syncthreads();
    for (int e = tid; e < G * HD; e += 256) {
        const int g = e / HD, dd = e % HD;
        float M = -INFINITY;
        for (int q = 0; q < 16; ++q) M = fmaxf(M, red[q][g][0]);
        float L = 0.f, O = 0.f;
        for (int q = 0; q < 16; ++q) {
            const float mq = red[q][g][0];
            const float w = (mq == -INFINITY) ? 0.f : expf(mq - M);
            L += red[q][g][1] * w;
            O += red[q][g][2 + dd] * w;
        }
        float* dst = partial + (((long)r * d.heads + kvh * G + g) * n_splits + sp) * (2 + HD);
        if (dd == 0) { dst[0] = M; dst[1] = L; }
        dst[2 + dd] = O;
    }
}

__global__ __launch_bounds__(128) void llm_attn_combine_kernel(const float* __restrict__ partial, const int* __restrict__ row_pos,
                                                               bf16_t* __restrict__ out, int heads, int n_splits) {
    const int h = blockIdx.x, r = blockIdx.y, dd = threadIdx.x;
    const int ns = row_pos[r] / LLM_ATTN_SPLIT + 1;  // splits that hold at least one key
    const float* src = partial + ((long)r * heads + h) * n_splits * (2 + HD);
    float M = -INFINITY;
    for (int s = 0; s < ns; ++s) M = fmaxf(M, src[s * (2 + HD)]);
    float L = 0.f, O = 0.f;
    for (int s = 0; s < ns; ++s) {
        const float w = expf(src[s * (2 + HD)] - M);
        L += src[s * (2 + HD) + 1] * w;
        O += src[s * (2 + HD) + 2 + dd] * w;
    }
    out[((long)r * heads + h) * HD + dd] = f2bf(O / L);
}

int launch_llm_attention(const bf16_t* qrot, const int* row_stream, const int* row_pos, const LlmStreamView* sv,
                         const bf16_t* rope_cos, const bf16_t* rope_sin, const bf16_t* kpool, const bf16_t* vpool,
                         float* partial, bf16_t* out, LlmAttnDims d, int layer, int rows, int max_pos, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    const int G = d.heads / d.kv_heads;
    const int n_splits = llm_attn_splits(max_pos);
    dim3 grid(n_splits, d.kv_heads, rows), block(256);
#define LAUNCH_G(GG) \
    hipLaunchKernelGGL(llm_attn_partial_kernel<GG>, grid, block, 0, s, qrot, row_stream, row_pos, sv, rope_cos, rope_sin, kpool, vpool, partial, d, layer, n_splits)
    switch (G) {
        case 1: LAUNCH_G(1); break;
        case 2: LAUNCH_G(2); break;
        case 4: LAUNCH_G(4); break;
        default: return ISST_ERR_ARG;
    }
#undef LAUNCH_G
    if (hipGetLastError() != hipSuccess) return ISST_ERR_HIP;
    hipLaunchKernelGGL(llm_attn_combine_kernel, dim3(d.heads, rows), dim3(HD), 0, s, partial, row_pos, out, d.heads, n_splits);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
