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
// fused q-rotation + KV append + split-KV attention partials.
// grid = (splits of 64 keys, kv_heads, rows), block = 256 = 16 groups of 16 lanes; every group owns 4 keys of the
// split and issues all their loads (K, V, cos, sin: 24 x 8 B) before the first use, so a block pays ONE round of
// memory latency.  Keys written by this launch (logical position >= new_start, i.e. the prompt rows of a prefill or
// the row itself in a decode step) are read straight from the qkv rows; the group that meets key == row_pos stores
// that row's unrotated k and v into the arena, so every new key is appended exactly once and nobody reads a slot
// that another workgroup writes in the same launch.
// partial layout: [row][head][split][2 + 128] fp32 (m, l, o).
// ------------------------------------------------------------------------------------------------
#define KEYS_PER_GROUP (LLM_ATTN_SPLIT / 16)

template <int G>
__global__ __launch_bounds__(256) void llm_attn_partial_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ row_stream,
                                                               const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
                                                               const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
                                                               bf16_t* kpool, bf16_t* vpool, float* __restrict__ partial,
                                                               LlmAttnDims d, int layer, int n_splits) {
    __shared__ float red[16][G][2 + HD];
    const int sp = blockIdx.x, kvh = blockIdx.y, r = blockIdx.z;
    const int p = row_pos[r];
    const int k_lo = sp * LLM_ATTN_SPLIT;
    const int k_hi = min(k_lo + LLM_ATTN_SPLIT, p + 1);
    if (k_lo >= k_hi) return;  // block-uniform
    const LlmStreamView v = sv[row_stream[r]];
    const int tid = threadIdx.x;
    const int j = tid & 15, grp = tid >> 4;
    const int H = d.heads, KV = d.kv_heads;
    const long ldq = (long)(H + 2 * KV) * HD;
    const long slots = (long)d.sys_cap + d.ring_cap;
    const long base = v.kv_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;
    bf16_t* kb = kpool + base;
    bf16_t* vb = vpool + base;
    const float scale = 0.08838834764831845f;  // 1/sqrt(128)

    // ---- issue every load of this group's keys first ----
    u32x2_t kx1[KEYS_PER_GROUP], kx2[KEYS_PER_GROUP], vx1[KEYS_PER_GROUP], vx2[KEYS_PER_GROUP], cx[KEYS_PER_GROUP], sx[KEYS_PER_GROUP];
#pragma unroll
    for (int u = 0; u < KEYS_PER_GROUP; ++u) {
        const int key = k_lo + grp + 16 * u;
        const int kk = key < k_hi ? key : k_lo;  // clamp: loads stay in bounds, result is discarded
        const bf16_t* kp;
        const bf16_t* vp;
        if (kk >= v.new_start) {
            const bf16_t* row = qkv + (long)(v.row0 + (kk - v.new_start)) * ldq;
            kp = row + (long)(H + kvh) * HD;
            vp = row + (long)(H + KV + kvh) * HD;
        } else {
            const long slot = llm_slot(v, d, kk);
            kp = kb + slot * HD;
            vp = vb + slot * HD;
        }
        kx1[u] = *reinterpret_cast<const u32x2_t*>(kp + 4 * j);
        kx2[u] = *reinterpret_cast<const u32x2_t*>(kp + 64 + 4 * j);
        vx1[u] = *reinterpret_cast<const u32x2_t*>(vp + 4 * j);
        vx2[u] = *reinterpret_cast<const u32x2_t*>(vp + 64 + 4 * j);
        cx[u] = *reinterpret_cast<const u32x2_t*>(rope_cos + (long)kk * 64 + 4 * j);
        sx[u] = *reinterpret_cast<const u32x2_t*>(rope_sin + (long)kk * 64 + 4 * j);
    }
    // ---- rotated queries of the G heads sharing this kv head ----
    float q1[G][4], q2[G][4], m[G], l[G], a1[G][4], a2[G][4];
    {
        float c[4], s[4];
        load4(rope_cos + (long)p * 64 + 4 * j, c);
        load4(rope_sin + (long)p * 64 + 4 * j, s);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bf16_t* qh = qkv + (long)r * ldq + (long)(kvh * G + g) * HD;
            float x1[4], x2[4];
            load4(qh + 4 * j, x1);
            load4(qh + 64 + 4 * j, x2);
            rope_half(x1, x2, c, s, q1[g], q2[g]);
            m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a1[g][i] = 0.f; a2[g][i] = 0.f; }
        }
    }
#pragma unroll
    for (int u = 0; u < KEYS_PER_GROUP; ++u) {
        const int key = k_lo + grp + 16 * u;
        if (key < k_hi) {  // uniform within the 16-lane group
            if (key == p) {  // this row's own key: append the unrotated k, v to the arena
                const long slot = llm_slot(v, d, key);
                *reinterpret_cast<u32x2_t*>(kb + slot * HD + 4 * j) = kx1[u];
                *reinterpret_cast<u32x2_t*>(kb + slot * HD + 64 + 4 * j) = kx2[u];
                *reinterpret_cast<u32x2_t*>(vb + slot * HD + 4 * j) = vx1[u];
                *reinterpret_cast<u32x2_t*>(vb + slot * HD + 64 + 4 * j) = vx2[u];
            }
            const float x1[4] = {lo_bf(kx1[u].x), hi_bf(kx1[u].x), lo_bf(kx1[u].y), hi_bf(kx1[u].y)};
            const float x2[4] = {lo_bf(kx2[u].x), hi_bf(kx2[u].x), lo_bf(kx2[u].y), hi_bf(kx2[u].y)};
            const float v1[4] = {lo_bf(vx1[u].x), hi_bf(vx1[u].x), lo_bf(vx1[u].y), hi_bf(vx1[u].y)};
            const float v2[4] = {lo_bf(vx2[u].x), hi_bf(vx2[u].x), lo_bf(vx2[u].y), hi_bf(vx2[u].y)};
            const float c[4] = {lo_bf(cx[u].x), hi_bf(cx[u].x), lo_bf(cx[u].y), hi_bf(cx[u].y)};
            const float s[4] = {lo_bf(sx[u].x), hi_bf(sx[u].x), lo_bf(sx[u].y), hi_bf(sx[u].y)};
            float r1[4], r2[4];
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

// Combine of the split partials.  A separate launch on purpose: folding it into the partial kernel (last-arriver
// ticket + agent-scope release/acquire, or write-through slabs + sc1 loads) measured 28-31 us per layer against
// 12.7 + 7.0 us for two launches on MI355X (profiles/r01), because every split pays the ticket round trip and the
// reducer serialises its slab loads.
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

int launch_llm_attention(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv,
                         const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* vpool,
                         float* partial, bf16_t* out, LlmAttnDims d, int layer, int rows, int max_pos, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    const int G = d.heads / d.kv_heads;
    const int n_splits = llm_attn_splits(max_pos);
    dim3 grid(n_splits, d.kv_heads, rows), block(256);
#define LAUNCH_G(GG) \
    hipLaunchKernelGGL(llm_attn_partial_kernel<GG>, grid, block, 0, s, qkv, row_stream, row_pos, sv, rope_cos, rope_sin, kpool, vpool, partial, d, layer, n_splits)
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
