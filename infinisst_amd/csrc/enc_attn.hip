// Streaming wav2vec2 encoder self-attention for gfx950: per-layer KV ring of UNROTATED keys, RoPE applied on read,
// block-bidirectional mask with a sliding window.
//
// Reference: uni_mha_forward (model/patches/patch_speech_encoder.py:692-933): append unrotated K,V to the layer
// cache (:797-821), rotate q at offsets K-Q..K-1 and ALL cached k at 0..K-1 (:824), scores = bmm rounded to bf16
// (:853), + mask from get_attn_mask_training/_inference (:30-77), fp32 softmax (:887-889), probs rounded to bf16
// (:890), bmm with V (:915).  The cache is trimmed to the last max_cache_size keys before the layer call
// (:516-520): here that is a ring-start advance done by the host (EncStreamView.start), no copy.
//
// Data layout in HBM: per stream and layer K ring [heads][cap][64] bf16 and V ring likewise; logical key j of a
// stream lives in slot (start + j) mod cap.  The mask needs no tensor: row i may see columns [lo_i, hi_i).
#include "common.h"
#include "kernels.h"

#define ENC_HD 64
#define ENC_KROW 72  // LDS row stride in bf16 (144 B): 16 consecutive rows hit disjoint bank quads for ds_read_b128
#define ENC_MAX_IT 12

__global__ void enc_kv_append_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ kring, bf16_t* __restrict__ vring,
                                     long stream_stride, const EncStreamView* __restrict__ sv, int Q, int heads, int cap,
                                     int max_cache) {
    const int i = blockIdx.x, s = blockIdx.y;
    const int h = threadIdx.x >> 3, c8 = threadIdx.x & 7;
    const int D = heads * ENC_HD;
    const int len = min(sv[s].prefix, max_cache);
    int phys = sv[s].start + len + i;
    phys %= cap;
    const bf16_t* row = qkv + ((long)s * Q + i) * 3 * D;
    const long dst = (long)s * stream_stride + ((long)h * cap + phys) * ENC_HD + c8 * 8;
    *reinterpret_cast<u32x4_t*>(kring + dst) = *reinterpret_cast<const u32x4_t*>(row + D + h * ENC_HD + c8 * 8);
    *reinterpret_cast<u32x4_t*>(vring + dst) = *reinterpret_cast<const u32x4_t*>(row + 2 * D + h * ENC_HD + c8 * 8);
}

int launch_enc_kv_append(const bf16_t* qkv, bf16_t* kring, bf16_t* vring, long stream_stride, const EncStreamView* sv,
                         int n_streams, int Q, int heads, int cap, int max_cache, hipStream_t s) {
    if (heads * 8 > 1024) return ISST_ERR_ARG;
    hipLaunchKernelGGL(enc_kv_append_kernel, dim3(Q, n_streams), dim3(heads * 8), 0, s, qkv, kring, vring, stream_stride, sv,
                       Q, heads, cap, max_cache);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// interleaved-pair rotation of 8 consecutive dims (4 pairs) [3P rotary_embedding_torch semantics, see oracle]
__device__ __forceinline__ void rot8(const float* x, const float* c, const float* sn, int round_each, float* y) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        if (round_each) {
            y[2 * i] = bfr(bfr(a * c[i]) + bfr(-b * sn[i]));
            y[2 * i + 1] = bfr(bfr(b * c[i]) + bfr(a * sn[i]));
        } else {
            y[2 * i] = bfr(a * c[i] - b * sn[i]);
            y[2 * i + 1] = bfr(b * c[i] + a * sn[i]);
        }
    }
}

__global__ __launch_bounds__(256) void enc_attention_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kring,
                                                            const bf16_t* __restrict__ vring, long stream_stride,
                                                            const EncStreamView* __restrict__ sv,
                                                            const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                            int round_each, bf16_t* __restrict__ out, int Q, int heads, int cap,
                                                            int C, int bs, int kalloc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* Ks = reinterpret_cast<bf16_t*>(smem);                                    // [kalloc][ENC_KROW]
    float* probs = reinterpret_cast<float*>(smem + (size_t)kalloc * ENC_KROW * 2);   // [4][kalloc]
    const int h = blockIdx.x, qt = blockIdx.y, s = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = sv[s].prefix, start = sv[s].start;
    const int len = min(P, C);
    const int K = len + Q;
    const int off = max(0, P - C);
    const int D = heads * ENC_HD;
    const bf16_t* kr = kring + (long)s * stream_stride + (long)h * cap * ENC_HD;
    const bf16_t* vr = vring + (long)s * stream_stride + (long)h * cap * ENC_HD;

    // ---- phase 1: rotate the whole key window once into LDS (bf16, like the reference's rotated K) ----
    for (int item = tid; item < K * 8; item += 256) {
        const int j = item >> 3, c8 = item & 7;
        int phys = start + j;
        if (phys >= cap) phys -= cap;
        float x[8], y[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(kr + (long)phys * ENC_HD + c8 * 8), x);
        const f32x4_t c = *reinterpret_cast<const f32x4_t*>(rope_cos + (long)j * 32 + c8 * 4);
        const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rope_sin + (long)j * 32 + c8 * 4);
        const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
        rot8(x, cc, ss, round_each, y);
        *reinterpret_cast<u32x4_t*>(Ks + (long)j * ENC_KROW + c8 * 8) = pack8(y);
    }
    __syncthreads();

    float* myp = probs + (long)wave * kalloc;
    for (int r = 0; r < 4; ++r) {
        const int qi = qt * 16 + wave * 4 + r;
        if (qi >= Q) break;  // wave-uniform
        // rotated query (every lane holds all 64 dims)
        float q[ENC_HD];
        {
            const bf16_t* qrow = qkv + ((long)s * Q + qi) * 3 * D + h * ENC_HD;
            const int qpos = K - Q + qi;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                float x[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(qrow + c8 * 8), x);
                const f32x4_t c = *reinterpret_cast<const f32x4_t*>(rope_cos + (long)qpos * 32 + c8 * 4);
                const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rope_sin + (long)qpos * 32 + c8 * 4);
                const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
                rot8(x, cc, ss, round_each, &q[c8 * 8]);
            }
        }
        // visible column range of this row (patch_speech_encoder.py:30-77; P == 0 is the training mask)
        const int a = qi + P;
        const int block_end = min((a / bs + 1) * bs, P + Q);
        const int hi = block_end - off;
        const int lo = max(0, qi + P - C) - off;

        float sc[ENC_MAX_IT];
        float mx = -INFINITY;
#pragma unroll
        for (int it = 0; it < ENC_MAX_IT; ++it) {
            const int j = lane + 64 * it;
            float sv_ = -INFINITY;
            if (j >= lo && j < hi) {
                const bf16_t* krow = Ks + (long)j * ENC_KROW;
                float dot = 0.f;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    float kk[8];
                    unpack8(*reinterpret_cast<const u32x4_t*>(krow + c8 * 8), kk);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dot += q[c8 * 8 + e] * kk[e];
                }
                sv_ = bfr(0.125f * dot);  // q * head_dim^-0.5 (:768) is an exact power-of-two scaling
            }
            sc[it] = sv_;
            mx = fmaxf(mx, sv_);
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ENC_MAX_IT; ++it) {
            sc[it] = (sc[it] == -INFINITY) ? 0.f : expf(sc[it] - mx);
            sum += sc[it];
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int it = 0; it < ENC_MAX_IT; ++it) {
            const int j = lane + 64 * it;
            if (j < kalloc) myp[j] = bfr(sc[it] * inv);
        }
        // P.V : lane = output dim
        float acc = 0.f;
        int j = lo;
        for (; j + 4 <= hi; j += 4) {
            float pv[4], vv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int phys = start + j + u;
                if (phys >= cap) phys -= cap;
                vv[u] = bf2f(vr[(long)phys * ENC_HD + lane]);
                pv[u] = myp[j + u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += pv[u] * vv[u];
        }
        for (; j < hi; ++j) {
            int phys = start + j;
            if (phys >= cap) phys -= cap;
            acc += myp[j] * bf2f(vr[(long)phys * ENC_HD + lane]);
        }
        out[((long)s * Q + qi) * D + h * ENC_HD + lane] = f2bf(acc);
    }
}

int launch_enc_attention(const bf16_t* qkv, const bf16_t* kring, const bf16_t* vring, long stream_stride,
                         const EncStreamView* sv, const float* rope_cos, const float* rope_sin, int rope_round_each,
                         bf16_t* out, int n_streams, int Q, int heads, int cap, int max_cache, int blocksize, hipStream_t s) {
    if (Q <= 0 || n_streams <= 0) return ISST_OK;
    const int kalloc = ((max_cache + Q + 63) / 64) * 64;
    if (kalloc > 64 * ENC_MAX_IT || max_cache + Q > cap) return ISST_ERR_ARG;
    const size_t lds = (size_t)kalloc * ENC_KROW * 2 + (size_t)4 * kalloc * sizeof(float);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return ISST_ERR_HIP;
        lds_set = lds;
    }
    dim3 grid(heads, (Q + 15) / 16, n_streams), block(256);
    hipLaunchKernelGGL(enc_attention_kernel, grid, block, lds, s, qkv, kring, vring, stream_stride, sv, rope_cos, rope_sin,
                       rope_round_each, out, Q, heads, cap, max_cache, blocksize, kalloc);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
