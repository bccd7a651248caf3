// Streaming wav2vec2 encoder self-attention for gfx950: per-layer KV ring of UNROTATED keys, RoPE applied on read,
// block-bidirectional mask with a sliding window, both products on MFMA.
//
// Reference: uni_mha_forward (model/patches/patch_speech_encoder.py:692-933): append unrotated K,V to the layer
// cache (:797-821), rotate q at offsets K-Q..K-1 and ALL cached k at 0..K-1 (:824), scores = bmm rounded to bf16
// (:853), + mask from get_attn_mask_training/_inference (:30-77), fp32 softmax (:887-889), probs rounded to bf16
// (:890), bmm with V (:915).  The cache is trimmed to the last max_cache_size keys before the layer call
// (:516-520): here that is a ring-start advance done by the host (EncStreamView.start), no copy.
//
// Data layout in HBM, per stream and layer: K ring [heads][cap][64] bf16 (row per key) and V ring TRANSPOSED
// [heads][64][cap] (row per dim), so that the B operand of P.V (8 consecutive keys of one dim) is one 16-byte load.
// Logical key j lives in physical slot (start + j) mod cap.  Attention is a sum over keys, so the kernel walks
// PHYSICAL slots (aligned 16/32-slot tiles) and derives each slot's logical index for RoPE and masking; slots
// outside the window get probability 0.
//
// One workgroup (4 waves) per (head, block of QB query rows, stream):
//   1. S = (q/8) K^T: q fragments (rotated, bf16) stay in registers; every wave takes key tiles nt = w, w+4, ..; the
//      K fragment of a lane is 8 consecutive dims of ONE key row, so the interleaved-pair rotation is lane-local.
//      Scores are rounded to bf16, masked, and written to LDS  S[QB][cap] (bf16).
//   2. row softmax in fp32 over the LDS row, probabilities rounded to bf16 in place.
//   3. O = P V: wave w owns output dims 16w..16w+15; A = P from LDS, B = V^T straight from global.
// The mask needs no tensor: row i may see logical columns [lo_i, hi_i) (closed form of :30-77).
#include "common.h"
#include "kernels.h"

#define ENC_HD 64
#define ENC_SPAD 8  // bf16 elements of padding per S row (keeps rows 16-byte aligned, shifts banks by 4 per row)

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
    const long base = (long)s * stream_stride + (long)h * cap * ENC_HD;
    *reinterpret_cast<u32x4_t*>(kring + base + (long)phys * ENC_HD + c8 * 8) =
        *reinterpret_cast<const u32x4_t*>(row + D + h * ENC_HD + c8 * 8);
    const bf16_t* v = row + 2 * D + h * ENC_HD + c8 * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) vring[base + (long)(c8 * 8 + d) * cap + phys] = v[d];  // transposed: [dim][slot]
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
// 8 dims starting at `dim0` of a row, rotated at position `pos`, as an MFMA operand fragment
__device__ __forceinline__ u32x4_t rot_frag(const bf16_t* p, int pos, int dim0, const float* __restrict__ rope_cos,
                                            const float* __restrict__ rope_sin, int round_each) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const u32x4_t*>(p), x);
    const f32x4_t c = *reinterpret_cast<const f32x4_t*>(rope_cos + (long)pos * 32 + (dim0 >> 1));
    const f32x4_t sn = *reinterpret_cast<const f32x4_t*>(rope_sin + (long)pos * 32 + (dim0 >> 1));
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
    rot8(x, cc, ss, round_each, y);
    return pack8(y);
}

template <int QT>  // m-tiles of 16 query rows per workgroup
__global__ __launch_bounds__(256) void enc_attention_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kring,
                                                            const bf16_t* __restrict__ vring, long stream_stride,
                                                            const EncStreamView* __restrict__ sv,
                                                            const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                            int round_each, bf16_t* __restrict__ out, int Q, int heads, int cap,
                                                            int C, int bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* S = reinterpret_cast<bf16_t*>(smem);  // [QT*16][cap + ENC_SPAD]
    const int ldS = cap + ENC_SPAD;
    const int h = blockIdx.x, qb = blockIdx.y, s = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = sv[s].prefix, start = sv[s].start;
    const int len = min(P, C);
    const int K = len + Q;               // keys in the window (logical 0..K-1)
    const int off = max(0, P - C);
    const int D = heads * ENC_HD;
    const int q0 = qb * (QT * 16);
    const bf16_t* kr = kring + (long)s * stream_stride + (long)h * cap * ENC_HD;
    const bf16_t* vt = vring + (long)s * stream_stride + (long)h * cap * ENC_HD;  // [64][cap]
    const int fr = lane & 15, fq = lane >> 4;

    // ---- rotated query fragments: A[row = fr][k = 8 fq + j] for both 32-dim k-steps ----
    u32x4_t qf[QT][2];
#pragma unroll
    for (int mt = 0; mt < QT; ++mt) {
        const int qi = q0 + mt * 16 + fr;
        const bf16_t* qrow = qkv + ((long)s * Q + qi) * 3 * D + h * ENC_HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[mt][ks] = rot_frag(qrow + ks * 32 + fq * 8, K - Q + qi, ks * 32 + fq * 8, rope_cos, rope_sin, round_each);
    }
    // visible logical column range of the rows this lane holds in the C layout (rows 4 fq + r of each m-tile)
    // (patch_speech_encoder.py:30-77; P == 0 is the training mask)
    int lo[QT][4], hi[QT][4];
#pragma unroll
    for (int mt = 0; mt < QT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + mt * 16 + fq * 4 + r;
            const int a = qi + P;
            hi[mt][r] = min((a / bs + 1) * bs, P + Q) - off;
            lo[mt][r] = max(0, qi + P - C) - off;
        }

    // ---- 1. scores ----
    const int n_tiles = cap >> 4;
    for (int nt = wave; nt < n_tiles; nt += 4) {
        const int cphys = nt * 16 + fr;       // physical slot of this lane's key
        int j = cphys - start;                // logical index
        if (j < 0) j += cap;
        const bool live = j < K;
        const int jpos = live ? j : 0;
        const bf16_t* krow = kr + (long)cphys * ENC_HD;
        u32x4_t kf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[ks] = rot_frag(krow + ks * 32 + fq * 8, jpos, ks * 32 + fq * 8, rope_cos, rope_sin, round_each);
#pragma unroll
        for (int mt = 0; mt < QT; ++mt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qf[mt][ks]), __builtin_bit_cast(bf16x8_t, kf[ks]), acc, 0, 0, 0);
            // C layout: this lane holds column fr (its own key), rows 4 fq + r
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = live && j >= lo[mt][r] && j < hi[mt][r];
                // q * head_dim^-0.5 (:768) is an exact power-of-two scaling, applied to the accumulated dot product
                const bf16_t v = ok ? f2bf(0.125f * acc[r]) : (bf16_t)0xFF80;  // -inf
                S[(long)(mt * 16 + fq * 4 + r) * ldS + cphys] = v;
            }
        }
    }
    __syncthreads();

    // ---- 2. softmax per row (fp32), probabilities rounded to bf16 in place ----
    for (int row = wave; row < QT * 16; row += 4) {
        bf16_t* srow = S + (long)row * ldS;
        float v[2][8];
        float mx = -INFINITY;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int c = (lane + 64 * pass) * 8;
            if (c < cap) {
                unpack8(*reinterpret_cast<const u32x4_t*>(srow + c), v[pass]);
#pragma unroll
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[pass][e]);
            }
        }
        mx = wave_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if ((lane + 64 * pass) * 8 < cap) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    v[pass][e] = (v[pass][e] == -INFINITY) ? 0.f : expf(v[pass][e] - mx);
                    sum += v[pass][e];
                }
            }
        }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int c = (lane + 64 * pass) * 8;
            if (c < cap) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[pass][e] *= inv;
                *reinterpret_cast<u32x4_t*>(srow + c) = pack8(v[pass]);
            }
        }
    }
    __syncthreads();

    // ---- 3. O = P V : wave w owns dims 16w .. 16w+15 ----
    f32x4_t o[QT];
#pragma unroll
    for (int mt = 0; mt < QT; ++mt) o[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bf16_t* vrow = vt + (long)(wave * 16 + fr) * cap + fq * 8;  // B[k = 8 fq + j][col = fr] = V^T[dim][slot]
    const int k_steps = cap >> 5;  // cap is a multiple of 64 -> even
    for (int ks0 = 0; ks0 < k_steps; ks0 += 4) {
        u32x4_t vf[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // four V^T fragments in flight before the first use
            const int ks = ks0 + u < k_steps ? ks0 + u : k_steps - 1;
            vf[u] = *reinterpret_cast<const u32x4_t*>(vrow + ks * 32);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ks0 + u < k_steps) {
#pragma unroll
                for (int mt = 0; mt < QT; ++mt) {
                    const u32x4_t pf = *reinterpret_cast<const u32x4_t*>(S + (long)(mt * 16 + fr) * ldS + (ks0 + u) * 32 + fq * 8);
                    o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf), __builtin_bit_cast(bf16x8_t, vf[u]), o[mt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int mt = 0; mt < QT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + mt * 16 + fq * 4 + r;
            out[((long)s * Q + qi) * D + h * ENC_HD + wave * 16 + fr] = f2bf(o[mt][r]);
        }
}

int launch_enc_attention(const bf16_t* qkv, const bf16_t* kring, const bf16_t* vring, long stream_stride,
                         const EncStreamView* sv, const float* rope_cos, const float* rope_sin, int rope_round_each,
                         bf16_t* out, int n_streams, int Q, int heads, int cap, int max_cache, int blocksize, hipStream_t s) {
    if (Q <= 0 || n_streams <= 0) return ISST_OK;
    if (Q % 16 != 0 || cap % 64 != 0 || cap > 1024 || max_cache + Q > cap) return ISST_ERR_ARG;
    const int QT = (Q % 48 == 0) ? 3 : 1;
    const size_t lds = (size_t)QT * 16 * (cap + ENC_SPAD) * 2;
    dim3 grid(heads, Q / (QT * 16), n_streams), block(256);
    static size_t lds_set[2] = {0, 0};
    if (QT == 3) {
        if (lds > 64 * 1024 && lds > lds_set[0]) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(enc_attention_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
            lds_set[0] = lds;
        }
        hipLaunchKernelGGL(enc_attention_kernel<3>, grid, block, lds, s, qkv, kring, vring, stream_stride, sv, rope_cos, rope_sin,
                           rope_round_each, out, Q, heads, cap, max_cache, blocksize);
    } else {
        hipLaunchKernelGGL(enc_attention_kernel<1>, grid, block, lds, s, qkv, kring, vring, stream_stride, sv, rope_cos, rope_sin,
                           rope_round_each, out, Q, heads, cap, max_cache, blocksize);
    }
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
