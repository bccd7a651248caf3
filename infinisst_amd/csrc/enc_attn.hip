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
// The chunk's own keys (logical index >= cached length) are read from the qkv rows and appended to the rings by the
// workgroup of query block 0, so no separate append launch is needed.
// One workgroup (8 waves) per (head, block of QB query rows, stream):
//   1. S = (q/8) K^T: q fragments (rotated, bf16) stay in registers; every wave takes key tiles nt = w, w+8, ..; the
//      K fragment of a lane is 8 consecutive dims of ONE key row, so the interleaved-pair rotation is lane-local.
//      Scores are rounded to bf16, masked, and written to LDS  S[QB][cap] (bf16).
//   2. row softmax in fp32 over the LDS row, probabilities rounded to bf16 in place.
//   3. O = P V: wave w owns output dims 16(w&3)..+15 and half of the key steps (w>>2); A = P from LDS, B = V^T straight
//      from global; the two halves meet in LDS.
// The mask needs no tensor: row i may see logical columns [lo_i, hi_i) (closed form of :30-77).
#include "common.h"
#include "kernels.h"

#define ENC_HD 64
#define ENC_SPAD 8  // bf16 elements of padding per S row (keeps rows 16-byte aligned, shifts banks by 4 per row)
#define ENC_WAVES 8

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

// the same from registers: the raw 8 dims and the table entries were loaded earlier (several key tiles' loads in flight before the first rotation)
__device__ __forceinline__ u32x4_t rot_frag_regs(const u32x4_t& raw, const f32x4_t& c, const f32x4_t& sn, int round_each) {
    float x[8], y[8];
    unpack8(raw, x);
    const float cc[4] = {c.x, c.y, c.z, c.w}, ss[4] = {sn.x, sn.y, sn.z, sn.w};
    rot8(x, cc, ss, round_each, y);
    return pack8(y);
}

// -DISST_ENC_TRACE (make trace): wave 0 of every workgroup stamps the 100 MHz wall clock at entry / queries rotated / scores written / softmax done /
// P.V done / stored (profiles/enc_attn_trace_probe.py)
#ifdef ISST_ENC_TRACE
__device__ unsigned long long g_enc_trace[4096 * 8];
#define ENC_STAMP(i) do { if (threadIdx.x == 0) { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (wg_ < 4096) g_enc_trace[wg_ * 8 + (i)] = wall_clock64(); } } while (0)
extern "C" int isst_debug_enc_trace_read(void* dst, long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_enc_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define ENC_STAMP(i) do {} while (0)
#endif

template <int QT, bool RND>  // QT: m-tiles of 16 query rows per workgroup; RND: every product of the rotation rounds to bf16 (enc_rope_mode "bf16": as a compile-time
                              // constant -- as a runtime flag every rotated pair carried a branch, 160 of them per key tile in a phase that is bound by instruction issue)
__global__ __launch_bounds__(512, QT == 1 ? 2 : 4) void enc_attention_kernel(  // (48-row blocks run two workgroups per CU: 4 waves per SIMD, 128 registers -- stated, not left to luck)
    const bf16_t* __restrict__ qkv, bf16_t* kring, bf16_t* vring, long stream_stride,
                                                            const EncStreamView* __restrict__ sv,
                                                            const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                                            int /*round_each: RND*/, bf16_t* __restrict__ out, int Q, int heads, int cap,
                                                            int C, int bs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    ENC_STAMP(0);
    bf16_t* S = reinterpret_cast<bf16_t*>(smem);  // [QT*16][cap + ENC_SPAD]
    const int ldS = cap + ENC_SPAD;
    float* Ohalf = reinterpret_cast<float*>(smem + (size_t)QT * 16 * ldS * 2);  // [QT*16][64] partial O of the upper key half
    const int h = blockIdx.x, qb = blockIdx.y, s = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int P = sv[s].prefix, start = sv[s].start;
    const int len = min(P, C);
    const int K = len + Q;               // keys in the window (logical 0..K-1)
    const int off = max(0, P - C);
    const int D = heads * ENC_HD;
    const int q0 = qb * (QT * 16);
    bf16_t* kr = kring + (long)s * stream_stride + (long)h * cap * ENC_HD;
    bf16_t* vt = vring + (long)s * stream_stride + (long)h * cap * ENC_HD;  // [64][cap]
    const bf16_t* knew = qkv + (long)s * Q * 3 * D + D + h * ENC_HD;      // K of new frame i at knew + i*3D
    const bf16_t* vnew = qkv + (long)s * Q * 3 * D + 2 * D + h * ENC_HD;  // V of new frame i at vnew + i*3D
    const int fr = lane & 15, fq = lane >> 4;

    // ---- rotated query fragments: A[row = fr][k = 8 fq + j] for both 32-dim k-steps ----
    u32x4_t qf[QT][2];
#pragma unroll
    for (int mt = 0; mt < QT; ++mt) {
        const int qi = q0 + mt * 16 + fr;
        const bf16_t* qrow = qkv + ((long)s * Q + qi) * 3 * D + h * ENC_HD;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[mt][ks] = rot_frag(qrow + ks * 32 + fq * 8, K - Q + qi, ks * 32 + fq * 8, rope_cos, rope_sin, RND ? 1 : 0);
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

#ifdef ISST_ENC_TRACE
    asm volatile("s_nop 0" :: "v"(qf[0][0].x), "v"(qf[0][1].x));
#endif
    ENC_STAMP(1);
    // ---- 1. scores ----
    // One stream (QT == 1: 48 workgroups on 256 CUs, nothing else to hide a round trip behind) walks its 5 key tiles per wave with ALL their loads -- key rows and
    // rotary table entries -- in flight before the first rotation: the phase was a chain of 5 dependent round trips, 9.0 of the launch's 18.4 us
    // (profiles/r05/enc_attention_trace_1_stream.txt).  Many streams (QT == 3, two workgroups per CU, VALU-bound) keep one tile at a time and their registers.
    // (16 waves x 3 tiles instead of 8 x 5 for the lone stream: the phase took 10.9 us instead of 9.0 -- it is bound by the instructions a SIMD has to issue at the clock
    //  the chip holds, not by latency: profiles/r05/enc_attention_trace_1_stream_16_waves_SLOWER.txt)
    constexpr int TCH = QT == 1 ? 5 : 1;
    const int n_tiles = cap >> 4;
    for (int nt0 = wave; nt0 < n_tiles; nt0 += ENC_WAVES * TCH) {
        u32x4_t kraw[TCH][2];
        f32x4_t kc[TCH][2], ksn[TCH][2];
        int jj[TCH], cph[TCH];
        bool lv[TCH];
#pragma unroll
        for (int t = 0; t < TCH; ++t) {
            const int nt = min(nt0 + t * ENC_WAVES, n_tiles - 1);  // (a tile past the end re-reads the last one; it is not used)
            const int cphys = nt * 16 + fr;       // physical slot of this lane's key
            int j = cphys - start;                // logical index
            if (j < 0) j += cap;
            const bool live = j < K;
            const int jpos = live ? j : 0;
            const bool is_new = live && j >= len;  // written by this chunk: still only in the qkv rows
            const bf16_t* krow = is_new ? knew + (long)(j - len) * 3 * D : kr + (long)cphys * ENC_HD;
            jj[t] = j; cph[t] = cphys; lv[t] = live;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                kraw[t][ks] = *reinterpret_cast<const u32x4_t*>(krow + ks * 32 + fq * 8);
                kc[t][ks] = *reinterpret_cast<const f32x4_t*>(rope_cos + (long)jpos * 32 + ((ks * 32 + fq * 8) >> 1));
                ksn[t][ks] = *reinterpret_cast<const f32x4_t*>(rope_sin + (long)jpos * 32 + ((ks * 32 + fq * 8) >> 1));
            }
        }
#pragma unroll
        for (int t = 0; t < TCH; ++t) {
            if (nt0 + t * ENC_WAVES >= n_tiles) break;  // (wave-uniform)
            const int j = jj[t], cphys = cph[t];
            const bool live = lv[t];
            if (live && j >= len && qb == 0) {  // append the unrotated key to the ring (query block 0 owns the append)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) *reinterpret_cast<u32x4_t*>(kr + (long)cphys * ENC_HD + ks * 32 + fq * 8) = kraw[t][ks];
            }
            u32x4_t kf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[ks] = rot_frag_regs(kraw[t][ks], kc[t][ks], ksn[t][ks], RND ? 1 : 0);
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
    }
    __syncthreads();
    ENC_STAMP(2);

    // ---- 2. softmax per row (fp32), probabilities rounded to bf16 in place ----
    for (int row = wave; row < QT * 16; row += ENC_WAVES) {
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
                    v[pass][e] = (v[pass][e] == -INFINITY) ? 0.f : __expf(v[pass][e] - mx);  // v_exp_f32(x log2 e): the launch is VALU-bound at many streams (-8 %)
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
    ENC_STAMP(3);

    // ---- 3. O = P V : wave w owns dims 16(w&3) .. +15 and the key steps of half (w>>2) ----
    const int dt = wave & 3, half = wave >> 2;
    f32x4_t o[QT];
#pragma unroll
    for (int mt = 0; mt < QT; ++mt) o[mt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int dim = dt * 16 + fr;
    const bf16_t* vrow = vt + (long)dim * cap + fq * 8;  // B[k = 8 fq + j][col = fr] = V^T[dim][slot]
    const int k_steps = cap >> 5, k_half = (k_steps + 1) >> 1;
    const int ks_lo = half * k_half, ks_hi = min(ks_lo + k_half, k_steps);
    // physical slot range of the new keys: [nlo, nlo + Q) mod cap
    int nlo = start + len;
    if (nlo >= cap) nlo -= cap;
    constexpr int VB = QT == 1 ? 10 : 4;  // V^T fragments in flight before the first use (one stream: the whole half of a full window in ONE round trip instead of three)
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += VB) {
        u32x4_t vf[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int ks = ks0 + u < ks_hi ? ks0 + u : ks_hi - 1;
            const int t0 = ks * 32 + fq * 8;  // first of this lane's 8 slots
            int rel = t0 - nlo;
            if (rel < 0) rel += cap;
            const bool any_new = rel < Q || rel + 7 >= cap;  // the 8 slots touch [nlo, nlo+Q) (possibly wrapping)
            if (!any_new) {
                vf[u] = *reinterpret_cast<const u32x4_t*>(vrow + ks * 32);
            } else {  // mixed fragment: new keys come from the qkv rows
                bf16_t e[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    int r2 = t0 + q - nlo;
                    if (r2 < 0) r2 += cap;
                    e[q] = (r2 < Q) ? vnew[(long)r2 * 3 * D + dim] : vt[(long)dim * cap + t0 + q];
                }
                vf[u].x = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
                vf[u].y = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
                vf[u].z = (uint32_t)e[4] | ((uint32_t)e[5] << 16);
                vf[u].w = (uint32_t)e[6] | ((uint32_t)e[7] << 16);
            }
        }
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            if (ks0 + u < ks_hi) {
#pragma unroll
                for (int mt = 0; mt < QT; ++mt) {
                    const u32x4_t pf = *reinterpret_cast<const u32x4_t*>(S + (long)(mt * 16 + fr) * ldS + (ks0 + u) * 32 + fq * 8);
                    o[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pf), __builtin_bit_cast(bf16x8_t, vf[u]), o[mt], 0, 0, 0);
                }
            }
        }
    }
#ifdef ISST_ENC_TRACE
    asm volatile("s_nop 0" :: "v"(o[0][0]));
#endif
    ENC_STAMP(4);
    if (half == 1) {
#pragma unroll
        for (int mt = 0; mt < QT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Ohalf[(mt * 16 + fq * 4 + r) * ENC_HD + dim] = o[mt][r];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int mt = 0; mt < QT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + mt * 16 + fq * 4 + r;
                out[((long)s * Q + qi) * D + h * ENC_HD + dim] = f2bf(o[mt][r] + Ohalf[(mt * 16 + fq * 4 + r) * ENC_HD + dim]);
            }
    }
    ENC_STAMP(5);
    // ---- append V^T of the chunk's own keys (query block 0): [dim][slot] <- V[new frame][dim] ----
    if (qb == 0) {
        for (int e = tid; e < Q * ENC_HD; e += ENC_WAVES * 64) {
            const int i = e / ENC_HD, dd = e % ENC_HD;
            int slot = nlo + i;
            if (slot >= cap) slot -= cap;
            vt[(long)dd * cap + slot] = vnew[(long)i * 3 * D + dd];
        }
    }
    ENC_STAMP(6);
}

int launch_enc_attention(const bf16_t* qkv, bf16_t* kring, bf16_t* vring, long stream_stride,
                         const EncStreamView* sv, const float* rope_cos, const float* rope_sin, int rope_round_each,
                         bf16_t* out, int n_streams, int Q, int heads, int cap, int max_cache, int blocksize, hipStream_t s) {
    if (Q <= 0 || n_streams <= 0) return ISST_OK;
    if (Q % 16 != 0 || cap % 64 != 0 || cap > 1024 || max_cache + Q > cap) return ISST_ERR_ARG;
    const int QT = (Q % 48 == 0 && n_streams * (Q / 16) * heads >= 1024) ? 3 : 1;  // few streams: 16-row query blocks give 3x the workgroups (the K rotation is redone per block)
    const size_t lds = (size_t)QT * 16 * (cap + ENC_SPAD) * 2 + (size_t)QT * 16 * ENC_HD * sizeof(float);
    dim3 grid(heads, Q / (QT * 16), n_streams), block(ENC_WAVES * 64);
    auto go = [&](auto kern, size_t& lds_set) -> int {
        if (lds > 64 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
            lds_set = lds;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, qkv, kring, vring, stream_stride, sv, rope_cos, rope_sin, rope_round_each, out, Q, heads, cap, max_cache, blocksize);
        return ISST_OK;
    };
    static size_t lds_set[4] = {0, 0, 0, 0};
    int rc;
    if (QT == 3) rc = rope_round_each ? go(enc_attention_kernel<3, true>, lds_set[0]) : go(enc_attention_kernel<3, false>, lds_set[1]);
    else rc = rope_round_each ? go(enc_attention_kernel<1, true>, lds_set[2]) : go(enc_attention_kernel<1, false>, lds_set[3]);
    if (rc != ISST_OK) return rc;
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
