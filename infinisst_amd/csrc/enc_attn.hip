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
#include <cstdlib>
#include <type_traits>

#define ENC_HD 64
#define ENC_SPAD 8  // bf16 elements of padding per S row (keeps rows 16-byte aligned, shifts banks by 4 per row)
#define ENC_WAVES 8

// interleaved-pair rotation of 8 consecutive dims (4 pairs) [3P rotary_embedding_torch semantics, see oracle]; returns the four rotated pairs packed as bf16.
// round_each ("bf16" mode, the reference's production numerics): every product and every sum is a bf16 number.  Written word by word: the two products of an output
// element are rounded by ONE v_cvt_pk_bf16_f32 and come back as fp32 by a shift and a mask, and the sums of a pair are rounded by the conversion that packs them
// -- rounding twice (bfr, then pack8) gave the same bits for ~16 instead of ~10 instructions per pair in a phase that is bound by instruction issue.
__device__ __forceinline__ u32x4_t rot8(const float* x, const float* c, const float* sn, int round_each) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        if (round_each) {
            const uint32_t t = pack_bf(a * c[i], -b * sn[i]), u = pack_bf(b * c[i], a * sn[i]);
            w[i] = pack_bf(lo_bf(t) + hi_bf(t), lo_bf(u) + hi_bf(u));
        } else {
            w[i] = pack_bf(a * c[i] - b * sn[i], b * c[i] + a * sn[i]);
        }
    }
    return (u32x4_t){w[0], w[1], w[2], w[3]};
}
// The rotary table entries of one (position, 8-dim group) as they travel in registers.
//   TB = false: fp32 tables [pos][32] cos and sin, as handed to isst_set_rope_tables: 2 x 16 B per group.
//   TB = true:  ONE packed bf16 table [pos][k-step 2][fq 4][cos x 4 | sin x 4] (16 B per group) -- built by the library when every table value IS a bf16 number,
//               which is the reference's production setting (the encoder is cast to bf16, so its rotary module computes cos / sin in bf16: rope.py "bf16" mode).
//               Same values, half the bytes and a third fewer load instructions per key tile, 8 instead of 16 registers per tile in flight.
template <bool TB> struct EncTab;
template <> struct EncTab<false> { f32x4_t c, s; };
template <> struct EncTab<true> { u32x4_t cs; };
template <bool TB>
__device__ __forceinline__ void enc_tab_load(EncTab<TB>& t, const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const bf16_t* __restrict__ rope_cs,
                                             int pos, int ks, int fq) {
    if constexpr (TB) {
        t.cs = *reinterpret_cast<const u32x4_t*>(rope_cs + ((long)pos * 8 + ks * 4 + fq) * 8);
    } else {
        t.c = *reinterpret_cast<const f32x4_t*>(rope_cos + (long)pos * 32 + ks * 16 + fq * 4);
        t.s = *reinterpret_cast<const f32x4_t*>(rope_sin + (long)pos * 32 + ks * 16 + fq * 4);
    }
}
// raw 8 dims (the lane's 16 bytes of a q or k row) rotated with the table entries `t`, as an MFMA operand fragment
template <bool TB>
__device__ __forceinline__ u32x4_t rot_frag_regs(const u32x4_t& raw, const EncTab<TB>& t, int round_each) {
    float x[8], cc[4], ss[4];
    unpack8(raw, x);
    if constexpr (TB) {
        cc[0] = lo_bf(t.cs.x); cc[1] = hi_bf(t.cs.x); cc[2] = lo_bf(t.cs.y); cc[3] = hi_bf(t.cs.y);
        ss[0] = lo_bf(t.cs.z); ss[1] = hi_bf(t.cs.z); ss[2] = lo_bf(t.cs.w); ss[3] = hi_bf(t.cs.w);
    } else {
        cc[0] = t.c.x; cc[1] = t.c.y; cc[2] = t.c.z; cc[3] = t.c.w;
        ss[0] = t.s.x; ss[1] = t.s.y; ss[2] = t.s.z; ss[3] = t.s.w;
    }
    return rot8(x, cc, ss, round_each);
}
// 8 dims (k-step ks, group fq) of the QUERY row at `p`, rotated at position `pos` and scaled by head_dim^-0.5 = 1/8.  The reference scales q before the
// product (patch_speech_encoder.py:768); a power of two commutes with every rounding on the way (bf16 and fp32 share their exponent range), so whether the
// rotated query, the products or the accumulated score is scaled gives the same bits -- scaled here, once per query, the 12 scores a lane holds per key tile
// need no multiplication
template <bool TB>
__device__ __forceinline__ u32x4_t rot_frag(const bf16_t* p, int pos, int ks, int fq, const float* __restrict__ rope_cos, const float* __restrict__ rope_sin,
                                            const bf16_t* __restrict__ rope_cs, int round_each) {
    EncTab<TB> t;
    enc_tab_load<TB>(t, rope_cos, rope_sin, rope_cs, pos, ks, fq);
    const u32x4_t r = rot_frag_regs<TB>(*reinterpret_cast<const u32x4_t*>(p + ks * 32 + fq * 8), t, round_each);
    float y[8];
    unpack8(r, y);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] *= 0.125f;
    return pack8(y);
}

// Scores of ONE key tile (16 keys: this lane's key is column fr) against the workgroup's QT m-tiles of queries: rotation of the lane's 2 x 8 key dims, two MFMAs per
// m-tile, then  bf16(acc / 8)  or -inf by the mask  into S[row][cphys].  `ok(mt, r)`: may row 4 fq + r of m-tile mt see this lane's key?
// `qfrag(mt, ks)`: the rotated query fragment (registers at one stream, LDS at many: see the kernel).
template <int QT, bool RND, bool TB, typename QFrag, typename Mask>
__device__ __forceinline__ void enc_score_tile(const u32x4_t (&kraw)[2], const EncTab<TB> (&kt)[2], QFrag qfrag,
                                               bf16_t* S, int ldS, int cphys, int fq, Mask ok) {
    u32x4_t kf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) kf[ks] = rot_frag_regs<TB>(kraw[ks], kt[ks], RND ? 1 : 0);
#pragma unroll
    for (int mt = 0; mt < QT; ++mt) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, qfrag(mt, ks)), __builtin_bit_cast(bf16x8_t, kf[ks]), acc, 0, 0, 0);
        // C layout: this lane holds column fr (its own key), rows 4 fq + r.  Masked scores become -inf BEFORE the conversion, two scores per v_cvt_pk_bf16_f32
        // (bf16(-inf) = 0xFF80); the query carries the 1/8 (rot_frag)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const uint32_t w = pack_bf(ok(mt, r) ? acc[r] : -INFINITY, ok(mt, r + 1) ? acc[r + 1] : -INFINITY);
            S[(long)(mt * 16 + fq * 4 + r) * ldS + cphys] = (bf16_t)(w & 0xffffu);
            S[(long)(mt * 16 + fq * 4 + r + 1) * ldS + cphys] = (bf16_t)(w >> 16);
        }
    }
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

// Softmax of one row of S in place (phase 2) for rings of 512 + 128 TD slots: lane l owns columns 8 l .. 8 l + 7 and 512 + 2 TD l .. + 2 TD - 1 -- one 16-byte access and
// TD dwords, no lane predicate, no branch (the general form below tests `lane < n8` and `e < te` per element: ~350 instructions per row against ~130 here, in a phase
// that is bound by what the SIMDs issue).  Same arithmetic and the same summation order as the general form.
template <int TD>
__device__ __forceinline__ void enc_softmax_row(bf16_t* srow, int lane) {
    float v[8 + 2 * TD];
    unpack8(*reinterpret_cast<const u32x4_t*>(srow + lane * 8), v);
    uint32_t* tw = reinterpret_cast<uint32_t*>(srow + 512) + lane * TD;
#pragma unroll
    for (int d = 0; d < TD; ++d) { const uint32_t w = tw[d]; v[8 + 2 * d] = lo_bf(w); v[9 + 2 * d] = hi_bf(w); }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8 + 2 * TD; ++e) mx = fmaxf(mx, v[e]);
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8 + 2 * TD; ++e) {
        v[e] = __expf(v[e] - mx);  // v_exp_f32(x log2 e); a masked score is -inf and gives exactly 0 (every row sees its own key: mx is finite)
        sum += v[e];
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int e = 0; e < 8 + 2 * TD; ++e) v[e] *= inv;
    *reinterpret_cast<u32x4_t*>(srow + lane * 8) = pack8(v);
#pragma unroll
    for (int d = 0; d < TD; ++d) tw[d] = pack_bf(v[8 + 2 * d], v[9 + 2 * d]);
}

template <int QT, bool RND, bool TB>  // TB: the packed bf16 rotary table (EncTab); QT: m-tiles of 16 query rows per workgroup; RND: every product of the rotation rounds to bf16 (enc_rope_mode "bf16": as a compile-time
                              // constant -- as a runtime flag every rotated pair carried a branch, 160 of them per key tile in a phase that is bound by instruction issue)
__global__ __launch_bounds__(512, QT == 1 ? 2 : 4) void enc_attention_kernel(  // (48-row blocks run two workgroups per CU: 4 waves per SIMD, 128 registers -- stated, not left to luck)
    const bf16_t* __restrict__ qkv, bf16_t* kring, bf16_t* vring, long stream_stride,
                                                            const EncStreamView* __restrict__ sv,
                                                            const float* __restrict__ rope_cos, const float* __restrict__ rope_sin, const bf16_t* __restrict__ rope_cs,
                                                            int /*round_each: RND*/, bf16_t* __restrict__ out, int Q, int heads, int cap,
                                                            int C, int bs, int vn_off) {
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

    // ---- the chunk's own V rows [Q][64] into LDS (vn_off >= 0: the launcher found room): 16-byte loads of whole rows, issued before anything else.  Phase 3 needs
    //      them TRANSPOSED (8 keys of one dim) wherever a fragment of the V^T ring touches slots this chunk is only now filling, and so does the append to the ring:
    //      both were 2-byte global loads (8 per lane and mixed fragment, each fragment a round trip of its own behind a divergent branch) ----
    const bool use_vn = QT == 1 && vn_off >= 0;  // (48-row blocks never stage: the launcher says why)
    bf16_t* Vn = reinterpret_cast<bf16_t*>(smem + (use_vn ? vn_off : 0));
    if (use_vn)
        for (int e = tid; e < Q * 8; e += ENC_WAVES * 64)
            *reinterpret_cast<u32x4_t*>(Vn + (e >> 3) * ENC_HD + (e & 7) * 8) = *reinterpret_cast<const u32x4_t*>(vnew + (long)(e >> 3) * 3 * D + (e & 7) * 8);

    // ---- append the chunk's own keys, unrotated, to the ring (query block 0): one 16-byte load + store per thread, before anything else.  Nobody reads these
    //      slots in this launch (every workgroup takes the chunk's keys from the qkv rows), so it does not matter when the stores land; inside the tile loop of
    //      phase 1 a conditional store made hipcc drain vmcnt(0) at every tile, which took the tile in flight down with it ----
    // physical slot range of the chunk's own keys: [nlo, nlo + Q) mod cap
    int nlo = start + len;
    if (nlo >= cap) nlo -= cap;
    // (Q <= 64: one 16-byte piece per thread, LOADED here and stored behind the query rotation -- a store right here would make its wave wait a whole round trip
    //  before it even asks for its query rows)
    const bool k_split = Q * 8 <= ENC_WAVES * 64;
    const bool k_mine = qb == 0 && k_split && tid < Q * 8;
    u32x4_t k_val = {0u, 0u, 0u, 0u};
    bf16_t* k_dst = kr;
    if (k_mine) {
        int slot = nlo + (tid >> 3);
        if (slot >= cap) slot -= cap;
        k_dst = kr + (long)slot * ENC_HD + (tid & 7) * 8;
        k_val = *reinterpret_cast<const u32x4_t*>(knew + (long)(tid >> 3) * 3 * D + (tid & 7) * 8);
    }
    if (qb == 0 && !k_split) {
        for (int e = tid; e < Q * 8; e += ENC_WAVES * 64) {
            int slot = nlo + (e >> 3);
            if (slot >= cap) slot -= cap;
            *reinterpret_cast<u32x4_t*>(kr + (long)slot * ENC_HD + (e & 7) * 8) = *reinterpret_cast<const u32x4_t*>(knew + (long)(e >> 3) * 3 * D + (e & 7) * 8);
        }
    }
    // ---- ... and V^T [dim][slot] <- V[new frame][dim].  A thread owns (dim, one ALIGNED group of 8 ring slots that the chunk's keys touch): it reads the group
    //      (16 B), replaces the slots of the new keys -- eight 2-byte loads at most; a wave's lanes are 64 consecutive dims of one frame: one line per load -- and
    //      stores the group back (slots of older keys get their own bytes again: this workgroup is the ring's only writer in the launch).  Rounds 1-5 stored element
    //      by element: 3072 two-byte stores per workgroup, every lane of a wave into a line of its own.
    //      Many streams (48-row blocks) whose workgroup is the only one of its (head, stream) do it HERE: phase 3, three barriers on, then reads every fragment of
    //      V^T from the ring -- no fragment mixed from ring and qkv rows (8 two-byte loads per lane behind a divergent branch) -- and nobody else reads these slots.
    const bool v_first = QT > 1 && gridDim.y == 1 && vn_off != -2;  // (-2: A/B knob ISST_ENC_VFIRST=0)
    auto append_vt8 = [&]() {
        const int a0 = nlo & ~7, ng = ((nlo & 7) + Q + 7) >> 3;
        for (int e = tid; e < ng * ENC_HD; e += ENC_WAVES * 64) {
            const int dd = e % ENC_HD;
            int base = a0 + (e / ENC_HD) * 8;
            if (base >= cap) base -= cap;  // (cap is a multiple of 8: an aligned group does not wrap)
            bf16_t* gp = vt + (long)dd * cap + base;
            const u32x4_t old = *reinterpret_cast<const u32x4_t*>(gp);
            uint32_t w[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                int rel = base + q - nlo;
                if (rel < 0) rel += cap;
                if (rel < Q) {
                    const uint32_t nv = vnew[(long)rel * 3 * D + dd];
                    w[q >> 1] = (q & 1) ? ((w[q >> 1] & 0x0000ffffu) | (nv << 16)) : ((w[q >> 1] & 0xffff0000u) | nv);
                }
            }
            *reinterpret_cast<u32x4_t*>(gp) = (u32x4_t){w[0], w[1], w[2], w[3]};
        }
    };
    // (the common shape -- at most 512 (dim, group) pairs: Q = 48 -- in two halves: the loads here, the patched group's store behind the query rotation, so that
    //  no wave sits waiting for these loads; the barrier that follows the rotation does not wait for the stores)
    const int v_ng = ((nlo & 7) + Q + 7) >> 3;
    const bool v_split = v_first && v_ng * ENC_HD <= ENC_WAVES * 64;
    if (v_first && !v_split) append_vt8();
    u32x4_t v_old = {0u, 0u, 0u, 0u};
    uint32_t v_nv[8];
    int v_rel0 = 0;
    bf16_t* v_gp = vt;
    const bool v_mine = v_split && tid < v_ng * ENC_HD;
    if (v_mine) {
        const int dd = tid % ENC_HD;
        int base = (nlo & ~7) + (tid / ENC_HD) * 8;
        if (base >= cap) base -= cap;
        v_gp = vt + (long)dd * cap + base;
        v_old = *reinterpret_cast<const u32x4_t*>(v_gp);
        v_rel0 = base - nlo;
        if (v_rel0 < -7) v_rel0 += cap;  // slot q of the group is new key v_rel0 + q where that lies in [0, Q) (v_rel0 in -7 .. Q - 1)
#pragma unroll
        for (int q = 0; q < 8; ++q) v_nv[q] = vnew[(long)min(max(v_rel0 + q, 0), Q - 1) * 3 * D + dd];  // (unconditional loads; the unused ones are dropped below)
    }

    // ---- rotated query fragments: A[row = fr][k = 8 fq + j] for both 32-dim k-steps ----
    // One stream (QT == 1): both fragments in every wave's registers, no barrier.  Many streams (QT == 3): six fragments -- 24 registers in each of 8 waves, rotated
    // eight times over -- are rotated ONCE (wave w < 2 QT takes fragment w) into LDS (the bytes Ohalf uses after phase 1) and re-read per key tile: the registers
    // are what the second key tile in flight needs (phase 1 below) at two workgroups per CU.
    constexpr bool QLDS = QT > 1;
    u32x4_t* Qs = reinterpret_cast<u32x4_t*>(Ohalf);  // [QT * 2][64 lanes] x 16 B
    static_assert(QT * 2 * 64 * 16 <= QT * 16 * ENC_HD * 4, "the query fragments fit the bytes of Ohalf");
    u32x4_t qf[QLDS ? 1 : QT][2];
    if constexpr (QLDS) {
        if (wave < QT * 2) {
            const int mt = wave >> 1, ks = wave & 1;
            const int qi = q0 + mt * 16 + fr;
            const bf16_t* qrow = qkv + ((long)s * Q + qi) * 3 * D + h * ENC_HD;
            Qs[wave * 64 + lane] = rot_frag<TB>(qrow, K - Q + qi, ks, fq, rope_cos, rope_sin, rope_cs, RND ? 1 : 0);
        }
        if (k_mine) *reinterpret_cast<u32x4_t*>(k_dst) = k_val;
        if (v_mine) {
            uint32_t w[4] = {v_old.x, v_old.y, v_old.z, v_old.w};
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if ((unsigned)(v_rel0 + q) < (unsigned)Q) w[q >> 1] = (q & 1) ? ((w[q >> 1] & 0x0000ffffu) | (v_nv[q] << 16)) : ((w[q >> 1] & 0xffff0000u) | v_nv[q]);
            *reinterpret_cast<u32x4_t*>(v_gp) = (u32x4_t){w[0], w[1], w[2], w[3]};
        }
        // LDS-only barrier: the query fragments are in LDS for every wave.  (__syncthreads would also wait for the ring stores above -- the keys and V^T groups of
        // the chunk -- to be acknowledged: ~4 us at the head of every workgroup; nobody reads them before the barrier that ends phase 1, a full fence)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    } else {
#pragma unroll
        for (int mt = 0; mt < QT; ++mt) {
            const int qi = q0 + mt * 16 + fr;
            const bf16_t* qrow = qkv + ((long)s * Q + qi) * 3 * D + h * ENC_HD;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) qf[mt][ks] = rot_frag<TB>(qrow, K - Q + qi, ks, fq, rope_cos, rope_sin, rope_cs, RND ? 1 : 0);
        }
        if (k_mine) *reinterpret_cast<u32x4_t*>(k_dst) = k_val;
    }
    auto qfrag = [&](int mt, int ks) -> u32x4_t {
        if constexpr (QLDS) return Qs[(mt * 2 + ks) * 64 + lane];
        else return qf[mt][ks];
    };
#ifdef ISST_ENC_TRACE
    if constexpr (!QLDS) asm volatile("s_nop 0" :: "v"(qf[0][0].x), "v"(qf[0][1].x));
#endif
    ENC_STAMP(1);
    // ---- 1. scores ----
    // The mask needs no tensor: row qi may see logical columns [lo, hi) (patch_speech_encoder.py:30-77; P == 0 is the training mask):
    //   hi = min(end of the block of frame qi + P, P + Q) - off,   lo = max(0, qi + P - C) - off,   off = max(0, P - C).
    // lo in one form for both signs of P - C:  j >= lo  <=>  j - min(P - C, 0) - qi >= 0  (j >= 0 always).
    const int n_tiles = cap >> 4;
    // one key tile's loads: the lane's key row (2 x 16 B) and its rotary table entries (4 x 16 B); returns the key's logical index (>= K: outside the window)
    auto issue = [&](int nt, u32x4_t (&kraw)[2], EncTab<TB> (&kt)[2]) -> int {
        const int cphys = nt * 16 + fr;       // physical slot of this lane's key
        int j = cphys - start;                // logical index
        if (j < 0) j += cap;
        const bool live = j < K;
        const int jpos = live ? j : 0;
        // written by this chunk: still only in the qkv rows for all this launch knows
        const bool is_new = live && j >= len;
        // (one base + one 32-bit element offset, both chosen by selects: written as two pointer expressions hipcc put a divergent branch with two 64-bit
        //  multiplications into every tile)
        const bf16_t* kbase = is_new ? knew : kr;
        const int koff = is_new ? (j - len) * 3 * D : cphys * ENC_HD;
        const bf16_t* krow = kbase + koff;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            kraw[ks] = *reinterpret_cast<const u32x4_t*>(krow + ks * 32 + fq * 8);
            enc_tab_load<TB>(kt[ks], rope_cos, rope_sin, rope_cs, jpos, ks, fq);
        }
        return j;
    };
    const int a0 = q0 + P;
    const int blk0 = a0 / bs;
    // every row of the workgroup in ONE block of the mask (always, unless the block size changed mid-stream or a query block straddles two blocks): hi is one
    // number and lo a comparison against a constant -- no per-row bounds in registers, which is what pays for the second tile in flight below
    const bool one_block = QT > 1 && (a0 + QT * 16 - 1) / bs == blk0;
    if (one_block) {
        // Many streams (QT == 3, two workgroups per CU): a wave's tiles were a chain of [6 loads -> rotation -> MFMAs -> 12 stores], ~2.5 us each with nothing of its
        // own in flight behind the loads (13 of a workgroup's 37 us, profiles/r03/enc_attention_trace_64_streams.txt).  Two register sets: tile i + 1 is requested
        // before tile i is rotated.
        const int hi_s = min((blk0 + 1) * bs, P + Q) - off;      // (<= K: a key below hi is live)
        const int jb = min(P - C, 0) + q0 + fq * 4;              // row 4 fq + r of m-tile mt sees key j  <=>  j - jb >= 16 mt + r
        u32x4_t rawA[2], rawB[2];
        EncTab<TB> tA[2], tB[2];
        int jA = 0, jB = 0;
        // (every issue inside the loop is unconditional -- past the end the wave's last tile is re-read and not used --: a branch around loads makes hipcc wait
        //  with vmcnt(0), i.e. for the tile in flight too)
        // row 4 fq + r of m-tile mt sees key j  <=>  j < hi_s  and  j - jb >= 16 mt + r: ONE comparison per score against  jm = j < hi_s ? j - jb : INT_MIN
        // Loop shape: tiles in PAIRS with no branch between a tile's loads and its use, the odd last tile behind the loop.  (With an early exit between the issue of
        // tile i + 1 and its use hipcc sinks those loads below the exit test -- they are only needed on one side of it -- i.e. behind the rotation of tile i.)
        const int cnt = wave < n_tiles ? (n_tiles - wave + ENC_WAVES - 1) / ENC_WAVES : 0;  // this wave's tiles: wave, wave + 8, ..
        if (cnt > 0) {
            const int last = wave + (cnt - 1) * ENC_WAVES;
            auto score = [&](const u32x4_t (&raw)[2], const EncTab<TB> (&t)[2], int j, int nt) {
                const int jm = j < hi_s ? j - jb : (int)0x80000000;
                enc_score_tile<QT, RND, TB>(raw, t, qfrag, S, ldS, nt * 16 + fr, fq, [&](int mt, int r) { return jm >= mt * 16 + r; });
            };
#ifndef ENC_SCORE_DEPTH
#define ENC_SCORE_DEPTH 2   // register sets of key tiles: 2 = one tile in flight behind the one being rotated, 3 = two.  Measured equal (61.3 / 62.5 against
                            // 61.7 / 62.4 us per launch at 64 streams, same box, profiles/r06/enc_attention_rework_ab4_depth.txt): with one tile in flight
                            // the phase is no longer waiting for loads
#endif
            if constexpr (ENC_SCORE_DEPTH == 3 && TB) {
                u32x4_t rawC[2];
                EncTab<TB> tC[2];
                int jC = 0;
                jA = issue(wave, rawA, tA);
                jB = issue(min(wave + ENC_WAVES, last), rawB, tB);
                __builtin_amdgcn_sched_barrier(0);
                int nt = wave, i = 0;
                for (; i + 2 < cnt; i += 3, nt += 3 * ENC_WAVES) {  // TRIPLES: no branch between a tile's loads and its use
                    jC = issue(nt + 2 * ENC_WAVES, rawC, tC);
                    __builtin_amdgcn_sched_barrier(0);  // (program order pinned: left alone the scheduler moves the loads behind the rotation below, to shorten their live ranges)
                    score(rawA, tA, jA, nt);
                    jA = issue(min(nt + 3 * ENC_WAVES, last), rawA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    score(rawB, tB, jB, nt + ENC_WAVES);
                    jB = issue(min(nt + 4 * ENC_WAVES, last), rawB, tB);
                    __builtin_amdgcn_sched_barrier(0);
                    score(rawC, tC, jC, nt + 2 * ENC_WAVES);
                }
                if (i < cnt) score(rawA, tA, jA, nt);
                if (i + 1 < cnt) score(rawB, tB, jB, nt + ENC_WAVES);
            } else {
                jA = issue(wave, rawA, tA);
                int nt = wave;
                for (int i = 0; i + 1 < cnt; i += 2, nt += 2 * ENC_WAVES) {
                    jB = issue(nt + ENC_WAVES, rawB, tB);
                    __builtin_amdgcn_sched_barrier(0);  // (program order pinned: left alone the scheduler moves these loads behind the rotation below, to shorten their live ranges)
                    score(rawA, tA, jA, nt);
                    jA = issue(min(nt + 2 * ENC_WAVES, last), rawA, tA);
                    __builtin_amdgcn_sched_barrier(0);
                    score(rawB, tB, jB, nt + ENC_WAVES);
                }
                if (cnt & 1) score(rawA, tA, jA, last);
            }
        }
    } else {
        // visible logical column range of the rows this lane holds in the C layout (rows 4 fq + r of each m-tile)
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
        // One stream (QT == 1: 48 workgroups on 256 CUs, nothing else to hide a round trip behind) walks its 5 key tiles per wave with ALL their loads -- key rows and
        // rotary table entries -- in flight before the first rotation: the phase was a chain of 5 dependent round trips, 9.0 of the launch's 18.4 us
        // (profiles/r05/enc_attention_trace_1_stream.txt).
        // (16 waves x 3 tiles instead of 8 x 5 for the lone stream: the phase took 10.9 us instead of 9.0 -- it is bound by the instructions a SIMD has to issue at the clock
        //  the chip holds, not by latency: profiles/r05/enc_attention_trace_1_stream_16_waves_SLOWER.txt)
        constexpr int TCH = QT == 1 ? 5 : 1;
        for (int nt0 = wave; nt0 < n_tiles; nt0 += ENC_WAVES * TCH) {
            u32x4_t kraw[TCH][2];
            EncTab<TB> kt[TCH][2];
            int jj[TCH];
#pragma unroll
            for (int t = 0; t < TCH; ++t) jj[t] = issue(min(nt0 + t * ENC_WAVES, n_tiles - 1), kraw[t], kt[t]);  // (a tile past the end re-reads the last one; it is not used)
#pragma unroll
            for (int t = 0; t < TCH; ++t) {
                const int nt = nt0 + t * ENC_WAVES;
                if (nt >= n_tiles) break;  // (wave-uniform)
                const int j = jj[t];
                enc_score_tile<QT, RND, TB>(kraw[t], kt[t], qfrag, S, ldS, nt * 16 + fr, fq, [&](int mt, int r) { return (j >= lo[mt][r]) & (j < hi[mt][r]); });  // (hi <= K: a key below hi is live; & not &&: no branches)
            }
        }
    }
    __syncthreads();
    ENC_STAMP(2);
    // ---- append V^T of the chunk's own keys (query block 0): [dim][slot] <- V[new frame][dim], from the staged rows; the stores drain under phases 2 and 3
    //      (phase 3 patches every fragment that touches these slots from LDS, so it does not matter to anyone when they land) ----
    if (qb == 0 && use_vn) {
        for (int e = tid; e < Q * ENC_HD; e += ENC_WAVES * 64) {
            int slot = nlo + e / ENC_HD;
            if (slot >= cap) slot -= cap;
            vt[(long)(e % ENC_HD) * cap + slot] = Vn[e];
        }
    }

    // ---- 2. softmax per row (fp32), probabilities rounded to bf16 in place ----
    // A lane takes 8 consecutive columns of the first 512 (one 16-byte access) and `te` = (cap - 512) / 64 consecutive columns of the rest: a 640-slot ring is 10
    // elements per lane.  (Rounds 1-5 ran a second 8-column pass for the columns past 512 with 16 of 64 lanes active: 16 element-slots of exp / compare / multiply
    // per row for 10 elements of work, in a phase that is VALU-bound at many streams.)
    const int n8 = min(cap, 512) >> 3;               // lanes of the 16-byte pass
    const int te = cap > 512 ? (cap - 512) >> 6 : 0;  // 0..8 tail columns per lane, from column 512 + lane * te
    if (cap >= 512 && (te & 1) == 0) {  // (ldS = cap + 8: rows and their column 512 are 16-byte aligned)
        auto rows = [&](auto td) {
            for (int row = wave; row < QT * 16; row += ENC_WAVES) enc_softmax_row<decltype(td)::value>(S + (long)row * ldS, lane);
        };
        switch (te >> 1) {
            case 0: rows(std::integral_constant<int, 0>{}); break;
            case 1: rows(std::integral_constant<int, 1>{}); break;  // 640 slots: max_cache_size 576 + one block of 48
            case 2: rows(std::integral_constant<int, 2>{}); break;
            case 3: rows(std::integral_constant<int, 3>{}); break;
            default: rows(std::integral_constant<int, 4>{}); break;
        }
    } else
    for (int row = wave; row < QT * 16; row += ENC_WAVES) {
        bf16_t* srow = S + (long)row * ldS;
        bf16_t* trow = srow + 512 + lane * te;
        float v[8], tv[8];
        float mx = -INFINITY;
        if (lane < n8) {
            unpack8(*reinterpret_cast<const u32x4_t*>(srow + lane * 8), v);
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, v[e]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < te) { tv[e] = bf2f(trow[e]); mx = fmaxf(mx, tv[e]); }
        mx = wave_max(mx);
        float sum = 0.f;
        if (lane < n8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = __expf(v[e] - mx);  // v_exp_f32(x log2 e); a masked score is -inf and gives exactly 0 (every row sees its own key: mx is finite)
                sum += v[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < te) { tv[e] = __expf(tv[e] - mx); sum += tv[e]; }
        sum = wave_sum(sum);
        const float inv = 1.0f / sum;
        if (lane < n8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= inv;
            *reinterpret_cast<u32x4_t*>(srow + lane * 8) = pack8(v);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (e < te) trow[e] = f2bf(tv[e] * inv);
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
    constexpr int VB = 10;  // V^T fragments in flight before the first use: the whole half of a full 640-slot window in ONE round trip instead of three (the registers of phase 1 are free by now)
    for (int ks0 = ks_lo; ks0 < ks_hi; ks0 += VB) {
        u32x4_t vf[VB];
#pragma unroll
        for (int u = 0; u < VB; ++u) {
            const int ks = ks0 + u < ks_hi ? ks0 + u : ks_hi - 1;
            const int t0 = ks * 32 + fq * 8;  // first of this lane's 8 slots
            int rel = t0 - nlo;
            if (rel < 0) rel += cap;
            const bool any_new = rel < Q || rel + 7 >= cap;  // the 8 slots touch [nlo, nlo+Q) (possibly wrapping)
            if (!any_new || use_vn || v_first) {
                vf[u] = *reinterpret_cast<const u32x4_t*>(vrow + ks * 32);  // (staged rows: the new slots' stale bytes are replaced below)
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
        if (use_vn) {  // fragments that touch the slots this chunk is filling take those keys from the staged rows
#pragma unroll
            for (int u = 0; u < VB; ++u) {
                const int t0 = (ks0 + u < ks_hi ? ks0 + u : ks_hi - 1) * 32 + fq * 8;
                int rel = t0 - nlo;
                if (rel < 0) rel += cap;
                if (rel < Q || rel + 7 >= cap) {
                    uint32_t w[4] = {vf[u].x, vf[u].y, vf[u].z, vf[u].w};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        int r2 = rel + q;
                        if (r2 >= cap) r2 -= cap;
                        if (r2 < Q) {
                            const uint32_t nv = Vn[r2 * ENC_HD + dim];
                            w[q >> 1] = (q & 1) ? ((w[q >> 1] & 0x0000ffffu) | (nv << 16)) : ((w[q >> 1] & 0xffff0000u) | nv);
                        }
                    }
                    vf[u] = (u32x4_t){w[0], w[1], w[2], w[3]};
                }
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
    // ---- append V^T of the chunk's own keys (query block 0) where it has not happened yet ----
    if (qb == 0 && !use_vn && !v_first) append_vt8();
    ENC_STAMP(6);
}

int launch_enc_attention(const bf16_t* qkv, bf16_t* kring, bf16_t* vring, long stream_stride,
                         const EncStreamView* sv, const float* rope_cos, const float* rope_sin, int rope_round_each,
                         bf16_t* out, int n_streams, int Q, int heads, int cap, int max_cache, int blocksize, hipStream_t s, const bf16_t* rope_cs) {
    if (Q <= 0 || n_streams <= 0) return ISST_OK;
    if (Q % 16 != 0 || cap % 64 != 0 || cap > 1024 || max_cache + Q > cap) return ISST_ERR_ARG;
    const int QT = (Q % 48 == 0 && n_streams * (Q / 16) * heads >= 1024) ? 3 : 1;  // few streams: 16-row query blocks give 3x the workgroups (the K rotation is redone per block)
    size_t lds = (size_t)QT * 16 * (cap + ENC_SPAD) * 2 + (size_t)QT * 16 * ENC_HD * sizeof(float);
    // the chunk's own V rows staged in LDS (Q x 128 B): few streams only (16-row query blocks).  Measured on one box (profiles/r06/enc_attention_rework_ab.txt):
    // one stream 20.9 -> 19.7 us per launch with the rows staged; 64 streams (48-row blocks, two workgroups per CU, bound by what the SIMDs issue, not by round
    // trips) 81.1 -> 85.8 us -- there the patching of the fragments costs more than the few lanes of 2-byte loads it replaces, so those launches keep the global path
    int vn_off = -1;
    static const bool vn_on = !(getenv("ISST_ENC_VN") && atoi(getenv("ISST_ENC_VN")) == 0);  // A/B knob: 0 = the rows stay in global memory (rounds 1-5)
    if (vn_on && QT == 1 && lds + (size_t)Q * 128 <= 160 * 1024) { vn_off = (int)lds; lds += (size_t)Q * 128; }
    static const bool vfirst_on = !(getenv("ISST_ENC_VFIRST") && atoi(getenv("ISST_ENC_VFIRST")) == 0);  // A/B knob: 0 = 48-row blocks append V^T at the end and mix fragments
    if (QT == 3 && !vfirst_on) vn_off = -2;
    dim3 grid(heads, Q / (QT * 16), n_streams), block(ENC_WAVES * 64);
    auto go = [&](auto kern, size_t& lds_set) -> int {
        if (lds > 64 * 1024 && lds > lds_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
            lds_set = lds;
        }
        hipLaunchKernelGGL(kern, grid, block, lds, s, qkv, kring, vring, stream_stride, sv, rope_cos, rope_sin, rope_cs, rope_round_each, out, Q, heads, cap, max_cache, blocksize, vn_off);
        return ISST_OK;
    };
    static size_t lds_set[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc;
    if (rope_cs) {
        if (QT == 3) rc = rope_round_each ? go(enc_attention_kernel<3, true, true>, lds_set[4]) : go(enc_attention_kernel<3, false, true>, lds_set[5]);
        else rc = rope_round_each ? go(enc_attention_kernel<1, true, true>, lds_set[6]) : go(enc_attention_kernel<1, false, true>, lds_set[7]);
    } else {
        if (QT == 3) rc = rope_round_each ? go(enc_attention_kernel<3, true, false>, lds_set[0]) : go(enc_attention_kernel<3, false, false>, lds_set[1]);
        else rc = rope_round_each ? go(enc_attention_kernel<1, true, false>, lds_set[2]) : go(enc_attention_kernel<1, false, false>, lds_set[3]);
    }
    if (rc != ISST_OK) return rc;
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
