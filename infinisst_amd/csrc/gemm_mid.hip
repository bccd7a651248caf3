// Packed-weight GEMM for 13..64 rows on gfx950:  out[M,N] = epi(A[M,K] @ W[N,K]^T), weights read exactly once.
//
// Where it runs: the LLM prefill of one stream (22-66 prompt rows, reference model/llm.py:86-113 -> LlamaDecoderLayer) and
// every decode pass of a 64-stream batch (64 rows, one per stream).  These are still weight-streaming problems (HBM-bound),
// but unlike the 1..16-row case (gemm.hip) the ACTIVATIONS dominate the on-chip traffic: with 16-column workgroups every
// workgroup re-reads all of A through L2 in fragment-shaped 64-byte pieces, 4x the weight bytes at 64 rows (measured: the
// skinny kernel's time tracks A + W bytes at ~8 TB/s of L2->CU traffic, not W bytes at HBM rate).  So here
//   * a workgroup (4 or 8 waves) owns 2 or 4 n-tiles (32 or 64 columns) and ALL rows: A is staged through LDS in full 128-byte lines
//     (8 rows x 128 B per wave-load, cdna_hip_programming.md section 4 "x operand through LDS in full lines"), written as
//     ready-made MFMA A fragments ([k-step][m-tile][lane] x 16 B: conflict-free ds_write_b128 and ds_read_b128), double
//     buffered, one barrier per 256-deep K chunk; every wave reads the same staged fragments;
//   * a wave owns one PAIR of n-tiles and one of the chunk's four k-steps, so each A fragment it reads from LDS feeds two
//     MFMAs (LDS bandwidth, not HBM, bounds the skeleton of this kernel: measured with emptied descriptors); the weight
//     fragments stream straight from the fragment-major packed layout into VGPRs through bounds-checked buffer loads, two
//     chunks in flight, non-temporal (read once);
//   * the K halves meet in LDS (fixed order: deterministic) and the epilogue applies bias / GELU / residual / SwiGLU with
//     the reference's bf16 rounding points, exactly as gemm.hip;
//   * narrow outputs (o_proj, down_proj: N = 4096 -> 64-128 workgroups) additionally split K over workgroups
//     (EPI_PARTIAL): fp32 slabs [slice][M][N], reduced -- in fixed order -- by the residual + RMSNorm kernel that follows
//     anyway (rowops.hip rmsnorm_reduce_kernel), so the split costs no extra launch.
#include "common.h"

#define MID_KS 8                  // k-steps per chunk (KSW per wave: 4 k-step waves x 2, or -- the narrow forms -- 8 k-step waves x 1)
#define MID_CK (32 * MID_KS)      // K elements per staged chunk (256): one barrier per chunk, 4 * MT MFMAs per wave between barriers

// NP: n-tile pairs per workgroup (1, 2 or 4) -> 32, 64 or 128 columns, 4 * NP waves.  Wave (np, wk) owns the two n-tiles of pair np
// (for SwiGLU exactly one (gate, up) pair) and k-steps wk*KSW .. of every chunk: per chunk it reads its KSW * MT A fragments from
// LDS once and feeds 2 MFMAs on independent accumulators from each.
// KSW: k-steps per wave and chunk.  2: 4 k-step waves per n-tile pair.  1: 8 of them -- twice the waves on the same tile.  A 32- or 64-column
// workgroup of the KSW = 2 form is 4 or 8 waves, and the narrow projections give a CU one workgroup: one or two waves per SIMD, whose LDS pass,
// loads, MFMAs and barrier then run one after the other (q/k/v at 64 rows: 1.4 us per 256-deep chunk = the SUM of its parts).
// NORM: A rows are RMS-normalised while they are staged (GemmArgs::ssq, the consumer half of the launch-free residual + RMSNorm).
template <int MT, int NP, int EPI, int KSW = 2, bool NORM = false>
__global__ __launch_bounds__(NP * 64 * (MID_KS / KSW), KSW == 1 ? 1 : (NP == 2 ? 4 : (NP == 4 ? 4 : 2))) void gemm_mid_kernel(GemmArgs g, int k_chunks, int dbg) {
    constexpr int MID_KSW = KSW;
    constexpr int KW = MID_KS / KSW;               // k-step waves per n-tile pair
    constexpr int NW = NP * KW;                    // waves per workgroup
    constexpr int WN = NP * 2;                     // n-tiles per workgroup
    constexpr int UNITS = MT * 2 * MID_KS / 2;     // staging units (8 rows x 128 B = two k-steps) per chunk
    constexpr int AU = (UNITS + NW - 1) / NW;      // ... per wave
    constexpr int BUF = MID_KS * MT * 1024;        // bytes of one staged chunk
    constexpr int LDS_MAIN = (2 * BUF > KW * MT * WN * 1024) ? 2 * BUF : KW * MT * WN * 1024;  // NORM: the norm weight follows
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x [MID_KS k-steps][MT][64 lanes][16 B]; later the K-reduction buffer [4][MT*WN][256] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the buffer descriptors below are built from it
    const int np = wave % NP, wk = wave / NP;
    const int KT = g.K >> 5, NTILES = g.N >> 4;
    const int nt = blockIdx.x * WN + np * 2;       // first n-tile of the pair
    const int chunk0 = blockIdx.y * k_chunks;
    const int m0 = blockIdx.z * (MT * 16);

    f32x4_t acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[mt][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // weight fragments of (chunk c, k-steps wk*KSW + j) for the two n-tiles: two chunks in flight, slot c % 2 reloaded right
    // after use.  Loads go through buffer descriptors that cover exactly this slice of each n-tile: prefetches past the end
    // (and tiles past N) return zeros without memory traffic and without a branch (a conditional load makes hipcc drain
    // vmcnt(0) in the loop).
    __amdgpu_buffer_rsrc_t wrsrc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const bool wvalid = nt + nb < NTILES && !(dbg & 2);
        wrsrc[nb] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.Wp) + ((long)(wvalid ? nt + nb : 0) * KT + (long)chunk0 * MID_KS) * 512, 0,
                                                      wvalid ? k_chunks * (MID_KS * 1024) : 0, 0x00020000);
    }
    const int woff = (wk * MID_KSW * 64 + lane) * 16;
    auto load_b = [&](int c, int j, int nb) -> u32x4_t {
        return __builtin_amdgcn_raw_buffer_load_b128(wrsrc[nb], woff + c * (MID_KS * 1024) + j * 1024, 0, 2 /* nt: read once */);
    };
    u32x4_t wf[2][MID_KSW][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < MID_KSW; ++j)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wf[u][j][nb] = load_b(u, j, nb);

    __shared__ float rsS[NORM ? 64 : 1];
    // ---- A staging: unit = 8 rows x 128 B (two k-steps); lane -> (row rlo = lane & 7, 16-byte piece p = lane >> 3) ----
    // One register set: chunk c+1 is written to LDS at the START of iteration c (its loads were issued a whole iteration
    // earlier) and the same registers are re-armed with chunk c+2 right away.
    const int rlo = lane & 7, p = lane >> 3;
    // rows >= M re-read row M-1 (their products land in output rows >= M, which are never stored); chunks past the slice
    // read on inside the row or, past the last row, hit the descriptor's bound
    const __amdgpu_buffer_rsrc_t arsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.A), 0, (dbg & 1) ? 0 : (int)((((long)g.M - 1) * g.lda + g.K) * 2), 0x00020000);
    int aoff[AU];     // byte offset of this lane's piece of chunk 0
    int adst[AU];     // byte offset inside one LDS buffer, -1: this wave has no such unit
#pragma unroll
    for (int a = 0; a < AU; ++a) {
        const int unit = wave + a * NW;
        const int kp = unit % (MID_KS / 2), rg = unit / (MID_KS / 2);  // k-step pair inside the chunk, 8-row group
        const int row = rg * 8 + rlo;  // 0 .. MT*16-1
        const int grow = min(m0 + row, g.M - 1);
        aoff[a] = (int)(((long)grow * g.lda + (long)chunk0 * MID_CK + kp * 64 + p * 8) * 2);
        const int ks = kp * 2 + (p >> 2), q = p & 3;
        adst[a] = (((ks * MT + (row >> 4)) * 64) + q * 16 + (row & 15)) * 16;
        if (unit >= UNITS) adst[a] = -1;
    }
    auto load_a = [&](int c, int a) -> u32x4_t { return __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[a] + c * (MID_CK * 2), 0, 0); };
    // NORM: every unit of a wave sits at the same k-step pair (NW is a multiple of MID_KS / 2), so ONE 16-byte piece of the norm weight per
    // chunk serves all of the lane's units; it travels with the A registers (same distance ahead); [3P] HF LlamaRMSNorm rounding points
    static_assert(NW % (MID_KS / 2) == 0, "units of a wave share their k-step pair");
    const int nwoff = LDS_MAIN + (chunk0 * MID_CK + (wave % (MID_KS / 2)) * 64 + p * 8) * 2;
    // (1/rms is re-read from LDS per unit and the eight values go through one dword at a time: the 16-wave forms sit at the 128-VGPR cap)
    auto staged = [&](const u32x4_t& xv, int c, int a) -> u32x4_t {
        if constexpr (!NORM) return xv;
        const u32x4_t nwv = *reinterpret_cast<const u32x4_t*>(smem + nwoff + c * (MID_CK * 2));
        const float rs = rsS[min(m0 + ((wave + a * NW) / (MID_KS / 2)) * 8 + rlo, g.M - 1)];
        u32x4_t o;
        o.x = pack_bf(lo_bf(nwv.x) * bfr(lo_bf(xv.x) * rs), hi_bf(nwv.x) * bfr(hi_bf(xv.x) * rs));
        o.y = pack_bf(lo_bf(nwv.y) * bfr(lo_bf(xv.y) * rs), hi_bf(nwv.y) * bfr(hi_bf(xv.y) * rs));
        o.z = pack_bf(lo_bf(nwv.z) * bfr(lo_bf(xv.z) * rs), hi_bf(nwv.z) * bfr(hi_bf(xv.z) * rs));
        o.w = pack_bf(lo_bf(nwv.w) * bfr(lo_bf(xv.w) * rs), hi_bf(nwv.w) * bfr(hi_bf(xv.w) * rs));
        return o;
    };
    // ---- NORM: 1/rms of every row from the producer's per-pair sums of squares, 4 lanes per row, fixed summation order (behind the first A loads) ----
    auto norm_prologue = [&]() {
        static_assert(MT <= 4, "one 64-row block");
        const int row = tid >> 2, q = tid & 3;
        float pq = 0.f;
        if (row < g.M) {
            const int per = g.ssq_n >> 2;
            const float* sp = g.ssq + (long)row * g.ssq_n + q * per;
            if ((per & 15) == 0) {  // all loads of a 16-float group in flight before the first add (a plain loop waits for every load in turn;
                                    // 32 at a time spills in the 16-wave forms, which sit at the 128-VGPR cap)
                for (int i0 = 0; i0 < per; i0 += 16) {
                    f32x4_t t4[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) t4[u] = *reinterpret_cast<const f32x4_t*>(sp + i0 + 4 * u);
#pragma unroll
                    for (int u = 0; u < 4; ++u) { pq += t4[u].x; pq += t4[u].y; pq += t4[u].z; pq += t4[u].w; }
                }
            } else if ((per & 3) == 0) {
                for (int i0 = 0; i0 < per; i0 += 4) { const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(sp + i0); pq += t4.x; pq += t4.y; pq += t4.z; pq += t4.w; }
            } else {
                for (int i = 0; i < per; ++i) pq += sp[i];
            }
        }
        // the norm weight (K bf16) sits in LDS behind the staging / reduction buffers: a 16-byte ds_read at the point of use instead of a
        // register set travelling with the A registers (the 16-wave forms are at the 128-VGPR cap)
        for (int i = tid; i < (g.K >> 3); i += NW * 64)
            *reinterpret_cast<u32x4_t*>(smem + LDS_MAIN + i * 16) = *reinterpret_cast<const u32x4_t*>(g.norm_w + i * 8);
        const float o1 = __shfl_xor(pq, 1, WAVE);
        const float s2 = (q & 1) ? o1 + pq : pq + o1;   // (even + odd), the same operand order in both lanes
        const float o2 = __shfl_xor(s2, 2, WAVE);
        const float tot = (q & 2) ? o2 + s2 : s2 + o2;
        if (row < g.M && q == 0) rsS[row] = rsqrtf(tot / g.K + g.norm_eps);
        __syncthreads();
    };
    // (the prologue runs behind the first A loads, except in the 16-wave forms: there that order costs two spilled registers at the 128-VGPR cap)
    u32x4_t areg[AU];
    if constexpr (NORM && NP == 4) norm_prologue();
#pragma unroll
    for (int a = 0; a < AU; ++a) areg[a] = load_a(0, a);
    if constexpr (NORM && NP != 4) norm_prologue();
#pragma unroll
    for (int a = 0; a < AU; ++a)
        if (adst[a] >= 0) *reinterpret_cast<u32x4_t*>(smem + adst[a]) = staged(areg[a], 0, a);
#pragma unroll
    for (int a = 0; a < AU; ++a) areg[a] = load_a(1, a);
    __syncthreads();

    for (int c0 = 0; c0 < k_chunks; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = c0 + u;
            if (c < k_chunks) {  // uniform over the workgroup
                // order inside the wave's (in-order) LDS queue: this chunk's fragment reads FIRST, then the writes of the next chunk --
                // the MFMAs below wait for the reads only and the ~400-cycle write pass drains underneath them
                const unsigned char* buf = smem + (c & 1) * BUF + ((wk * MID_KSW * MT) * 64 + lane) * 16;
                u32x4_t af[MID_KSW][MT];
#pragma unroll
                for (int j = 0; j < MID_KSW; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) af[j][mt] = *reinterpret_cast<const u32x4_t*>(buf + (j * MT + mt) * 1024);
                if (c + 1 < k_chunks) {
                    unsigned char* nbuf = smem + ((c + 1) & 1) * BUF;
#pragma unroll
                    for (int a = 0; a < AU; ++a)
                        if (adst[a] >= 0) *reinterpret_cast<u32x4_t*>(nbuf + adst[a]) = staged(areg[a], c + 1, a);
                }
#pragma unroll
                for (int a = 0; a < AU; ++a) areg[a] = load_a(c + 2, a);
#pragma unroll
                for (int j = 0; j < MID_KSW; ++j)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[j][mt]), __builtin_bit_cast(bf16x8_t, wf[u][j][nb]), acc[mt][nb], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < MID_KSW; ++j)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) wf[u][j][nb] = load_b(c + 2, j, nb);
                __syncthreads();
            }
        }
    }

    // ---- K reduction across the KW k-step waves through LDS: red[wk][(mt*WN + n)*4 + r][lane], n = 2 np + nb ----
    float* red = reinterpret_cast<float*>(smem);
    constexpr int TILES = MT * WN;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(long)wk * (TILES * 256) + ((mt * WN + np * 2 + nb) * 4 + r) * 64 + lane] = acc[mt][nb][r];
    __syncthreads();

    const int nt0 = blockIdx.x * WN;
    if constexpr (EPI == EPI_PARTIAL) {
        if (g.tickets) {
            // ---- launch-free residual + RMSNorm, producer half (MI355X_MICROARCH.md "Hand-offs measured with sc1 loads", row 1) ----
            // 1. this slice's slab goes out write-through (sc1), every storing wave drains its stores
            float* slab = reinterpret_cast<float*>(g.out);
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)((long)gridDim.y * g.out_batch * 4), 0x00020000);
            for (int e = tid; e < TILES * 256; e += NW * 64) {
                const int ot = e >> 8, rl = e & 255, r = rl >> 6, l = rl & 63;
                const int mt = ot / WN, nb = ot % WN;
                float sacc = 0.f;
#pragma unroll
                for (int w = 0; w < KW; ++w) sacc += red[(long)w * (TILES * 256) + ((mt * WN + nb) * 4 + r) * 64 + l];
                const int row = m0 + mt * 16 + (l >> 4) * 4 + r, col = (nt0 + nb) * 16 + (l & 15);
                if (row < g.M && col < g.n_valid)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sacc), srs, (unsigned)(((long)blockIdx.y * g.out_batch + (long)row * g.ldo + col) * 4), 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // 2. one agent-scope add per workgroup on the column block's counter; the workgroup that draws the last ticket reduces
            __shared__ int s_tk;
            if (tid == 0) s_tk = __hip_atomic_fetch_add(g.tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (s_tk != (int)gridDim.y - 1) return;
            // 3. x = bf16(x + bf16(sum of the slabs, slice order)) for this block's columns, all rows (rowops.hip rmsnorm_reduce_kernel's arithmetic:
            //    the same bits); sums of squares per (row, 32-column pair): 8 lanes x 4 columns, fixed tree
            constexpr int C4 = WN * 4;  // float4 tasks per row
            bf16_t* x = const_cast<bf16_t*>(g.res);
            const int tasks = g.M * C4;
            for (int t0 = 0; t0 < tasks; t0 += NW * 64) {
                const int t = t0 + tid;
                const bool tv = t < tasks;
                const int row = tv ? t / C4 : 0, col = nt0 * 16 + (tv ? t % C4 : 0) * 4;
                const bool valid = tv && col < g.n_valid;
                f32x4_t acc4 = {0.f, 0.f, 0.f, 0.f};
                // (all slices' loads in flight before the first add: they come from memory, ~1.5 us each if taken one by one; slices past gridDim.y
                //  fall outside the descriptor and read zeros, and adding +0 changes no bit of a partial sum)
                for (unsigned s0 = 0; s0 < gridDim.y; s0 += 8) {
                    u32x4_t raw[8];
#pragma unroll
                    for (unsigned u = 0; u < 8; ++u)
                        raw[u] = __builtin_amdgcn_raw_buffer_load_b128(srs, (valid && s0 + u < gridDim.y) ? (unsigned)(((long)(s0 + u) * g.out_batch + (long)row * g.ldo + col) * 4) : 0xffffffffu, 0, 16);
#pragma unroll
                    for (unsigned u = 0; u < 8; ++u) {
                        acc4.x += __uint_as_float(raw[u].x); acc4.y += __uint_as_float(raw[u].y); acc4.z += __uint_as_float(raw[u].z); acc4.w += __uint_as_float(raw[u].w);
                    }
                }
                float sq = 0.f;
                if (valid && g.reduce_plain) {
                    u32x2_t xo;
                    xo.x = pack_bf(acc4.x, acc4.y); xo.y = pack_bf(acc4.z, acc4.w);
                    *reinterpret_cast<u32x2_t*>(x + (long)row * g.ldres + col) = xo;
                } else if (valid) {
                    u32x2_t* xp = reinterpret_cast<u32x2_t*>(x + (long)row * g.ldres + col);
                    const u32x2_t xin = *xp;
                    const float v0 = bfr(lo_bf(xin.x) + bfr(acc4.x)), v1 = bfr(hi_bf(xin.x) + bfr(acc4.y));
                    const float v2 = bfr(lo_bf(xin.y) + bfr(acc4.z)), v3 = bfr(hi_bf(xin.y) + bfr(acc4.w));
                    u32x2_t xo;
                    xo.x = pack_bf(v0, v1); xo.y = pack_bf(v2, v3);
                    *xp = xo;
                    sq = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
                }
                sq += __shfl_xor(sq, 1, WAVE);
                sq += __shfl_xor(sq, 2, WAVE);
                sq += __shfl_xor(sq, 4, WAVE);
                if (valid && (t & 7) == 0 && g.ssq && !g.reduce_plain) g.ssq[(long)row * g.ssq_n + (col >> 5)] = sq;  // (plain: ssq, if any, is this launch's INPUT)
            }
            if (tid == 0) g.tickets[blockIdx.x] = 0;  // re-armed for the next launch (launch boundary = visibility)
            return;
        }
    }
    constexpr int OUT_TILES = (EPI == EPI_SWIGLU) ? TILES / 2 : TILES;
    for (int e = tid; e < OUT_TILES * 256; e += NW * 64) {
        const int ot = e >> 8, rl = e & 255;
        const int r = rl >> 6, l = rl & 63;
        int mt, nb;
        if constexpr (EPI == EPI_SWIGLU) { mt = ot / (WN / 2); nb = (ot % (WN / 2)) * 2; }
        else { mt = ot / WN; nb = ot % WN; }
        float s = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < KW; ++w) {
            const float* pr = red + (long)w * (TILES * 256);
            s += pr[((mt * WN + nb) * 4 + r) * 64 + l];
            if constexpr (EPI == EPI_SWIGLU) s2 += pr[((mt * WN + nb + 1) * 4 + r) * 64 + l];
        }
        const int row = m0 + mt * 16 + (l >> 4) * 4 + r;
        int col;
        if constexpr (EPI == EPI_SWIGLU) col = ((nt0 + nb) >> 1) * 16 + (l & 15);
        else col = (nt0 + nb) * 16 + (l & 15);
        if (row >= g.M || col >= g.n_valid) continue;
        if constexpr (EPI == EPI_PARTIAL) {
            reinterpret_cast<float*>(g.out)[(long)blockIdx.y * g.out_batch + (long)row * g.ldo + col] = s;
        } else if constexpr (EPI == EPI_F32) {
            reinterpret_cast<float*>(g.out)[(long)row * g.ldo + col] = bfr(s);
        } else {
            float v;
            if constexpr (EPI == EPI_NONE) v = s;
            else if constexpr (EPI == EPI_BIAS) v = s + bf2f(g.bias[col]);
            else if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(bfr(s + bf2f(g.bias[col])));
            else if constexpr (EPI == EPI_RES) v = bf2f(g.res[(long)row * g.ldres + col]) + bfr(s);
            else if constexpr (EPI == EPI_BIAS_RES) v = bf2f(g.res[(long)row * g.ldres + col]) + bfr(s + bf2f(g.bias[col]));
            else /* EPI_SWIGLU */ v = bfr(silu(bfr(s))) * bfr(s2);
            reinterpret_cast<bf16_t*>(g.out)[(long)row * g.ldo + col] = f2bf(v);
        }
    }
}

static int g_mid_min_rows = ISST_MID_MIN_ROWS;  // rows from which (exclusive) the kernel takes over from the skinny one (gemm_mid_set_min_rows)
void gemm_mid_set_min_rows(int rows) { g_mid_min_rows = rows; }
static int g_mid_max_rows = 64;  // above 64 rows the kernel runs 64-row m-blocks (grid.z) that re-read the weights through L2
static int g_mid_wn = 0, g_mid_dbg = 0, g_mid_kw4 = 0;  // tuning override (profiles/mid_probe.py): 0 = heuristic; bits 4-5: timing-only builds (A / W descriptor emptied)
void gemm_mid_set_tuning(int wn) { g_mid_wn = wn & 15; g_mid_dbg = (wn >> 4) & 3; g_mid_kw4 = (wn >> 6) & 1; if (wn >> 8) g_mid_max_rows = wn >> 8; }

bool gemm_mid_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    // up to 64 rows always; 65..128 rows as two 64-row blocks for the short weight streams only (q/k/v at 88 rows: 35 us against 56 us
    // on the 128x128 dense kernel, whose 48 column blocks leave most CUs idle; gate/up is faster there: profiles/rows_probe.py)
    const int max_rows = (g.ksplit <= 1 && (long)g.N * g.K <= (32L << 20) && g_mid_max_rows < 128) ? 128 : g_mid_max_rows;
    return g.batch == 1 && g.M > g_mid_min_rows && g.M <= max_rows && g.K % (MID_CK * ks) == 0 && g.N % 16 == 0 && g.lda % 8 == 0 && (!g.norm_w || (g.ssq && g.M <= 64 && g.ssq_n % 4 == 0 && g.ssq_n * 32 == g.K)) &&
           (!g.tickets || (g.epi == EPI_PARTIAL && g.M <= 64 && g.res && g.N % 32 == 0 && (g.reduce_plain || !g.norm_w)));
}
// worth it only when the weight stream is long: the encoder's 2-8 MB projections at 48 rows are latency-bound and run
// faster on the skinny kernel's many 16-column workgroups (11.2 vs 14.9 us for fc2)
bool gemm_mid_preferred(const GemmArgs& g) { return g.ssq || g.tickets || g.ksplit > 1 || g.epi == EPI_PARTIAL || (long)g.N * g.K >= (8L << 20) || g_mid_wn != 0; }

template <int MT, int NP, int EPI, int KSW, bool NORM>
static int launch_mid_cfg2(const GemmArgs& g, hipStream_t stream);
template <int MT, int NP, int EPI, int KSW = 2>
static int launch_mid_cfg(const GemmArgs& g, hipStream_t stream) {
    if constexpr (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_F32 || EPI == EPI_PARTIAL) {
        if (g.norm_w) return launch_mid_cfg2<MT, NP, EPI, KSW, true>(g, stream);
    }
    return launch_mid_cfg2<MT, NP, EPI, KSW, false>(g, stream);
}
template <int MT, int NP, int EPI, int KSW, bool NORM>
static int launch_mid_cfg2(const GemmArgs& g, hipStream_t stream) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const int NTILES = g.N / 16, WN = NP * 2;
    constexpr int KW = MID_KS / KSW;
    dim3 grid((NTILES + WN - 1) / WN, ks, (g.M + MT * 16 - 1) / (MT * 16)), block(NP * 64 * KW);
    size_t lds = (size_t)2 * MID_KS * MT * 1024;  // A double buffer 16 MT KiB >= K-reduction buffer KW * MT * WN KiB (NP <= 2 at KW = 4, NP = 1 at KW = 8)
    if ((size_t)KW * MT * WN * 1024 > lds) lds = (size_t)KW * MT * WN * 1024;
    if (NORM) lds += (size_t)g.K * 2;  // the norm weight (gemm_mid_kernel LDS_MAIN)
    if (lds > 64 * 1024) {
        static size_t attr_lds = 0;  // (NORM: the size depends on K)
        if (lds > attr_lds && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_mid_kernel<MT, NP, EPI, KSW, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        if (lds > attr_lds) attr_lds = lds;
    }
    hipLaunchKernelGGL((gemm_mid_kernel<MT, NP, EPI, KSW, NORM>), grid, block, lds, stream, g, g.K / MID_CK / ks, g_mid_dbg);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

template <int MT, int EPI>
static int launch_mid_wn(const GemmArgs& g, hipStream_t stream) {
    // columns per workgroup (profiles/mid_probe.py, r01/mid_probe.txt): 64 when that still gives most CUs a workgroup, else 32
    // (A is then staged twice as often); up to 32 rows the 4-wave / 32-column form wins except for the widest projection
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    int wn;
    if (g.M <= 32) wn = (g.N >= 16384 || (long)(g.N / 64) * ks >= 256) ? 4 : 2;  // (K slices with the in-launch reduction, 22 rows: down_proj 4 x 64 columns 23.1 us, 4 x 32: 24.9)
    else wn = ((long)(g.N / 64) * ks >= 192) ? 4 : 2;
    // the widest projections (gate/up, lm_head: >= 192 workgroups even at 128 columns) take 128 columns and 16 waves: A is staged half as
    // often -- at 64 rows its L2->LDS traffic equals the weight bytes otherwise (gate/up 53.8 -> 44.3 us, the weight-only time; 22 rows
    // 42.9 -> 41.7; q/k/v, o_proj, down lose: too few workgroups)
    if (ks == 1 && g.N / 128 >= 192) wn = 8;
    if (g_mid_wn == 2 || g_mid_wn == 4 || g_mid_wn == 8) wn = g_mid_wn;
    if (wn == 8) return launch_mid_cfg<MT, 4, EPI>(g, stream);
    // 32- and 64-column workgroups take the 8-k-step-wave form (profiles/mid_kw8_probe.py, r02/mid_kw8_probe.txt: q/k/v 23.1 -> 21.6 us at 64 rows,
    // 15.0 -> 13.9 at 22; o_proj + norm 19.6 -> 18.4 / 14.3 -> 13.8; down + norm 33.0 -> 32.6 / 29.0 -> 27.8); gemm_mid_set_tuning bit 6 forces
    // the 4-wave form back for A/B runs
    if (!g_mid_kw4) {
        if (wn >= 4) return launch_mid_cfg<MT, 2, EPI, 1>(g, stream);
        return launch_mid_cfg<MT, 1, EPI, 1>(g, stream);
    }
    if (wn >= 4) return launch_mid_cfg<MT, 2, EPI>(g, stream);
    return launch_mid_cfg<MT, 1, EPI>(g, stream);
}

template <int EPI>
static int launch_mid_mt(const GemmArgs& g, hipStream_t stream) {
    if (g.M <= 16) return launch_mid_wn<1, EPI>(g, stream);
    if (g.M <= 32) return launch_mid_wn<2, EPI>(g, stream);
    if (g.M <= 48) return launch_mid_wn<3, EPI>(g, stream);
    return launch_mid_wn<4, EPI>(g, stream);
}

int launch_gemm_mid(const GemmArgs& g, hipStream_t stream) {
    if (!gemm_mid_supported(g)) return ISST_ERR_ARG;
    if (g.ksplit > 1 && g.epi != EPI_PARTIAL) return ISST_ERR_ARG;
    switch (g.epi) {
        case EPI_NONE: return launch_mid_mt<EPI_NONE>(g, stream);
        case EPI_BIAS: return g.bias ? launch_mid_mt<EPI_BIAS>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_GELU: return g.bias ? launch_mid_mt<EPI_BIAS_GELU>(g, stream) : ISST_ERR_ARG;
        case EPI_RES: return g.res ? launch_mid_mt<EPI_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_RES: return (g.res && g.bias) ? launch_mid_mt<EPI_BIAS_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_SWIGLU: return (g.N % 32 == 0) ? launch_mid_mt<EPI_SWIGLU>(g, stream) : ISST_ERR_ARG;
        case EPI_F32: return launch_mid_mt<EPI_F32>(g, stream);
        case EPI_PARTIAL: return launch_mid_mt<EPI_PARTIAL>(g, stream);
    }
    return ISST_ERR_ARG;
}
