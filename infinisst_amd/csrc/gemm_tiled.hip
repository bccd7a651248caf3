// LDS-tiled MFMA GEMM for the dense (large-M) shapes of the InfiniSST path on gfx950:
//   out[M,N] = epi(A[M,K] @ W[N,K]^T),  M > 64 rows or batched (conv stack as implicit GEMM, multi-stream prefill,
//   encoder at many streams).  The skinny kernel (gemm.hip) covers the weight-streaming M <= 64 shapes.
//
// Workgroup = 4 waves, output tile 128 x 128, K step 64:
//   * A (activations, row stride lda, rows may overlap for the implicit-GEMM convs) is staged through LDS in full
//     128-byte lines, double-buffered: the global loads of step t+1 are issued before the MFMAs of step t and written
//     to the other buffer afterwards (async-STAGE split), one barrier per K step;
//   * B comes from the fragment-major packed weights (one contiguous 1 KiB per (n-tile, k-tile) and wave): straight to
//     registers, prefetched one K step ahead -- no LDS image, no transposition;
//   * wave w owns n-tiles 2w, 2w+1 of the block's 8 (32 columns) x all 8 m-tiles: 16 accumulators, 32 MFMAs
//     (v_mfma_f32_16x16x32_bf16) per K step; the (gate, up) pair of the SwiGLU epilogue sits in one wave.
// LDS image of one A step: [128 rows][64 k] bf16 with the 16-byte chunk index XOR-swizzled by (row & 7): a wave's
// ds_read_b128 (16 rows x 4 chunks) then touches 16 distinct 16-byte slots per 256-byte bank row.
#include "common.h"

#define TM 128
#define TN_TILES 8
#define TK 64

__device__ __forceinline__ int a_off(int row, int chunk) { return row * TK + ((chunk ^ (row & 7)) << 3); }  // bf16 elements

template <int EPI>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(GemmArgs g, int raster) {
    __shared__ __attribute__((aligned(16))) bf16_t As[2][TM * TK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int KT = g.K >> 5, NTILES = g.N >> 4;
    // XCD-aware rasterisation (raster != 0: 1-D launch of 8 * n_per workgroups).  Workgroups go to the 8 XCDs round-robin, and with the
    // plain (column block, row block) grid the ~64 workgroups an XCD runs at a time are 64 column blocks of ONE row block: its L2 shares the
    // A rows 64 ways and the weights not at all, so every row block re-reads all of W through the fabric -- at 1408 rows 11 x 235 MB for
    // gate/up in 405 us = 6.4 TB/s: the kernel was bound by memory-side bandwidth, not by MFMA or LDS (the same 6.0 / 5.3 / 7.6 TB/s fall
    // out of down_proj, q/k/v and a 4096 x 8192 x 8192 square).  Here XCD c takes a contiguous run of the tile sequence, and the sequence
    // walks patches of 8 column blocks x all row blocks, column fastest: an XCD's 64 concurrent workgroups are 8 x 8 tiles, A and W
    // both shared 8 ways in its L2.
    int bx = blockIdx.x, by = blockIdx.y;
    if (raster) {
        const int X = (NTILES + TN_TILES - 1) / TN_TILES, Y = (g.M + TM - 1) / TM;
        const int n_per = gridDim.x >> 3;
        const int S = (blockIdx.x & 7) * n_per + (blockIdx.x >> 3);
        if (S >= X * Y) return;  // padding of the 1-D grid (uniform over the workgroup)
        const int patch = S / (8 * Y), r = S - patch * (8 * Y);
        const int pw = min(8, X - patch * 8);
        by = r / pw;
        bx = patch * 8 + r % pw;
    }
    const int m0 = by * TM;
    const int nt0 = bx * TN_TILES + wave * 2;
    // blockIdx.z: batch index, or -- EPI_PARTIAL, batch == 1 -- the K slice of a split-K launch (narrow N at 65..512 rows: without
    // it the 32 column blocks of o_proj / down_proj walk all of K alone, 217 us for down_proj whatever the row count)
    const int ks = (EPI == EPI_PARTIAL && g.ksplit > 1) ? g.ksplit : 1;
    const long b = (EPI == EPI_PARTIAL) ? 0 : blockIdx.z;
    const int slice = (EPI == EPI_PARTIAL) ? blockIdx.z : 0;
    // K steps of slice s: (K / TK) / ks, the first (K / TK) % ks slices one more (the same partition as gemm_dense.hip's folded grid: slab s is the partial sum over
    // exactly these steps in both kernels)
    const int ksteps = g.K / TK, sq = ksteps / ks, srem = ksteps - sq * ks;
    const int steps = sq + (slice < srem ? 1 : 0);
    const long kstart = (long)slice * sq + min(slice, srem);  // first K step of the slice
    const bf16_t* A = g.A + b * g.a_batch + kstart * TK;

    // staging role: thread t moves 4 x 16 bytes of row (t >> 1): chunks 4*(t & 1) .. +3
    const int srow = tid >> 1, sc0 = (tid & 1) * 4;
    const int grow = min(m0 + srow, g.M - 1);
    const bf16_t* aptr = A + (long)grow * g.lda + sc0 * 8;
    const bool nv0 = nt0 < NTILES, nv1 = nt0 + 1 < NTILES;
    const u32x4_t* w0 = reinterpret_cast<const u32x4_t*>(g.Wp) + ((long)(nv0 ? nt0 : 0) * KT + kstart * 2) * 64 + lane;
    const u32x4_t* w1 = reinterpret_cast<const u32x4_t*>(g.Wp) + ((long)(nv1 ? nt0 + 1 : 0) * KT + kstart * 2) * 64 + lane;

    f32x4_t acc[8][2];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) { acc[mt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    u32x4_t areg[4], breg[2][2];  // staged A chunks of the next step; B fragments [k-tile of the step][n-tile]
#pragma unroll
    for (int c = 0; c < 4; ++c) areg[c] = *reinterpret_cast<const u32x4_t*>(aptr + c * 8);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) { breg[kk][0] = w0[(long)kk * 64]; breg[kk][1] = w1[(long)kk * 64]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4_t*>(&As[0][a_off(srow, sc0 + c)]) = areg[c];
    __syncthreads();

    for (int t = 0; t < steps; ++t) {
        const int cur = t & 1;
        u32x4_t bcur[2][2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { bcur[kk][0] = breg[kk][0]; bcur[kk][1] = breg[kk][1]; }
        if (t + 1 < steps) {  // issue the next step's loads before this step's MFMAs
            const long ko = (long)(t + 1) * TK;
#pragma unroll
            for (int c = 0; c < 4; ++c) areg[c] = *reinterpret_cast<const u32x4_t*>(aptr + ko + c * 8);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) { breg[kk][0] = w0[((long)(t + 1) * 2 + kk) * 64]; breg[kk][1] = w1[((long)(t + 1) * 2 + kk) * 64]; }
        }
        // all 8 A fragments of a k-tile are requested before its 16 MFMAs (left to itself hipcc emits 2 reads -> wait -> 4 MFMAs,
        // eight times per k-tile: every group exposes the LDS latency and the matrix pipe idles 70 % of the time)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            u32x4_t af[8];
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) af[mt] = *reinterpret_cast<const u32x4_t*>(&As[cur][a_off(mt * 16 + fr, kk * 4 + fq)]);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[mt]), __builtin_bit_cast(bf16x8_t, bcur[kk][0]), acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[mt]), __builtin_bit_cast(bf16x8_t, bcur[kk][1]), acc[mt][1], 0, 0, 0);
            }
        }
        if (t + 1 < steps) {
#pragma unroll
            for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4_t*>(&As[cur ^ 1][a_off(srow, sc0 + c)]) = areg[c];
        }
        __syncthreads();
    }

    // ---- epilogue straight from the accumulators: acc[mt][nb][r] = C[m0 + mt*16 + 4fq + r][(nt0 + nb)*16 + fr] ----
    const bf16_t* res = g.res ? g.res + b * g.res_batch : nullptr;
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mt * 16 + fq * 4 + r;
            if (row >= g.M) continue;
            if constexpr (EPI == EPI_SWIGLU) {
                const int col = (nt0 >> 1) * 16 + fr;
                if (nv0 && col < g.n_valid)
                    reinterpret_cast<bf16_t*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = f2bf(bfr(silu(bfr(acc[mt][0][r]))) * bfr(acc[mt][1][r]));
            } else {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = (nt0 + nb) * 16 + fr;
                    if (!(nb ? nv1 : nv0) || col >= g.n_valid) continue;
                    const float s = acc[mt][nb][r];
                    if constexpr (EPI == EPI_PARTIAL) {
                        reinterpret_cast<float*>(g.out)[(long)slice * g.out_batch + (long)row * g.ldo + col] = s;
                    } else if constexpr (EPI == EPI_F32) {
                        reinterpret_cast<float*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = bfr(s);
                    } else {
                        float v;
                        if constexpr (EPI == EPI_NONE) v = s;
                        else if constexpr (EPI == EPI_BIAS) v = s + bf2f(g.bias[col]);
                        else if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(bfr(s + bf2f(g.bias[col])));
                        else if constexpr (EPI == EPI_RES) v = bf2f(res[(long)row * g.ldres + col]) + bfr(s);
                        else v = bf2f(res[(long)row * g.ldres + col]) + bfr(s + bf2f(g.bias[col]));
                        reinterpret_cast<bf16_t*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = f2bf(v);
                    }
                }
            }
        }
    }
}

bool gemm_tiled_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    if (g.epi == EPI_PARTIAL && g.batch != 1) return false;
    return g.K % TK == 0 && (g.K / TK) / ks >= 1 && g.lda % 8 == 0 && !g.norm_w && (g.M > 64 || g.batch > 1);
}

static int g_tiled_raster = 1;  // tuning hook (gemm_tiled_set_raster): 0 = plain 2-D grid always, 1 = heuristic, 2 = rasterised always
void gemm_tiled_set_raster(int on) { g_tiled_raster = on; }

int launch_gemm_tiled(const GemmArgs& g, hipStream_t stream) {
    const int NTILES = g.N / 16;
    dim3 grid((NTILES + TN_TILES - 1) / TN_TILES, (g.M + TM - 1) / TM, g.epi == EPI_PARTIAL ? (g.ksplit > 1 ? g.ksplit : 1) : g.batch), block(256);
    // (worth it from ~1.5 rounds of workgroups on: gate/up at 1408 rows, 2464 tiles, 403 -> 356 us = 821 -> 928 TFLOP/s, a 4096 x 8192 x 8192
    //  square 988 -> 1025; grids that are resident all at once already put (few column blocks) x (all row blocks) on an XCD and gain nothing:
    //  q/k/v at 1408 rows, 528 tiles, 103 -> 110 us -- profiles/raster_probe.py)
    const int raster = (g_tiled_raster == 1 ? grid.x * grid.y >= 768 : g_tiled_raster == 2) && grid.y > 1 ? 1 : 0;
    if (raster) {
        const unsigned tiles = grid.x * grid.y;
        grid.x = ((tiles + 7) / 8) * 8;
        grid.y = 1;
    }
#define LAUNCH_T(E) hipLaunchKernelGGL(gemm_tiled_kernel<E>, grid, block, 0, stream, g, raster)
    switch (g.epi) {
        case EPI_NONE: LAUNCH_T(EPI_NONE); break;
        case EPI_BIAS: LAUNCH_T(EPI_BIAS); break;
        case EPI_BIAS_GELU: LAUNCH_T(EPI_BIAS_GELU); break;
        case EPI_RES: LAUNCH_T(EPI_RES); break;
        case EPI_BIAS_RES: LAUNCH_T(EPI_BIAS_RES); break;
        case EPI_SWIGLU: LAUNCH_T(EPI_SWIGLU); break;
        case EPI_F32: LAUNCH_T(EPI_F32); break;
        case EPI_PARTIAL: LAUNCH_T(EPI_PARTIAL); break;
        default: return ISST_ERR_ARG;
    }
#undef LAUNCH_T
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
