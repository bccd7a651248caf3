// Packed-weight skinny GEMM for gfx950:  out[M,N] = epi(A[M,K] @ W[N,K]^T).
//
// This is the kernel every Linear / Conv1d of the InfiniSST hot path runs through (reference call sites:
// model/patches/patch_speech_encoder.py:741-743,923,586-589 (encoder q/k/v/out/fc1/fc2),
// model/speech_encoder.py:233-234 (shrink convs as implicit GEMM, projector), patch_llm.py:260-262,334 and
// HF LlamaMLP / lm_head (model/llm.py:237)).  With one stream per GPU every one of them is a weight-streaming
// problem (M = 1..64 rows against a weight read exactly once), so the design is HBM-first:
//
//   * weights are re-laid out ONCE at load time into MFMA-fragment-major order
//         Wp[n_tile][k_tile][lane 0..63][8 bf16]   (n_tile = 16 rows of W, k_tile = 32 columns)
//     so that the B operand of v_mfma_f32_16x16x32_bf16 for (n_tile,k_tile) is one fully contiguous 1 KiB
//     global_load_dwordx4 per wave -- no LDS round trip, no 64-byte row fragments;
//   * a workgroup owns NTB n-tiles and all of K; its waves interleave k-tiles (wave w takes kt = w, w+W, ..),
//     so the block streams one contiguous [NTB][K/32] KiB run; partial sums meet in LDS once at the end;
//   * A (activations, L2-resident) goes straight to VGPRs in the A-operand layout; rows >= M are not loaded;
//   * the epilogue applies bias / GELU / residual / SwiGLU with the reference's bf16 rounding points.
//
// Operand maps (cdna_hip_programming.md section 3): A lane l holds A[row l&15][k = 8(l>>4)+j], B lane l holds
// B[k = 8(l>>4)+j][col l&15], C/D lane l reg r is (row 4(l>>4)+r, col l&15).
#include "common.h"

#define GEMM_UNROLL 4
// k-tiles in flight per wave and batch: halved for the widest variants (MT*NTB >= 6) so that nothing spills
constexpr int gemm_unroll(int MT, int NTB) { return (MT * NTB >= 6) ? 2 : GEMM_UNROLL; }

// ------------------------------------------------------------------------------------------------
// weight packer: src row-major [n_rows][K] (or Conv1d [n_rows][Cin][conv_k]) -> fragment-major tiles
// dst tile index of source row-tile rt is  row_offset_tiles + rt * tile_stride + tile_phase.
// ------------------------------------------------------------------------------------------------
__global__ void pack_weight_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n_rows, int K,
                                   int row_offset_tiles, int tile_stride, int tile_phase, int conv_k) {
    const int KT = K >> 5;
    const long tile = blockIdx.x;  // rt * KT + kt
    const int rt = (int)(tile / KT), kt = (int)(tile % KT);
    const int lane = threadIdx.x;
    const int n = rt * 16 + (lane & 15);
    const int k0 = kt * 32 + 8 * (lane >> 4);
    bf16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        bf16_t x = 0;
        if (n < n_rows) {
            if (conv_k > 0) {  // GEMM column k = kk * Cin + ci  <-  Conv1d weight [n][ci][kk]
                const int cin = K / conv_k;
                const int ci = k % cin, kk = k / cin;
                x = src[(long)n * K + (long)ci * conv_k + kk];
            } else {
                x = src[(long)n * K + k];
            }
        }
        v[j] = x;
    }
    const long dt = (long)row_offset_tiles + (long)rt * tile_stride + tile_phase;
    bf16_t* d = dst + (dt * KT + kt) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = v[j];
}

int launch_pack_weight(const bf16_t* src, bf16_t* dst, int n_rows, int K, int row_offset_tiles, int tile_stride,
                       int tile_phase, int conv_k, hipStream_t stream) {
    if (K % 32 != 0 || n_rows <= 0) return ISST_ERR_ARG;
    const int RT = (n_rows + 15) / 16;
    const long tiles = (long)RT * (K / 32);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, src, dst, n_rows, K,
                       row_offset_tiles, tile_stride, tile_phase, conv_k);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ u32x4_t load_w(const u32x4_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    return *p;
}

// AMODE 0: A fragments come straight from global memory (any M).
// AMODE 1: decode shapes (M <= 16 rows, M*K*2 <= 64 KiB): the workgroup stages its A rows ONCE in LDS while the first
//          weight fragments are in flight; the k-loop then reads A with ds_read_b128 -- no global A load and no L2
//          round trip inside the loop.
// AMODE 2: as 1, and the staged rows are RMS-normalised in LDS ([3P] HF LlamaRMSNorm: var = mean(x^2) in fp32;
//          bf16(x * rsqrt(var + eps)); bf16(weight * that)) -- removes the separate norm launch in front of the q/k/v,
//          gate/up and lm_head projections (pure launch latency at M = 1).
template <int MT, int NTB, int EPI, bool NT, int AMODE>
__global__ void gemm_skinny_kernel(GemmArgs g) {
    constexpr int UNR = gemm_unroll(MT, NTB);
    // AMODE 0: the reduction buffer [W][MT*NTB*4][64].  AMODE >= 1: the staged A rows [M][K] + [W] partial sums during the k-loop;
    // the reduction buffer then REUSES the same bytes (one more barrier) -- kept apart, 4 rows needed 41 KB per workgroup: 3 instead
    // of 4 workgroups per CU and a second round of workgroups for the 896-workgroup gate/up launch (+7 us)
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int W = blockDim.x >> 6;
    const int KT = g.K >> 5;
    const int NTILES = g.N >> 4;
    const int nt0 = blockIdx.x * NTB;
    const int m0 = blockIdx.y * (MT * 16);
    const long b = blockIdx.z;
    const bf16_t* A = g.A + b * g.a_batch;

    f32x4_t acc[MT][NTB];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) acc[mt][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int arow = lane & 15;
    const int kq = (lane >> 4) * 8;
    const bf16_t* aptr[MT];
    bool avalid[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int row = m0 + mt * 16 + arow;
        avalid[mt] = row < g.M;
        aptr[mt] = A + (long)(avalid[mt] ? row : 0) * g.lda + kq;
    }
    const u32x4_t* wptr[NTB];
    bool wvalid[NTB];
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) {
        const int nt = nt0 + nb;
        wvalid[nb] = nt < NTILES;
        wptr[nb] = reinterpret_cast<const u32x4_t*>(g.Wp) + ((long)(wvalid[nb] ? nt : 0) * KT) * 64 + lane;
    }
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};

    // the first TWO batches of weight fragments go in flight before anything else (also before the norm prologue)
    u32x4_t wf[UNR][NTB], wn[UNR][NTB];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int kt = wave + u * W;
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) wf[u][nb] = (kt < KT && wvalid[nb]) ? load_w<NT>(wptr[nb] + (long)kt * 64) : zero4;
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int kt = wave + (UNR + u) * W;
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb) wn[u][nb] = (kt < KT && wvalid[nb]) ? load_w<NT>(wptr[nb] + (long)kt * 64) : zero4;
    }

    // (1-2 rows keep the two areas apart: they fit 4 workgroups per CU anyway and skip the extra barrier)
    const bool overlay = g.M > 2;
    bf16_t* xs = reinterpret_cast<bf16_t*>(overlay ? red : red + (long)W * (MT * NTB * 256));  // [M][K] staged rows (AMODE >= 1)
    if constexpr (AMODE >= 1) {
        // rows are staged (and normalised) side by side: each row gets wpr = max(1, W / M) waves, W / wpr rows per round
        // (one stream: 1 row on all waves; beam search: 4 rows, one wave each, one round)
        float* part = reinterpret_cast<float*>(xs + (long)g.M * g.K);  // [W] partial sums of squares
        const int wpr = (W >= g.M) ? W / g.M : 1;   // waves per row
        const int rpr = W / wpr;                     // rows per round
        const int my_slot = wave / wpr, my_sub = wave % wpr;
        for (int r0 = 0; r0 < g.M; r0 += rpr) {
            const int rr = r0 + my_slot;
            const bool active = my_slot < rpr && rr < g.M;
            float sq = 0.f;
            // (loads go out four at a time: a chunk-by-chunk loop pays one L2 round trip per chunk -- 8 in a row at 4+ rows)
            const int cstep = wpr * 512, cfirst = (my_sub * 64 + lane) * 8;
            if (active) {
                const bf16_t* xr = A + (long)(m0 + rr) * g.lda;
                for (int c0 = cfirst; c0 < g.K; c0 += 4 * cstep) {
                    u32x4_t xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u * cstep;
                        xv[u] = *reinterpret_cast<const u32x4_t*>(xr + (c < g.K ? c : cfirst));
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u * cstep;
                        if (c < g.K) {
                            *reinterpret_cast<u32x4_t*>(xs + (long)rr * g.K + c) = xv[u];
                            if constexpr (AMODE == 2) {
                                float f[8];
                                unpack8(xv[u], f);
#pragma unroll
                                for (int q = 0; q < 8; ++q) sq += f[q] * f[q];
                            }
                        }
                    }
                }
            }
            if constexpr (AMODE == 2) {
                sq = wave_sum(sq);
                if (lane == 0) part[wave] = sq;
                __syncthreads();
                if (active) {
                    float t = 0.f;
                    for (int w2 = 0; w2 < wpr; ++w2) t += part[my_slot * wpr + w2];
                    const float rs = rsqrtf(t / g.K + g.norm_eps);
                    for (int c0 = cfirst; c0 < g.K; c0 += 4 * cstep) {  // every lane rewrites the chunks it staged
                        u32x4_t wv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = c0 + u * cstep;
                            wv[u] = *reinterpret_cast<const u32x4_t*>(g.norm_w + (c < g.K ? c : cfirst));
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = c0 + u * cstep;
                            if (c < g.K) {
                                float f[8], nw[8];
                                unpack8(*reinterpret_cast<const u32x4_t*>(xs + (long)rr * g.K + c), f);
                                unpack8(wv[u], nw);
#pragma unroll
                                for (int q = 0; q < 8; ++q) f[q] = nw[q] * bfr(f[q] * rs);
                                *reinterpret_cast<u32x4_t*>(xs + (long)rr * g.K + c) = pack8(f);
                            }
                        }
                    }
                }
                __syncthreads();  // part[] is reused by the next round; the last one publishes the rows
            }
        }
        if constexpr (AMODE == 1) __syncthreads();
    }

    for (int kt0 = wave; kt0 < KT; kt0 += W * UNR) {
        u32x4_t af[UNR][MT];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int kt = kt0 + u * W;
            const bool kv = kt < KT;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (AMODE >= 1)
                    af[u][mt] = (kv && avalid[mt]) ? *reinterpret_cast<const u32x4_t*>(xs + (long)arow * g.K + (long)kt * 32 + kq) : zero4;
                else
                    af[u][mt] = (kv && avalid[mt]) ? *reinterpret_cast<const u32x4_t*>(aptr[mt] + (long)kt * 32) : zero4;
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nb = 0; nb < NTB; ++nb)
                    acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8_t, af[u][mt]), __builtin_bit_cast(bf16x8_t, wf[u][nb]), acc[mt][nb], 0, 0, 0);
        // rotate: the batch requested one iteration ago becomes current, and the batch after it is requested now
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int kt = kt0 + (2 * UNR + u) * W;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) {
                wf[u][nb] = wn[u][nb];
                wn[u][nb] = (kt < KT && wvalid[nb]) ? load_w<NT>(wptr[nb] + (long)kt * 64) : zero4;
            }
        }
    }

    // ---- cross-wave reduction through LDS: red[wave][(mt*NTB+nb)*4 + r][lane] ----
    if constexpr (AMODE >= 1) {
        if (overlay) __syncthreads();  // every wave is done reading the staged rows that `red` overlays
    }
    constexpr int TILES = MT * NTB;
    float* my = red + (long)wave * (TILES * 256);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < NTB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) my[((mt * NTB + nb) * 4 + r) * 64 + lane] = acc[mt][nb][r];
    __syncthreads();

    const bf16_t* res = g.res ? g.res + b * g.res_batch : nullptr;
    constexpr int OUT_TILES = (EPI == EPI_SWIGLU) ? TILES / 2 : TILES;
    for (int e = threadIdx.x; e < OUT_TILES * 256; e += blockDim.x) {
        const int ot = e >> 8;  // output tile index within the block
        const int rl = e & 255;
        const int r = rl >> 6, l = rl & 63;
        int mt, nb;
        if constexpr (EPI == EPI_SWIGLU) { mt = ot / (NTB / 2); nb = (ot % (NTB / 2)) * 2; }
        else { mt = ot / NTB; nb = ot % NTB; }
        float s = 0.f, s2 = 0.f;
        for (int w = 0; w < W; ++w) {
            const float* p = red + (long)w * (TILES * 256);
            s += p[((mt * NTB + nb) * 4 + r) * 64 + l];
            if constexpr (EPI == EPI_SWIGLU) s2 += p[((mt * NTB + nb + 1) * 4 + r) * 64 + l];
        }
        const int row = m0 + mt * 16 + (l >> 4) * 4 + r;
        int col;
        if constexpr (EPI == EPI_SWIGLU) col = ((nt0 + nb) >> 1) * 16 + (l & 15);
        else col = (nt0 + nb) * 16 + (l & 15);
        if (row >= g.M || col >= g.n_valid) continue;
        if constexpr (EPI == EPI_F32) {
            reinterpret_cast<float*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = bfr(s);
        } else {
            float v;
            if constexpr (EPI == EPI_NONE) v = s;
            else if constexpr (EPI == EPI_BIAS) v = s + bf2f(g.bias[col]);
            else if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(bfr(s + bf2f(g.bias[col])));
            else if constexpr (EPI == EPI_RES) v = bf2f(res[(long)row * g.ldres + col]) + bfr(s);
            else if constexpr (EPI == EPI_BIAS_RES) v = bf2f(res[(long)row * g.ldres + col]) + bfr(s + bf2f(g.bias[col]));
            else /* EPI_SWIGLU */ v = bfr(silu(bfr(s))) * bfr(s2);
            reinterpret_cast<bf16_t*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = f2bf(v);
        }
    }
}

// tuning overrides for profiles/gemv_sweep.py (0 = heuristic)
static int g_tune_w = 0, g_tune_ntb = 0, g_force_skinny = 0;
void gemm_set_tuning(int w, int ntb) { g_tune_w = w < 0 ? 0 : w; g_tune_ntb = ntb; g_force_skinny = w < 0; gemm_mid_set_tuning(ntb); }  // w < 0: never use the tiled kernel

static inline bool gemm_can_stage(const GemmArgs& g) {
    return g.batch == 1 && g.M <= 16 && (size_t)g.M * g.K * 2 <= 64 * 1024 && g.M <= GEMM_FUSED_NORM_MAX_M;
}

template <int MT, int EPI, bool NT, int AMODE>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    const int KT = g.K / 32, NTILES = g.N / 16;
    // NTB: n-tiles per block (SwiGLU needs the (gate, up) pair in one block)
    const bool swiglu = EPI == EPI_SWIGLU;
    int ntb = swiglu ? 2 : 1;  // measured (profiles/gemv_sweep.py): one n-tile per workgroup wins for every plain shape, lm_head included
    if (g_tune_ntb && MT == 1 && (!swiglu || g_tune_ntb % 2 == 0)) ntb = g_tune_ntb;
    const int blocks_x = (NTILES + ntb - 1) / ntb;
    const int blocks_y = (g.M + MT * 16 - 1) / (MT * 16);
    // waves per block: enough waves chip-wide to cover HBM latency (>= ~2048), bounded by K-tiles and LDS
    long blocks = (long)blocks_x * blocks_y * g.batch;
    int W = 4;
    while (W < 16 && blocks * W < 2048 && W * 2 * GEMM_UNROLL <= KT * 2) W *= 2;
    if (g_tune_w) W = g_tune_w;
    while (W > 1 && W > KT) W /= 2;
    while (W > 1 && (size_t)W * MT * ntb * 1024 > 64 * 1024) W /= 2;
    size_t lds = (size_t)W * MT * ntb * 1024;
    if (AMODE >= 1) {  // rows + partial sums, overlaid by the reduction buffer after the k-loop
        const size_t rows = (size_t)g.M * g.K * 2 + (W + 1) * 16 * sizeof(float);
        lds = g.M > 2 ? (lds > rows ? lds : rows) : lds + rows;
    }
    dim3 grid(blocks_x, blocks_y, g.batch), block(W * 64);
    if constexpr (MT == 1) {
        if (ntb == 4) {
            auto kern = gemm_skinny_kernel<1, 4, EPI, NT, AMODE>;
            if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
            hipLaunchKernelGGL(kern, grid, block, lds, stream, g);
            return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
        }
    }
    if (ntb == 2) {
        auto kern = gemm_skinny_kernel<MT, 2, EPI, NT, AMODE>;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        hipLaunchKernelGGL(kern, grid, block, lds, stream, g);
    } else {
        auto kern = gemm_skinny_kernel<MT, (EPI == EPI_SWIGLU ? 2 : 1), EPI, NT, AMODE>;
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        hipLaunchKernelGGL(kern, grid, block, lds, stream, g);
    }
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

template <int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t stream) {
    if (g.norm_w) {
        // fused norm: decode shapes only; every block re-normalises its rows, free at M <= 8 and wasteful beyond
        // (callers run the norm kernel first for larger M)
        if constexpr (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_F32) {
            if (gemm_can_stage(g)) return launch_cfg<1, EPI, true, 2>(g, stream);
        }
        return ISST_ERR_ARG;
    }
    // (plain LDS staging without the norm, AMODE 1, measured 1.2 us slower per launch than direct A loads: not used)
    const bool single = g.M <= 64 && g.batch == 1;  // weight read exactly once -> non-temporal loads
    if (!single) return launch_cfg<4, EPI, false, 0>(g, stream);
    if (g.M <= 16) return launch_cfg<1, EPI, true, 0>(g, stream);
    if (g.M <= 32) return launch_cfg<2, EPI, true, 0>(g, stream);
    if (g.M <= 48) return launch_cfg<3, EPI, true, 0>(g, stream);
    return launch_cfg<4, EPI, true, 0>(g, stream);
}

int launch_gemm(const GemmArgs& g, hipStream_t stream) {
    if (g.M <= 0 || g.batch <= 0) return ISST_OK;
    if (g.K % 32 != 0 || g.N % 16 != 0 || g.lda % 8 != 0) return ISST_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(g.A) & 15) || (reinterpret_cast<uintptr_t>(g.Wp) & 15)) return ISST_ERR_ARG;
    if (gemm_mid_supported(g) && gemm_mid_preferred(g) && !g_force_skinny) {
        if (g.epi == EPI_PARTIAL ? g.ksplit < 1 : g.ksplit > 1) return ISST_ERR_ARG;
        return launch_gemm_mid(g, stream);
    }
    if (gemm_tiled_supported(g) && !g_force_skinny) {
        const bool ok = (g.epi != EPI_BIAS && g.epi != EPI_BIAS_GELU && g.epi != EPI_BIAS_RES) || g.bias;
        if (g.epi == EPI_PARTIAL ? g.ksplit < 1 : g.ksplit > 1) return ISST_ERR_ARG;
        if (!ok || ((g.epi == EPI_RES || g.epi == EPI_BIAS_RES) && !g.res) || (g.epi == EPI_SWIGLU && g.N % 32)) return ISST_ERR_ARG;
        return launch_gemm_tiled(g, stream);
    }
    switch (g.epi) {
        case EPI_NONE: return launch_epi<EPI_NONE>(g, stream);
        case EPI_BIAS: return g.bias ? launch_epi<EPI_BIAS>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_GELU: return g.bias ? launch_epi<EPI_BIAS_GELU>(g, stream) : ISST_ERR_ARG;
        case EPI_RES: return g.res ? launch_epi<EPI_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_RES: return (g.res && g.bias) ? launch_epi<EPI_BIAS_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_SWIGLU: return (g.N % 32 == 0) ? launch_epi<EPI_SWIGLU>(g, stream) : ISST_ERR_ARG;
        case EPI_F32: return launch_epi<EPI_F32>(g, stream);
    }
    return ISST_ERR_ARG;
}
