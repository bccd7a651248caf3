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
//     buffer_load_dwordx4 per wave -- no LDS round trip, no 64-byte row fragments;
//   * a workgroup owns NTB n-tiles and all of K; its waves interleave k-tiles (wave w takes kt = w, w+W, ..),
//     so the block streams one contiguous [NTB][K/32] KiB run; partial sums meet in LDS once at the end;
//   * A (activations, L2-resident) goes straight to VGPRs in the A-operand layout through a bounds-checked descriptor (rows >= M read
//     zeros), in the same ring of loads as the weights; after an RMSNorm and at <= 8 rows it is staged and normalised in LDS instead;
//   * the epilogue applies bias / GELU / residual / SwiGLU with the reference's bf16 rounding points.
//
// Operand maps (cdna_hip_programming.md section 3): A lane l holds A[row l&15][k = 8(l>>4)+j], B lane l holds
// B[k = 8(l>>4)+j][col l&15], C/D lane l reg r is (row 4(l>>4)+r, col l&15).
#include "common.h"

// The weight stream is a statically indexed ring: GEMM_DEPTH stages of GEMM_UNR k-tiles per wave are in flight; every load is an
// unconditional bounds-checked buffer load (k-tiles past K and n-tiles past N read zeros), so the loop has no branch around a load
// and hipcc emits counted s_waitcnt vmcnt(n) -- with `cond ? load : 0` it emitted a branch per load, register copies for the
// rotation and a vmcnt(0) per iteration, which cost 2.3 .. 3.7 us per launch against a pure streaming read of the same bytes
// (profiles/probes/gemv_fixed_probe.hip: gate/up 39.3 -> 37.0 us, down 22.0 -> 18.3 us, o_proj 9.1 -> 6.6 us for the loads + MFMAs).
#define GEMM_UNR 2
#define GEMM_DEPTH 2
#define GEMM_DEPTH_STAGED 4  // AMODE >= 1: the ring also has to cover the staging / norm prologue
// (measured and dropped: the whole K extent of a wave in flight at once, 16 loads per wave on 8 waves, for the 384-workgroup q/k/v
//  launch: 15.1 us against 11.1-12.1 us -- more requests per wave than the memory pipeline takes without stalling the others;
//  rows staged in LDS without the norm (AMODE 1) for o_proj / down: 8.1 / 20.6 us against 7.3 / 19.0 us with A in the ring;
//  the residual / bias of the epilogue requested before the ring: 33.14 ms per chunk against 33.04 without, same box)
#define GEMM_OOB 0x80000000u  // + any in-range offset stays past every descriptor's extent (and does not wrap)

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

// Self-paired SwiGLU tiles (EPI_SWIGLU8): source row r of gate_proj (half 0) / up_proj (half 1) becomes row (r % 8) + 8 * half of tile r / 8.
__global__ void pack_weight_half_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int n_rows, int K, int half) {
    const int KT = K >> 5;
    const long tile = blockIdx.x;  // rt * KT + kt, rt over groups of 8 source rows
    const int rt = (int)(tile / KT), kt = (int)(tile % KT);
    const int lane = threadIdx.x;
    if (((lane & 15) >> 3) != half) return;
    const int n = rt * 8 + (lane & 7);
    const int k0 = kt * 32 + 8 * (lane >> 4);
    bf16_t* d = dst + ((long)rt * KT + kt) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = n < n_rows ? src[(long)n * K + k0 + j] : (bf16_t)0;
}
int launch_pack_weight_half(const bf16_t* src, bf16_t* dst, int n_rows, int K, int half, hipStream_t stream) {
    if (K % 32 != 0 || n_rows <= 0 || n_rows % 8 != 0 || (half != 0 && half != 1)) return ISST_ERR_ARG;
    hipLaunchKernelGGL(pack_weight_half_kernel, dim3((unsigned)((long)(n_rows / 8) * (K / 32))), dim3(64), 0, stream, src, dst, n_rows, K, half);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// -DISST_GEMV_TRACE (make trace -> libinfinisst_hip_trace.so, profiles/gemv_trace_probe.py): wave 0 of every workgroup of the skinny kernel
// stamps the 100 MHz wall clock at entry / rows staged / first weight batch consumed / k-loop done / output stored, plus its XCC and CU id.
#ifdef ISST_GEMV_TRACE
#define GEMV_TRACE_SLOTS 8
__device__ unsigned long long g_gemv_trace[16384 * GEMV_TRACE_SLOTS];
// (one 2048-workgroup region per Llama decode projection, so that one graph replay leaves the last layer's four launches side by side)
#define GEMV_REGION (g.N == 6144 ? 0 : g.N == 28672 ? 4096 : g.K == 4096 ? 2048 : 8192)
#define GEMV_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 2048) g_gemv_trace[(GEMV_REGION + blockIdx.x) * GEMV_TRACE_SLOTS + (i)] = wall_clock64(); } while (0)
extern "C" int isst_debug_gemv_trace_read(void* dst, long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gemv_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define GEMV_STAMP(i) do {} while (0)
#endif

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
// AMODE 3: as 1, but the staged rows do not exist in memory yet: they are the Llama attention output, merged here from the split-KV
//          partials of llm_attn_partial_kernel (GemmArgs::attn_partial; common.h attn_merge_pair = the combine kernel's arithmetic) while
//          the weight ring is in flight -- o_proj at one or two decode rows; removes the combine launch between attention and o_proj
//          (4.75 us + a 2.6 us gap per layer and pass, profiles/r01/trace_busy_v9.txt).
template <int MT, int NTB, int EPI, bool NT, int AMODE, int DEPTH, int MS = 0>  // MS: AMODE 3's unroll bound of the split-KV merge (>= attn_splits)
__global__ __launch_bounds__(AMODE == 3 ? 512 : 1024) void gemm_skinny_kernel(GemmArgs g) {
    constexpr int UNR = (MT * NTB >= 8) ? 1 : GEMM_UNR;  // (4 m-tiles x the SwiGLU pair: the two-k-tile stage spilled)
    // AMODE 0: the reduction buffer [W][MT*NTB*4][64].  AMODE >= 1: the staged A rows [M][K] + [W] partial sums during the k-loop;
    // the reduction buffer then REUSES the same bytes (one more barrier) -- kept apart, 4 rows needed 41 KB per workgroup: 3 instead
    // of 4 workgroups per CU and a second round of workgroups for the 896-workgroup gate/up launch (+7 us)
    extern __shared__ __attribute__((aligned(16))) float red[];
    GEMV_STAMP(0);
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
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};

    // descriptors: this workgroup's n-tiles [nt0, nt0 + NTB) clipped to N, and (AMODE 0) its A rows [m0, m0 + 16 MT) clipped to M
    const int my_tiles = min(NTB, NTILES - nt0);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.Wp) + (long)nt0 * KT * 512, 0, my_tiles * KT * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A) + (long)m0 * g.lda, 0, (int)(((min(MT * 16, g.M - m0) - 1) * g.lda + g.K) * 2), 0x00020000);  // (lda < K for the conv-as-GEMM views: rows overlap)
    unsigned woff[NTB], aoff[MT];
#pragma unroll
    for (int nb = 0; nb < NTB; ++nb) woff[nb] = (unsigned)((nb * KT * 64 + lane) * 16);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) aoff[mt] = (unsigned)(((long)(mt * 16 + arow) * g.lda + kq) * 2);

    // stage `batch` (k-tiles wave + (batch * UNR + u) * W) -> ring slot d
    u32x4_t wring[DEPTH][UNR][NTB];
    u32x4_t aring[DEPTH][UNR][AMODE == 0 ? MT : 1];
    auto issue = [&](int batch, int d) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int kt = wave + (batch * UNR + u) * W;
            const bool kv = kt < KT;
            const unsigned wk = kv ? (unsigned)kt * 1024u : GEMM_OOB, ak = kv ? (unsigned)kt * 64u : GEMM_OOB;
#pragma unroll
            for (int nb = 0; nb < NTB; ++nb) wring[d][u][nb] = __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[nb] + wk, 0, NT ? 2 : 0);
            if constexpr (AMODE == 0) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aring[d][u][mt] = __builtin_amdgcn_raw_buffer_load_b128(ars, aoff[mt] + ak, 0, 0);
            }
        }
    };

    // (1-2 rows keep the two LDS areas apart: they fit 4 workgroups per CU anyway and skip the extra barrier)
    const bool overlay = g.M > 2;
    bf16_t* xs = reinterpret_cast<bf16_t*>(overlay ? red : red + (long)W * (MT * NTB * 256));  // [M][K] staged rows (AMODE >= 1)
    // AMODE >= 1: rows are staged (and normalised) side by side: each row gets wpr = max(1, W / M) waves, W / wpr rows per round
    // (one stream: 1 row on all waves; beam search: 4 rows, one wave each, one round)
    const int wpr = (W >= g.M) ? W / g.M : 1;   // waves per row
    const int rpr = W / wpr;                     // rows per round
    const int my_slot = wave / wpr, my_sub = wave % wpr;
    const int cstep = wpr * 512, cfirst = (my_sub * 64 + lane) * 8;
    // where a lane's chunk lies past K it re-reads a chunk that exists (and drops it): its OWN first chunk -- or, with more waves on a row than the row has
    // 512-element groups (16 waves on K = 4096: forced through isst_op_set_gemm_tuning it faulted on the norm weight), the row's first
    const int csafe = cfirst < g.K ? cfirst : 0;
    // the first group of the first round (for one stream: everything) is requested BEFORE the weight ring, so that waiting for it
    // leaves the ring in flight (vmcnt counts in order); the norm weight comes with it instead of after the first barrier
    u32x4_t xv0[4], nw0[4];
    // AMODE 3: wave w merges heads w, w + W, ..: 64 lanes x 2 dims = one head's 128 outputs.  The loads of the wave's first NH heads (all of
    // them at 32 heads on 8 waves; 2 + 2 MS registers per head and lane, which is why this instance is bounded to 512 threads: 256 VGPRs)
    // go out BEFORE the weight ring: vmcnt counts in order, so waiting for them leaves the ring in flight, and the merge arithmetic runs
    // under the weights' HBM latency
    constexpr int MS1 = MS > 0 ? MS : 1;
    constexpr int NH = MS == 0 ? 1 : (MS <= 20 ? 4 : 2);
    AttnMergeLoads<MS1> mg[NH];
    if constexpr (AMODE == 3) {
        const int H = g.K >> 7;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hh = wave + i * W < H ? wave + i * W : (wave < H ? wave : 0);
            attn_merge_issue<MS1>(g.attn_partial + ((long)m0 * H + hh) * g.attn_splits * ATTN_SLAB, g.attn_splits, lane, mg[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (AMODE == 1 || AMODE == 2) {
        const bf16_t* xr = A + (long)(m0 + (my_slot < g.M ? my_slot : 0)) * g.lda;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = cfirst + u * cstep;
            xv0[u] = *reinterpret_cast<const u32x4_t*>(xr + (c < g.K ? c : csafe));
            if constexpr (AMODE == 2) nw0[u] = *reinterpret_cast<const u32x4_t*>(g.norm_w + (c < g.K ? c : csafe));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g.tune & 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
        issue(d, d);
        __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (AMODE == 3) {
        const int H = g.K >> 7;
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int hh = wave + i * W;
            const uint32_t v = attn_merge_finish<MS1>(mg[i], g.attn_splits, lane);
            if (hh < H) *reinterpret_cast<uint32_t*>(xs + hh * 128 + 2 * lane) = v;
        }
        for (int r = 0; r < g.M; ++r)  // further heads / the second row: one more round trip each (behind the ring)
            for (int hh = wave + (r == 0 ? NH * W : 0); hh < H; hh += W) {
                AttnMergeLoads<MS1> m1;
                attn_merge_issue<MS1>(g.attn_partial + ((long)(m0 + r) * H + hh) * g.attn_splits * ATTN_SLAB, g.attn_splits, lane, m1);
                *reinterpret_cast<uint32_t*>(xs + (long)r * g.K + hh * 128 + 2 * lane) = attn_merge_finish<MS1>(m1, g.attn_splits, lane);
            }
        __syncthreads();
    }
    if constexpr (AMODE == 1 || AMODE == 2) {
        float* part = reinterpret_cast<float*>(xs + (long)g.M * g.K);  // [W] partial sums of squares
        auto stage_group = [&](const u32x4_t (&xv)[4], int rr, int c0, float& sq) {
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
        };
        auto norm_group = [&](const u32x4_t (&wv)[4], int rr, int c0, float rs) {  // every lane rewrites the chunks it staged
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
        };
        auto round = [&](int r0, bool first) {
            const int rr = r0 + my_slot;
            const bool active = my_slot < rpr && rr < g.M;
            float sq = 0.f;
            if (active) {
                const bf16_t* xr = A + (long)(m0 + rr) * g.lda;
                if (first) stage_group(xv0, rr, cfirst, sq);
                // (loads go out four at a time: a chunk-by-chunk loop pays one L2 round trip per chunk -- 8 in a row at 4+ rows)
                for (int c0 = first ? cfirst + 4 * cstep : cfirst; c0 < g.K; c0 += 4 * cstep) {
                    u32x4_t xv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = c0 + u * cstep;
                        xv[u] = *reinterpret_cast<const u32x4_t*>(xr + (c < g.K ? c : csafe));
                    }
                    stage_group(xv, rr, c0, sq);
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
                    if (first) norm_group(nw0, rr, cfirst, rs);
                    for (int c0 = first ? cfirst + 4 * cstep : cfirst; c0 < g.K; c0 += 4 * cstep) {
                        u32x4_t wv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int c = c0 + u * cstep;
                            wv[u] = *reinterpret_cast<const u32x4_t*>(g.norm_w + (c < g.K ? c : csafe));
                        }
                        norm_group(wv, rr, c0, rs);
                    }
                }
                __syncthreads();  // part[] is reused by the next round; the last one publishes the rows
            }
        };
        round(0, true);
        for (int r0 = rpr; r0 < g.M; r0 += rpr) round(r0, false);
        if constexpr (AMODE == 1) __syncthreads();
    }

    const bool avalid = arow < g.M;  // (AMODE >= 1: one m-tile, m0 = 0)
#ifdef ISST_GEMV_TRACE
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 2048) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_gemv_trace[(GEMV_REGION + blockIdx.x) * GEMV_TRACE_SLOTS + 5] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
    GEMV_STAMP(1);
    const int batches = (KT + W * UNR - 1) / (W * UNR);
    for (int b0 = 0; b0 < batches; b0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if constexpr (AMODE >= 1) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int kt = wave + ((b0 + d) * UNR + u) * W;
                    aring[d][u][0] = (kt < KT && avalid) ? *reinterpret_cast<const u32x4_t*>(xs + (long)arow * g.K + (long)kt * 32 + kq) : zero4;
                }
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nb = 0; nb < NTB; ++nb)
                        acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, aring[d][u][AMODE == 0 ? mt : 0]),
                                                                             __builtin_bit_cast(bf16x8_t, wring[d][u][nb]), acc[mt][nb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifdef ISST_GEMV_TRACE
            if (b0 == 0 && d == 0) { asm volatile("s_nop 0" :: "v"(acc[0][0][0])); GEMV_STAMP(2); }
#endif
            issue(b0 + d + DEPTH, d);  // past the end: zeros, never used
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    GEMV_STAMP(3);
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
        if constexpr (EPI == EPI_SWIGLU8) { if ((l & 15) >= 8) continue; }  // (column c < 8 of the tile is the gate of output 8 t + c, column c + 8 its up)
        float s = 0.f, s2 = 0.f;
        for (int w = 0; w < W; ++w) {
            const float* p = red + (long)w * (TILES * 256);
            s += p[((mt * NTB + nb) * 4 + r) * 64 + l];
            if constexpr (EPI == EPI_SWIGLU) s2 += p[((mt * NTB + nb + 1) * 4 + r) * 64 + l];
            if constexpr (EPI == EPI_SWIGLU8) s2 += p[((mt * NTB + nb) * 4 + r) * 64 + l + 8];
        }
        const int row = m0 + mt * 16 + (l >> 4) * 4 + r;
        int col;
        if constexpr (EPI == EPI_SWIGLU) col = ((nt0 + nb) >> 1) * 16 + (l & 15);
        else if constexpr (EPI == EPI_SWIGLU8) col = (nt0 + nb) * 8 + (l & 15);
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
            else /* EPI_SWIGLU, EPI_SWIGLU8 */ v = bfr(silu(bfr(s))) * bfr(s2);
            reinterpret_cast<bf16_t*>(g.out)[b * g.out_batch + (long)row * g.ldo + col] = f2bf(v);
        }
    }
    GEMV_STAMP(4);
}

// tuning overrides for profiles/gemv_sweep.py (0 = heuristic)
static int g_tune_w = 0, g_force_skinny = 0, g_tune_merge_w = 0, g_tune_flags = 0;
// w: waves per workgroup of the skinny kernel (0 = heuristic, < 0 = never use the mid / tiled kernels); ntb: passed on to gemm_mid (its width / timing knobs)
void gemm_set_tuning(int w, int ntb) { if (w >= 900000) { gemm_wide_set((w - 900000) / 10, (w - 900000) % 10); return; } if (w >= 800000) { gemm_dense_set(w - 800000); return; } if (w >= 700000) { g_tune_flags = w - 700000; return; } if (w >= 300000) { g_tune_merge_w = w - 300000; return; } if (w >= 200000) { gemm_mid_set_min_rows(w - 200000); return; } if (w >= 100000) { gemm_tiled_set_raster(w - 100000); return; } g_tune_w = w < 0 ? 0 : w; g_force_skinny = w < 0; gemm_mid_set_tuning(ntb); }

static inline bool gemm_can_stage(const GemmArgs& g) {
    return g.batch == 1 && g.M <= 16 && (size_t)g.M * g.K * 2 <= 64 * 1024 && g.M <= GEMM_FUSED_NORM_MAX_M;
}

template <int MT, int EPI, bool NT, int AMODE, int MS = 0>
static int launch_cfg(const GemmArgs& g, hipStream_t stream) {
    const int KT = g.K / 32, NTILES = g.N / 16;
    // NTB: n-tiles per block (SwiGLU needs the (gate, up) pair in one block)
    const bool swiglu = EPI == EPI_SWIGLU;
    const int ntb = swiglu ? 2 : 1;  // measured (profiles/gemv_sweep.py): one n-tile per workgroup wins for every plain shape, lm_head included
    const int blocks_x = (NTILES + ntb - 1) / ntb;
    const int blocks_y = (g.M + MT * 16 - 1) / (MT * 16);
    // waves per block: enough waves chip-wide to cover HBM latency (>= ~2048), bounded by K-tiles and LDS
    long blocks = (long)blocks_x * blocks_y * g.batch;
    int W = 4;
    while (W < 16 && blocks * W < 1536 && W * 2 * GEMM_UNR * GEMM_DEPTH <= KT * 2) W *= 2;  // (384 q/k/v workgroups: 4 waves 11.1 us, 8 waves 11.7-12.2 us)
    // 3+ rows through the fused norm (beam search): with 4 waves each wave stages and normalises a whole row (8 chunks per lane, two
    // dependent groups of loads); 8 waves give every row two waves and one group (beam 4: 36.13 -> 35.99 ms per chunk, same box)
    if (AMODE == 2 && g.M > 2 && ntb == 1 && W < 8 && KT >= 64) W = 8;
    if (AMODE == 3) W = g_tune_merge_w ? (g_tune_merge_w > 8 ? 8 : g_tune_merge_w) : 8;  // 32 heads: 4 per wave, all in flight before the weight ring
    if (g_tune_w) W = g_tune_w;
    while (W > 1 && W > KT) W /= 2;
    while (W > 1 && (size_t)W * MT * ntb * 1024 > 64 * 1024) W /= 2;
    size_t lds = (size_t)W * MT * ntb * 1024;
    if (AMODE >= 1) {  // rows + partial sums, overlaid by the reduction buffer after the k-loop
        const size_t rows = (size_t)g.M * g.K * 2 + (W + 1) * 16 * sizeof(float);
        lds = g.M > 2 ? (lds > rows ? lds : rows) : lds + rows;
    }
    dim3 grid(blocks_x, blocks_y, g.batch), block(W * 64);
    GemmArgs gt = g;
    gt.tune = g_tune_flags;
    auto go = [&](auto kern) {
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        hipLaunchKernelGGL(kern, grid, block, lds, stream, gt);
        return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    };
    constexpr int NTB = EPI == EPI_SWIGLU ? 2 : 1;
    if constexpr (AMODE == 2) { if (g_tune_flags & 2) return go(gemm_skinny_kernel<MT, NTB, EPI, NT, AMODE, 2, MS>); }
    return go(gemm_skinny_kernel<MT, NTB, EPI, NT, AMODE, (AMODE >= 1 ? GEMM_DEPTH_STAGED : GEMM_DEPTH), MS>);
}

template <int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t stream) {
    if (g.attn_partial) {  // o_proj of one or two decode rows with the split-KV merge as its A-staging prologue
        if constexpr (EPI == EPI_RES) {
            if (g.batch == 1 && g.M <= ATTN_MERGE_MAX_ROWS && g.K % 128 == 0 && g.attn_splits >= 1 && g.attn_splits <= ATTN_MERGE_MAX_SPLITS && !g.norm_w &&
                (size_t)g.M * g.K * 2 <= 64 * 1024) {
                // unroll bound of the merge = registers held per lane while its loads are in flight: the smallest instance that covers the splits
                if (g.attn_splits <= 8) return launch_cfg<1, EPI, true, 3, 8>(g, stream);
                if (g.attn_splits <= 16) return launch_cfg<1, EPI, true, 3, 16>(g, stream);
                if (g.attn_splits <= 20) return launch_cfg<1, EPI, true, 3, 20>(g, stream);
                return launch_cfg<1, EPI, true, 3, 32>(g, stream);
            }
        }
        return ISST_ERR_ARG;
    }
    if (g.norm_w) {
        // fused norm: decode shapes only; every block re-normalises its rows, free at M <= 8 and wasteful beyond
        // (callers run the norm kernel first for larger M)
        if constexpr (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_SWIGLU8 || EPI == EPI_F32) {
            if (gemm_can_stage(g)) return launch_cfg<1, EPI, true, 2>(g, stream);
        }
        return ISST_ERR_ARG;
    }
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
    if ((!g.attn_partial && (reinterpret_cast<uintptr_t>(g.A) & 15)) || (reinterpret_cast<uintptr_t>(g.Wp) & 15)) return ISST_ERR_ARG;
    if (!g.attn_partial && gemm_mid_supported(g) && gemm_mid_preferred(g) && !g_force_skinny && !(gemm_wide_supported(g) && gemm_wide_preferred(g))) {
        if (g.epi == EPI_PARTIAL ? g.ksplit < 1 : g.ksplit > 1) return ISST_ERR_ARG;
        return launch_gemm_mid(g, stream);
    }
    // 65..256 rows against a long weight stream (decode passes of many streams / streams x beams): weights read once, deep register rings
    if (!g_force_skinny && gemm_wide_supported(g) && gemm_wide_preferred(g)) return launch_gemm_wide(g, stream);
    if (!g.attn_partial && gemm_tiled_supported(g) && !g_force_skinny) {
        const bool ok = (g.epi != EPI_BIAS && g.epi != EPI_BIAS_GELU && g.epi != EPI_BIAS_RES) || g.bias;
        if (g.epi == EPI_PARTIAL ? g.ksplit < 1 : g.ksplit > 1) return ISST_ERR_ARG;
        if (!ok || ((g.epi == EPI_RES || g.epi == EPI_BIAS_RES) && !g.res) || (g.epi == EPI_SWIGLU && g.N % 32)) return ISST_ERR_ARG;
        if (gemm_dense_supported(g) && gemm_dense_preferred(g)) return launch_gemm_dense(g, stream);
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
        case EPI_SWIGLU8: return (g.M <= 16 && g.batch == 1) ? launch_epi<EPI_SWIGLU8>(g, stream) : ISST_ERR_ARG;  // (the register-A GEMV only: the other kernels pair tiles)
    }
    return ISST_ERR_ARG;
}
