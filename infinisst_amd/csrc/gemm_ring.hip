// Packed-weight GEMM for 33..64 rows on gfx950:  out[M,N] = epi(A[M,K] @ W[N,K]^T), weights read exactly once -- weights to REGISTERS, activations by LDS-DMA.
//
// Where it runs: every decode pass of a 33..64-stream batch (one row per stream against q/k/v, o_proj, gate/up, down_proj, lm_head: reference
// patch_llm.py:260-262,334, HF LlamaMLP [3P], model/llm.py:237) and the prefill of a 34..64-row prompt.  Weight-streaming problems: a CU has to keep
// ~100 KB of weight loads in flight, and what else it does must not get in the way of that.
//   * gemm_mid.hip (13..64 rows until now) stages the activations through VGPRs and ds_write_b128 (79 B/clk per CU) behind a barrier per 256-deep
//     chunk: all its waves load, write, read and multiply in lockstep, and the phases add up instead of overlapping (q/k/v, o_proj, down_proj took
//     2.8 / 3.3 / 1.7 x their weight stream at 64 rows, profiles/r02/trace_busy_prof64.txt);
//   * a first form of this file put EVERY operand through an LDS-DMA ring: slower still -- an LDS ring holds ~72 KB in flight per CU, the
//     stream needs more (profiles/r03/gemm_ring_probe.txt).
// Here the two kinds of load are split over two kinds of wave, so that each wave's vmcnt queue holds ONE kind and nothing has to be counted by hand
// (hipcc drains the queue when it sees register-destined loads beside LDS-DMA in one wave, cdna_hip_programming.md section 5 item 4b):
//   * CONSUMER wave w (NP of them) owns n-tile pair w of the workgroup's 64 / 96 / 128 columns, all rows (4 m-tiles) and the whole K slice: its weight
//     fragments (1 KiB each, exactly as the packed weights lie in memory) stream through a statically indexed REGISTER ring RING_R = 8 K-tiles deep --
//     32 KB in flight per wave, bounds-checked buffer loads (past the slice: zeros, no traffic, no branch), non-temporal -- the structure of the
//     skinny GEMV (gemm.hip), where the compiler emits counted vmcnt waits; 8 accumulators, no cross-wave reduction;
//   * ONE LOADER wave stages the activations: per 64-deep K-tile 8 units of 8 rows x 128 B (full lines; 16-byte chunk index XORed with
//     ((row >> 1) & 7) on the source side and on the read side: conflict-free ds_read_b128, as gemm_dense.hip) by global_load_lds_dwordx4 into a ring
//     RING_D K-tiles deep, with a counted vmcnt in front of the one raw s_barrier per K-tile;
//   * NORM (the consumer half of the launch-free residual + RMSNorm, GemmArgs::ssq): between the barrier and the MFMAs of K-tile t the consumers rewrite
//     K-tile t+1 of the raw ring as bf16(w * bf16(x / rms)) (HF LlamaRMSNorm's rounding points) into a double-buffered image of the same layout;
//   * epilogues and the in-launch split-K reduction (tickets) are gemm_mid.hip's, shared through mid_epilogue.h.
#include "common.h"
#include "mid_epilogue.h"

#define RING_K 64
#define RING_R 8   // K-tiles of weights in flight per consumer wave (register ring)

typedef __attribute__((address_space(3))) void* rlds_ptr;

template <int N>
__device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NP, int EPI, bool NORM>
__global__ __launch_bounds__((NP + 1) * 64, 1) void gemm_ring_kernel(GemmArgs g, int T /* K-tiles per slice, a multiple of RING_R */) {
    constexpr int D = 6;                         // activation ring depth (K-tiles)
    constexpr int ASLOT = 8 * 1024;              // bytes per ring slot: 8 units of 8 rows x 128 B
    constexpr int A_RING = 0, F_BUF = D * ASLOT, TAIL = F_BUF + (NORM ? 2 * ASLOT : 0);
    constexpr int WN = 2 * NP, TILES = 4 * WN, NW = NP + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object: activation ring | normalised images | 1/rms | norm weight; later the epilogue's buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KT = g.K >> 5, NTILES = g.N >> 4;
    const long k0 = (long)blockIdx.y * T * RING_K;   // first K element of the slice
    float* rsS = reinterpret_cast<float*>(smem + TAIL);          // [64]
    bf16_t* nwS = reinterpret_cast<bf16_t*>(smem + TAIL + 256);   // [T * 64]
    float* red = reinterpret_cast<float*>(smem);
    f32x4_t acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    if (wave == NP) {
        // ================================ loader wave: activations, LDS-DMA only ================================
        const bf16_t* asrc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = u * 8 + (lane >> 3), rim = row & 15;
            const int chunk = (lane & 7) ^ ((rim >> 1) & 7);
            asrc[u] = g.A + (long)min(row, g.M - 1) * g.lda + k0 + chunk * 8;   // rows >= M re-read row M - 1 (never stored)
        }
        auto issue = [&](int t) {  // K-tile t -> ring slot t % D (past the slice: K-tile T - 1 again, into a slot nobody reads any more)
            const int ts = t < T ? t : T - 1;
            unsigned char* slot = smem + A_RING + (t % D) * ASLOT;
#pragma unroll
            for (int u = 0; u < 8; ++u) __builtin_amdgcn_global_load_lds((const void*)(asrc[u] + (long)ts * RING_K), (rlds_ptr)(slot + u * 1024), 16, 0, 0);
        };
#pragma unroll
        for (int t = 0; t < D - 1; ++t) issue(t);
        if constexpr (NORM) {
            ring_wait<(D - 2) * 8>();          // K-tile 0 landed
            __builtin_amdgcn_s_barrier();      // P0: the consumers may rewrite it (and have written 1/rms and the norm weight)
        }
        for (int t = 0; t < T; ++t) {
            ring_wait<(NORM ? D - 3 : D - 2) * 8>();   // K-tile t (NORM: t + 1 as well) landed
            __builtin_amdgcn_s_barrier();               // B_t: visible to the consumers; every read of K-tile t - 1 is over
            issue(t + D - 1);                           // into the slot K-tile t - 1 occupied
        }
        ring_wait<0>();   // the over-issued DMAs must have landed before the ring is reused by the epilogue
    } else {
        // ================================ consumer waves: weights to registers, MFMA ================================
        const int fr = lane & 15, fq = lane >> 4;
        const int nt = blockIdx.x * WN + wave * 2;
        // one descriptor per n-tile over exactly this slice of it: fragments past the slice (the ring runs RING_R K-tiles ahead) and n-tiles past N read
        // zeros without traffic and without a branch
        __amdgpu_buffer_rsrc_t wrs[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const bool valid = nt + nb < NTILES;
            wrs[nb] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.Wp) + ((long)(valid ? nt + nb : 0) * KT + (k0 >> 5)) * 512, 0, valid ? T * 2048 : 0, 0x00020000);
        }
        u32x4_t wr[RING_R][2][2];   // [ring slot][n-tile][k-step]
        auto load_w = [&](int t, int d) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int k = 0; k < 2; ++k) wr[d][nb][k] = __builtin_amdgcn_raw_buffer_load_b128(wrs[nb], (unsigned)(lane * 16 + k * 1024 + t * 2048), 0, 2 /* nt: read once */);
        };
        // (program order pinned: left to itself the scheduler fills the ring back to front, slot 0's load becomes the YOUNGEST at loop entry and the
        //  compiler's wait in front of the first MFMA -- merged over preheader and back edge -- turns into vmcnt(0): a full drain per trip)
#pragma unroll
        for (int d = 0; d < RING_R; ++d) {
            load_w(d, d);
            __builtin_amdgcn_sched_barrier(0);
        }

        if constexpr (NORM) {
            // 1/rms per row from the producer's sums of squares (fixed order: gemm_mid.hip's tree), the norm weight of this K slice into LDS
            for (int row = tid >> 2; row < 64; row += NP * 16) {
                const int q = tid & 3;
                float pq = 0.f;
                if (row < g.M) {
                    const int per = g.ssq_n >> 2;
                    const float* sp = g.ssq + (long)row * g.ssq_n + q * per;
                    if ((per & 3) == 0) {
                        for (int i0 = 0; i0 < per; i0 += 16) {  // a group of loads in flight before the first add
                            f32x4_t t4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) t4[u] = (i0 + 4 * u < per) ? *reinterpret_cast<const f32x4_t*>(sp + i0 + 4 * u) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int u = 0; u < 4; ++u) { pq += t4[u].x; pq += t4[u].y; pq += t4[u].z; pq += t4[u].w; }
                        }
                    } else {
                        for (int i = 0; i < per; ++i) pq += sp[i];
                    }
                }
                const float o1 = __shfl_xor(pq, 1, WAVE);
                const float s2 = (q & 1) ? o1 + pq : pq + o1;   // (even + odd), the same operand order in both lanes
                const float o2 = __shfl_xor(s2, 2, WAVE);
                const float tot = (q & 2) ? o2 + s2 : s2 + o2;
                if (q == 0) rsS[row] = row < g.M ? rsqrtf(tot / g.K + g.norm_eps) : 0.f;
            }
            for (int i = tid; i < T * 8; i += NP * 64) *reinterpret_cast<u32x4_t*>(nwS + i * 8) = *reinterpret_cast<const u32x4_t*>(g.norm_w + k0 + i * 8);
        }
        // raw K-tile t (ring) -> normalised image F[t & 1]: consumer w rewrites units w, w + NP, ...
        auto normalise = [&](int t) {
            const unsigned char* raw = smem + A_RING + (t % D) * ASLOT;
            unsigned char* f = smem + F_BUF + (t & 1) * ASLOT;
#pragma unroll
            for (int a = 0; a < (8 + NP - 1) / NP; ++a) {
                const int unit = wave + a * NP;
                if (unit < 8) {
                    const int row = unit * 8 + (lane >> 3), rim = row & 15;
                    const int chunk = (lane & 7) ^ ((rim >> 1) & 7);   // the K chunk this lane's 16 bytes hold
                    const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(raw + unit * 1024 + lane * 16);
                    const u32x4_t nwv = *reinterpret_cast<const u32x4_t*>(nwS + t * RING_K + chunk * 8);
                    const float rs = rsS[row];
                    u32x4_t o;
                    o.x = pack_bf(lo_bf(nwv.x) * bfr(lo_bf(xv.x) * rs), hi_bf(nwv.x) * bfr(hi_bf(xv.x) * rs));
                    o.y = pack_bf(lo_bf(nwv.y) * bfr(lo_bf(xv.y) * rs), hi_bf(nwv.y) * bfr(hi_bf(xv.y) * rs));
                    o.z = pack_bf(lo_bf(nwv.z) * bfr(lo_bf(xv.z) * rs), hi_bf(nwv.z) * bfr(hi_bf(xv.z) * rs));
                    o.w = pack_bf(lo_bf(nwv.w) * bfr(lo_bf(xv.w) * rs), hi_bf(nwv.w) * bfr(hi_bf(xv.w) * rs));
                    *reinterpret_cast<u32x4_t*>(f + unit * 1024 + lane * 16) = o;
                }
            }
        };
        const int a_rd = (fr >> 3) * 1024 + (fr & 7) * 128, a_sw = (fr >> 1) & 7;
        if constexpr (NORM) {
            __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's rsS / nwS stores
            __builtin_amdgcn_s_barrier();          // P0 (the loader has waited for K-tile 0)
            normalise(0);
        }
        for (int t0 = 0; t0 < T; t0 += RING_R) {
#pragma unroll
            for (int d = 0; d < RING_R; ++d) {
                const int t = t0 + d;
                if constexpr (NORM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's image writes of K-tile t
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();          // B_t
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* abase = (NORM ? smem + F_BUF + (t & 1) * ASLOT : smem + A_RING + (t % D) * ASLOT) + a_rd;
                u32x4_t fa[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int k = 0; k < 2; ++k) fa[mt][k] = *reinterpret_cast<const u32x4_t*>(abase + mt * 2048 + (((k * 4 + fq) ^ a_sw) << 4));
                if constexpr (NORM) {
                    if (t + 1 < T) normalise(t + 1);  // (its raw rows landed before B_t; F[(t + 1) & 1] was last read for K-tile t - 1)
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb)
                            acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[mt][k]), __builtin_bit_cast(bf16x8_t, wr[d][nb][k]), acc[mt][nb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                load_w(t + RING_R, d);   // refill the slot just consumed (past the slice: zeros)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    __syncthreads();   // every read of the rings is over, every DMA has landed

    // ---- hand the accumulators to the shared epilogue: red[0][(mt * WN + n) * 4 + r][lane] ----
    if (wave < NP) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((mt * WN + wave * 2 + nb) * 4 + r) * 64 + lane] = acc[mt][nb][r];
    }
    __syncthreads();
    mid_epilogue<4, WN, 1, NW, EPI>(g, red, tid, 0, reinterpret_cast<int*>(smem + TILES * 1024));
}

static int g_ring_mode = 1;  // tuning hook (gemm_ring_set): 0 = never (gemm_mid.hip runs), 1 = default
void gemm_ring_set(int mode) { g_ring_mode = mode; }

template <int NP, bool NORM>
static size_t ring_lds(int T) {
    size_t b = (size_t)6 * 8192 + (NORM ? 2 * 8192 : 0);
    if (NORM) b += 256 + (size_t)T * RING_K * 2;
    const size_t epi = (size_t)4 * 2 * NP * 1024 + 16;
    return b > epi ? b : epi;
}

bool gemm_ring_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    if (g_ring_mode == 0 || g.batch != 1 || g.M <= 32 || g.M > 64 || g.attn_partial) return false;
    if (g.K % (RING_K * RING_R * ks) != 0 || g.N % 16 != 0 || g.lda % 8 != 0) return false;
    if (g.norm_w && !(g.ssq && g.ssq_n % 4 == 0 && g.ssq_n * 32 == g.K && g.K / ks <= 8192)) return false;
    if (g.tickets && !(g.epi == EPI_PARTIAL && g.res && g.N % 32 == 0 && (g.reduce_plain || !g.norm_w))) return false;
    if (g.epi == EPI_SWIGLU && g.N % 32 != 0) return false;
    return true;
}

template <int NP, int EPI, bool NORM>
static int launch_ring3(const GemmArgs& g, hipStream_t stream) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const int T = g.K / RING_K / ks;
    const int NTILES = g.N / 16;
    dim3 grid((NTILES + 2 * NP - 1) / (2 * NP), ks), block((NP + 1) * 64);
    const size_t lds = ring_lds<NP, NORM>(T);
    static size_t attr_lds = 0;
    if (lds > 64 * 1024 && lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ring_kernel<NP, EPI, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        attr_lds = lds;
    }
    hipLaunchKernelGGL((gemm_ring_kernel<NP, EPI, NORM>), grid, block, lds, stream, g, T);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
template <int NP, int EPI>
static int launch_ring2(const GemmArgs& g, hipStream_t stream) {
    if constexpr (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_F32 || EPI == EPI_PARTIAL) {
        if (g.norm_w) return launch_ring3<NP, EPI, true>(g, stream);
    }
    return launch_ring3<NP, EPI, false>(g, stream);
}
static int g_ring_np = 0;  // profiling aid: force NP
void gemm_ring_set_np(int np) { g_ring_np = np; }
// columns per workgroup: the widest that still gives (nearly) every CU a workgroup
template <int EPI>
static int launch_ring1(const GemmArgs& g, hipStream_t stream) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const long pairs = g.N / 32;
    int np = 2;
    if (pairs / 4 * ks >= 200) np = 4;
    else if (pairs % 3 == 0 && pairs / 3 * ks >= 200) np = 3;
    if (g_ring_np >= 2 && g_ring_np <= 4) np = g_ring_np;
    if (np == 4) return launch_ring2<4, EPI>(g, stream);
    if (np == 3) return launch_ring2<3, EPI>(g, stream);
    return launch_ring2<2, EPI>(g, stream);
}

int launch_gemm_ring(const GemmArgs& g, hipStream_t stream) {
    if (!gemm_ring_supported(g)) return ISST_ERR_ARG;
    if (g.ksplit > 1 && g.epi != EPI_PARTIAL) return ISST_ERR_ARG;
    switch (g.epi) {
        case EPI_NONE: return launch_ring1<EPI_NONE>(g, stream);
        case EPI_BIAS: return g.bias ? launch_ring1<EPI_BIAS>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_GELU: return g.bias ? launch_ring1<EPI_BIAS_GELU>(g, stream) : ISST_ERR_ARG;
        case EPI_RES: return g.res ? launch_ring1<EPI_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_RES: return (g.res && g.bias) ? launch_ring1<EPI_BIAS_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_SWIGLU: return launch_ring1<EPI_SWIGLU>(g, stream);
        case EPI_F32: return launch_ring1<EPI_F32>(g, stream);
        case EPI_PARTIAL: return launch_ring1<EPI_PARTIAL>(g, stream);
    }
    return ISST_ERR_ARG;
}
