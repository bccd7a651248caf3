// Packed-weight GEMM for 33..64 rows on gfx950, all operands through an LDS-DMA ring:  out[M,N] = epi(A[M,K] @ W[N,K]^T), weights read exactly once.
//
// Where it runs: every decode pass of a 33..64-stream batch (one row per stream against q/k/v, o_proj, gate/up, down_proj, lm_head: reference
// patch_llm.py:260-262,334, HF LlamaMLP [3P], model/llm.py:237) and the prefill of a 34..64-row prompt.  These are weight-streaming problems: the
// kernel has to keep ~50 KB of loads in flight per CU and nothing else matters much (at the HBM rate the matrix pipes are 20 % busy).
// gemm_mid.hip -- which this replaces from 33 rows on -- stages the activations through VGPRs and ds_write_b128 (79 B/clk per CU) behind a barrier per
// 256-deep chunk and holds its weight fragments in two register sets: its q/k/v, o_proj and down_proj launches took 2.8 / 3.3 / 1.7 x their weight
// stream at 64 rows (profiles/r02/trace_busy_prof64.txt).  Here, after cdna_hip_programming.md section 5 ("glds vs register staging", "x through LDS in
// full lines") and the ring of MI355X_MICROARCH.md's ring-gemm row:
//   * a workgroup owns NP n-tile pairs (64 / 96 / 128 columns), all rows (4 m-tiles) and one K slice; wave w owns pair w for the whole slice:
//     8 accumulators, no cross-wave reduction;
//   * a K-TILE is 64 deep.  Per K-tile the workgroup stages A (8 units of 8 rows x 128 B, full lines, 16-byte chunk index XORed with
//     ((row >> 1) & 7) on the source side and on the read side: conflict-free ds_read_b128, as gemm_dense.hip) into a SHARED ring and every wave
//     its own 4 weight fragments (2 n-tiles x 2 k-steps, 1 KiB each exactly as the packed weights lie in memory) into a PRIVATE ring -- all by
//     global_load_lds_dwordx4: no staging registers, no ds_write pass, nothing the compiler could drain;
//   * RING_D K-tiles deep: while K-tile t is multiplied, t+1 .. t+RING_D-2 are in flight or landed (24 KiB per wave at depth 4); one counted
//     s_waitcnt vmcnt + one raw s_barrier per K-tile (the barrier publishes the A parts of K-tile t and frees the slot of K-tile t-1 for the DMA
//     issued right behind it);
//   * NORM (the consumer half of the launch-free residual + RMSNorm, GemmArgs::ssq): the raw rows land in the ring, and between the barrier and
//     the MFMAs of K-tile t every wave rewrites its share of K-tile t+1 as bf16(w * bf16(x / rms)) (HF LlamaRMSNorm's rounding points) into a
//     double-buffered image of the same layout -- still one barrier per K-tile;
//   * epilogues and the in-launch split-K reduction (tickets) are gemm_mid.hip's, shared through mid_epilogue.h.
#include "common.h"
#include "mid_epilogue.h"

#define RING_K 64

typedef __attribute__((address_space(3))) void* rlds_ptr;

template <int N>
__device__ __forceinline__ void ring_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int NP, int EPI, bool NORM>
__global__ __launch_bounds__(NP * 64, 1) void gemm_ring_kernel(GemmArgs g, int T /* K-tiles per slice */) {
    constexpr int D = NORM ? 5 : 4;              // ring depth (NORM consumes one K-tile earlier: its raw rows are rewritten one iteration ahead)
    constexpr int AU = (8 + NP - 1) / NP;        // A units a wave stages per K-tile (units past the 8 real ones go to a dummy KiB: every wave issues the same count)
    constexpr int PER = AU + 4;                  // DMA instructions per wave and K-tile
    constexpr int ASLOT = 8 * 1024 + 1024;       // bytes per A ring slot: 8 units + the dummy
    constexpr int WSLOT = 4 * 1024;              // bytes per (wave, K-tile) of weights
    constexpr int A_RING = 0, W_RING = D * ASLOT, F_BUF = W_RING + NP * D * WSLOT, TAIL = F_BUF + (NORM ? 2 * 8192 : 0);
    constexpr int WN = 2 * NP, TILES = 4 * WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the ONLY LDS object (a second one makes hipcc drain the DMAs): rings | F | norm weight | 1/rms | ticket
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int KT = g.K >> 5, NTILES = g.N >> 4;
    const int nt = blockIdx.x * WN + wave * 2;
    const long k0 = (long)blockIdx.y * T * RING_K;   // first K element of the slice

    // ---- DMA sources ----
    const bf16_t* asrc[AU];
    int adst[AU];
#pragma unroll
    for (int a = 0; a < AU; ++a) {
        const int unit = wave + a * NP;
        const int u = unit < 8 ? unit : 0;                // dummy units re-read unit 0
        const int row = u * 8 + (lane >> 3), rim = row & 15;
        const int chunk = (lane & 7) ^ ((rim >> 1) & 7);
        asrc[a] = g.A + (long)min(row, g.M - 1) * g.lda + k0 + chunk * 8;   // rows >= M re-read row M - 1 (never stored)
        adst[a] = (unit < 8 ? unit : 8) * 1024;
    }
    const bf16_t* wsrc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) wsrc[nb] = g.Wp + ((long)min(nt + nb, NTILES - 1) * KT + (k0 >> 5)) * 512 + lane * 8;
    auto issue = [&](int t) {  // K-tile t -> ring slot t % D (past the slice: K-tile T - 1 again, into a slot nobody reads any more)
        const int ts = t < T ? t : T - 1;
        unsigned char* aslot = smem + A_RING + (t % D) * ASLOT;
#pragma unroll
        for (int a = 0; a < AU; ++a) __builtin_amdgcn_global_load_lds((const void*)(asrc[a] + (long)ts * RING_K), (rlds_ptr)(aslot + adst[a]), 16, 0, 0);
        unsigned char* wslot = smem + W_RING + (wave * D + t % D) * WSLOT;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            __builtin_amdgcn_global_load_lds((const void*)(wsrc[nb] + (long)ts * 1024), (rlds_ptr)(wslot + (nb * 2) * 1024), 16, 0, 2 /* nt: read once */);
            __builtin_amdgcn_global_load_lds((const void*)(wsrc[nb] + (long)ts * 1024 + 512), (rlds_ptr)(wslot + (nb * 2 + 1) * 1024), 16, 0, 2);
        }
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);

    // ---- NORM: 1/rms per row from the producer's sums of squares (fixed order), the norm weight of this K slice into LDS ----
    float* rsS = reinterpret_cast<float*>(smem + TAIL);                 // [64]
    bf16_t* nwS = reinterpret_cast<bf16_t*>(smem + TAIL + 256);          // [T * 64]
    if constexpr (NORM) {
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
            const float s2 = (q & 1) ? o1 + pq : pq + o1;   // (even + odd), the same operand order in both lanes: gemm_mid.hip's tree
            const float o2 = __shfl_xor(s2, 2, WAVE);
            const float tot = (q & 2) ? o2 + s2 : s2 + o2;
            if (row < g.M && q == 0) rsS[row] = rsqrtf(tot / g.K + g.norm_eps);
            if (row >= g.M && q == 0) rsS[row] = 0.f;
        }
        for (int i = tid; i < T * 8; i += NP * 64) *reinterpret_cast<u32x4_t*>(nwS + i * 8) = *reinterpret_cast<const u32x4_t*>(g.norm_w + k0 + i * 8);
    }
    // raw K-tile t (ring) -> normalised image F[t & 1]: wave w rewrites units w, w + NP, ...
    auto normalise = [&](int t) {
        const unsigned char* raw = smem + A_RING + (t % D) * ASLOT;
        unsigned char* f = smem + F_BUF + (t & 1) * 8192;
#pragma unroll
        for (int a = 0; a < AU; ++a) {
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

    f32x4_t acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { acc[mt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
    const int a_rd = (fr >> 3) * 1024 + (fr & 7) * 128, a_sw = (fr >> 1) & 7;

    if constexpr (NORM) {
        ring_wait<(D - 2) * PER>();            // K-tile 0 has landed (this wave's parts)
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): this wave's rsS / nwS stores
        __builtin_amdgcn_s_barrier();          // (raw: a __syncthreads() would drain the ring)
        normalise(0);
    }
    for (int t = 0; t < T; ++t) {
        // K-tile t (NORM: t + 1 as well) landed for this wave; the barrier makes every wave's parts visible and retires all reads of K-tile t - 1
        ring_wait<(NORM ? D - 3 : D - 2) * PER>();
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): (NORM) this wave's image writes of K-tile t
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        issue(t + D - 1);                     // into the slot K-tile t - 1 occupied
        const unsigned char* abase = (NORM ? smem + F_BUF + (t & 1) * 8192 : smem + A_RING + (t % D) * ASLOT) + a_rd;
        const unsigned char* wbase = smem + W_RING + (wave * D + t % D) * WSLOT + lane * 16;
        u32x4_t fa[4][2], fw[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 2; ++k) fw[nb][k] = *reinterpret_cast<const u32x4_t*>(wbase + (nb * 2 + k) * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) fa[mt][k] = *reinterpret_cast<const u32x4_t*>(abase + mt * 2048 + (((k * 4 + fq) ^ a_sw) << 4));
        if constexpr (NORM) {
            if (t + 1 < T) normalise(t + 1);  // (its raw rows landed with the wait above; F[(t + 1) & 1] was last read for K-tile t - 1)
        }
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                    acc[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[mt][k]), __builtin_bit_cast(bf16x8_t, fw[nb][k]), acc[mt][nb], 0, 0, 0);
    }
    ring_wait<0>();   // the over-issued DMAs must have landed before the rings are reused below
    __syncthreads();

    // ---- hand the accumulators to the shared epilogue: red[0][(mt * WN + n) * 4 + r][lane] ----
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((mt * WN + wave * 2 + nb) * 4 + r) * 64 + lane] = acc[mt][nb][r];
    __syncthreads();
    mid_epilogue<4, WN, 1, NP, EPI>(g, red, tid, 0, reinterpret_cast<int*>(smem + TILES * 1024));
}

static int g_ring_mode = 1;  // tuning hook (gemm_ring_set): 0 = never (gemm_mid.hip runs), 1 = default
void gemm_ring_set(int mode) { g_ring_mode = mode; }

template <int NP, bool NORM>
static size_t ring_lds(int T) {
    const int D = NORM ? 5 : 4;
    size_t b = (size_t)D * (9 * 1024) + (size_t)NP * D * 4096 + (NORM ? 2 * 8192 : 0);
    if (NORM) b += 256 + (size_t)T * RING_K * 2;
    const size_t epi = (size_t)4 * 2 * NP * 1024 + 16;
    return b > epi ? b : epi;
}

bool gemm_ring_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    if (g_ring_mode == 0 || g.batch != 1 || g.M <= 32 || g.M > 64 || g.attn_partial) return false;
    if (g.K % (RING_K * ks) != 0 || g.K / (RING_K * ks) < 4 || g.N % 16 != 0 || g.lda % 8 != 0) return false;
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
    dim3 grid((NTILES + 2 * NP - 1) / (2 * NP), ks), block(NP * 64);
    const size_t lds = ring_lds<NP, NORM>(T);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
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
// columns per workgroup: the widest that still gives (nearly) every CU a workgroup -- 64 columns of o_proj / down_proj x 4 K slices and 96 columns of
// q/k/v x 4 slices are 256 workgroups, gate/up's 128 columns 224, lm_head's 128 columns 1003
static int g_ring_np = 0;  // profiling aid: force NP
void gemm_ring_set_np(int np) { g_ring_np = np; }
template <int EPI>
static int launch_ring1(const GemmArgs& g, hipStream_t stream) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    const long pairs = g.N / 32;
    int np = 2;
    if (pairs / 4 * ks >= 200 && EPI != EPI_PARTIAL) np = 4;
    else if (pairs % 3 == 0 && pairs / 3 * ks >= 200) np = 3;
    else if (pairs / 4 * ks >= 200) np = 4;
    if (EPI == EPI_SWIGLU && np == 3) np = 4;  // ((gate, up) pairs are one n-tile pair each: any NP works; 4 measured)
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
