// Weight-streaming MFMA GEMM for 65..256 rows on gfx950:  out[M,N] = epi(A[M,K] @ W[N,K]^T), every weight byte read ONCE.
//
// Where it runs: the decode passes of 65..256 batched rows -- 65..256 concurrent streams, or 17..64 streams x beam 4, the reference's
// production decoding (agents/infinisst.py:86 asserts beam > 1; scripts/infer/infinisst.sh:48) -- through q/k/v, o_proj, gate/up, down_proj
// (patch_llm.py:260-262,334 and HF LlamaMLP [3P]) and lm_head (model/llm.py:236-262).  These launches are still HBM-bound (intensity
// 128..256 flop per weight byte against a ridge of ~300), but until round 4 they fell onto gemm_tiled.hip, whose one-K-step prefetch keeps
// ~32 KB in flight per CU: 3.3 TB/s on gate/up at 128 rows (profiles/r03/trace_busy_prof128.txt), and at 129..256 rows it read the weights twice.
//
// Structure -- gemm_tiled's dataflow (bit-identical results: same MFMA, same ascending K order per accumulator, same epilogue arithmetic)
// rebuilt around what a weight stream needs on this part, ~100 KB in flight per CU (MI355X_MICROARCH.md: ~25 GB/s per CU x 3-4 us loaded latency):
//   * workgroup = 8 waves (512 threads, __launch_bounds__(512, 2): at most 256 registers per wave), one per CU in practice, in two ROLES -- the two kinds
//     of load must not share a wave's in-order vmcnt queue (a wait for an L2-resident activation line would also wait for every weight fragment
//     issued before it: the first form of this kernel, 4 waves with both rings in registers, ran 56 us on gate/up at 128 rows for that reason);
//   * CONSUMER waves 0-3, one per SIMD: wave w owns n-tiles 2w, 2w+1 of the workgroup's 8 (128 columns; the (gate, up) pair of SwiGLU sits in one
//     wave) x ALL rows (MT m-tiles: 64 / 128 / 256 rows = 32 / 64 / 128 accumulator registers) x all of K: no K split across waves, no cross-wave
//     reduction, one accumulation chain per output element in ascending K.  W ring: the wave's weight fragments stream straight from the
//     fragment-major packed layout into a statically indexed register ring (1 KiB coalesced per load, non-temporal), DW K-steps (64 deep each: 4 KiB
//     per wave and step) ahead.  A fragments come from LDS in groups of 8 through two register sets, the next group requested under the MFMAs of
//     the current one; every MFMA (inline asm, accumulator tied) is followed by at most one filler instruction, pinned pair by pair;
//   * LOADER waves 4-7: the activation ring, NS stages of one K-step each ([m-tiles][16 rows][128 B], 16-byte chunks XOR-swizzled on the SOURCE
//     side), filled by LDS-DMA (global_load_lds_dwordx4: full 128-byte lines, no registers, no ds_write) with a hand-counted vmcnt in front of the
//     one barrier per K-step that publishes step t+2 while steps t+3 .. t+NS-1 stay in flight.  NORM: the loaders also RMS-normalise every staged
//     step in place (1/rms from the producer's sums of squares, GemmArgs::ssq);
//   * every load is unconditional: weight steps past the slice read zeros through a descriptor that covers exactly the slice, activation steps
//     past it re-read the last step into a stage nobody reads -- the loop has no branch around a load, every wait is counted, and both roles pass
//     the same number of barriers;
//   * epilogue through the idle ring: every value is written once into a [rows][workgroup columns] LDS image and stored as whole row segments;
//   * narrow outputs (q/k/v, o_proj, down_proj: 32..48 column blocks) split K over blockIdx.y into fp32 slabs (EPI_PARTIAL) that the
//     residual + RMSNorm kernel (or launch_slab_reduce) sums in slice order, exactly as gemm_tiled's split-K does; a K-sliced launch that would
//     leave half the CUs idle runs as two 128-row workgroups per (column block, slice) (blockIdx.z).
#include "common.h"

#define WIDE_TK 64          // K elements per step (two MFMA k-steps); one barrier per step
#define WIDE_NT 8           // n-tiles per workgroup (128 columns)

constexpr int wide_gcd(int a, int b) { return b == 0 ? a : wide_gcd(b, a % b); }
constexpr int wide_lcm(int a, int b) { return a / wide_gcd(a, b) * b; }

// acc += A x B (v_mfma_f32_16x16x32_bf16) with the accumulator TIED to one AGPR quad.  Through the builtin hipcc renames the 16..32 accumulators
// inside the unrolled ring period (dst != srcC) and repairs the names at the back edge with ~50 v_accvgpr_read / _write / _mov per K-step -- 40 %
// on top of the step's 32 MFMAs, with one wave per SIMD and nothing to hide them behind.  What the compiler no longer does for this statement
// (cdna_hip_programming.md section 5.7): it pads no hazard around it.  Inside the k-loop the only consumer of an accumulator is the next MFMA
// taking it whole as C (no wait states needed); the operands come from ds_read / buffer_load, which the compiler still waits for (they are
// ordinary register operands of the statement); the first non-MFMA reader is the epilogue, behind the s_nop pair after the loop.
__device__ __forceinline__ void wide_mfma(f32x4_t& c, const u32x4_t& a, const u32x4_t& b) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));  // ("+v": with an AGPR operand hipcc splits the 256 registers of a two-waves-per-SIMD kernel 128 / 128)
}

#ifdef ISST_WIDE_TRACE
// diagnostic build (make trace -> libinfinisst_hip_trace.so, profiles/wide_trace_probe.py): wave 0 of every workgroup stamps the shader clock
// (s_memtime; slots 0 and 3: the 100 MHz wall clock) into a device array the probe reads back: loop entry / exit, the end of every K-step, and the
// phases of step 8.  In the product build no stamp executes.
#define WIDE_TRACE_SLOTS 256
__device__ unsigned long long g_wide_trace[2048 * WIDE_TRACE_SLOTS];
extern "C" int isst_debug_wide_trace_read(void* dst, long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wide_trace), bytes < (long)sizeof(g_wide_trace) ? bytes : (long)sizeof(g_wide_trace)) == hipSuccess ? 0 : -2;
}
#define WIDE_SLOT(i) g_wide_trace[(long)((blockIdx.x + gridDim.x * blockIdx.y) & 2047) * WIDE_TRACE_SLOTS + (i)]
#define WIDE_STAMP(i) do { if (wave == 0) { const unsigned long long tt = __builtin_amdgcn_s_memtime(); if (lane == 0) WIDE_SLOT(i) = tt; } } while (0)
#define WIDE_STAMP_RT(i) do { if (wave == 0) { const unsigned long long tt = __builtin_amdgcn_s_memrealtime(); if (lane == 0) WIDE_SLOT(i) = tt; } } while (0)
#else
#define WIDE_STAMP(i) do { } while (0)
#define WIDE_STAMP_RT(i) do { } while (0)
#endif

typedef __attribute__((address_space(3))) void* wlds_ptr;
// s_waitcnt vmcnt(N) alone (expcnt / lgkmcnt left at their maxima): simm16 = vmcnt[3:0] | expcnt 7 << 4 | lgkmcnt 15 << 8 | vmcnt[5:4] << 14
template <int N>
__device__ __forceinline__ void wide_wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xf00 | ((N >> 4) << 14));
}

// NORM: A = x is RMS-normalised on its way to the MFMAs with 1/rms from the producer's sums of squares (GemmArgs::ssq: the consumer half of the launch-free
// residual + RMSNorm of gemm_mid.hip) -- by the LOADER waves, in LDS, one step ahead of its publication (an LDS-DMA has no register stage to do it in).
template <int MT, int DW, int NS, int EPI, bool NORM = false>
__global__ __launch_bounds__(512, 2) void gemm_wide_kernel(GemmArgs g, int steps, int dbg) {
    constexpr int AU = MT / 2;            // LDS-DMA units (8 rows x 128 B = 1 KiB) per loader wave and step
    constexpr int ABUF = MT * 2048;       // bytes of one staged step: [MT m-tiles][16 rows][128 B], 16-byte chunks XOR-swizzled
    static_assert(NS >= 4 && NS * ABUF <= 160 * 1024, "ring stages");
    constexpr int U = DW < 4 ? 4 : DW;    // unroll period of the consumers' loop (a multiple of DW that divides the usual step counts: every trip round the back edge costs ~350 cycles)
    static_assert(U % DW == 0, "ring slots are compile-time constants");
    constexpr int RING = NS * ABUF;       // NORM: behind the ring 1/rms of the tile's rows (MT*16 fp32), then the norm weight (K bf16)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NS x ABUF (all LDS of the kernel: cdna_hip_programming.md section 5, trap 4a)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: roles, descriptors and DMA destinations are built from it
    const int fr = lane & 15, fq = lane >> 4;
    const int KT = g.K >> 5, NTILES = g.N >> 4;
    const int slice = blockIdx.y;
    const int step0 = slice * steps;
    const int m0 = blockIdx.z * (MT * 16);

    if (wave >= 4) {
        // ================= loader waves: the A ring, by LDS-DMA =================
        // unit un = lw*AU + a = rows un*8 .. un*8+7 of the step, 128 B each; lane -> (row = lane >> 3, destination chunk = lane & 7); the chunk a lane
        // FETCHES is its destination chunk XOR ((row in m-tile >> 1) & 7): the DMA writes lane-linear, so the swizzle that makes the consumers'
        // ds_read_b128 conflict-free sits on the source side (gemm_dense.hip's image; rule 21).  Rows past M repeat row M-1 (never stored).
        const int lw = wave - 4;
#ifndef WIDE_LOADER_PRIO
#define WIDE_LOADER_PRIO 0   // s_setprio of the loader waves / of the consumer waves (A/B aids; round 6: see below)
#endif
#ifndef WIDE_CONSUMER_PRIO
#define WIDE_CONSUMER_PRIO 0
#endif
        if (WIDE_LOADER_PRIO) __builtin_amdgcn_s_setprio(WIDE_LOADER_PRIO);
        const bf16_t* asrc[AU];
#pragma unroll
        for (int a = 0; a < AU; ++a) {
            const int un = lw * AU + a;
            const int rim = (un & 1) * 8 + (lane >> 3);
            const int row = min(m0 + un * 8 + (lane >> 3), g.M - 1);
            asrc[a] = g.A + (long)row * g.lda + (long)step0 * WIDE_TK + (((lane & 7) ^ ((rim >> 1) & 7)) << 3);
        }
        auto dma = [&](int s /* step */, int stage) {  // steps past the slice re-read its last step into a stage nobody reads (issue and wait counts stay uniform)
            const long ko = (long)(s < steps ? s : steps - 1) * WIDE_TK;
            unsigned char* dst = smem + stage * ABUF + lw * AU * 1024;
            if (!(dbg & 2)) {
#pragma unroll
                for (int a = 0; a < AU; ++a) __builtin_amdgcn_global_load_lds((const void*)(asrc[a] + ko), (wlds_ptr)(dst + a * 1024), 16, 0, 0);
            }
        };
        // before the first barrier: steps 0 .. NS-2 issued, steps 0 and 1 landed
#pragma unroll
        for (int s0 = 0; s0 < NS - 1; ++s0) dma(s0, s0);
        float* rsS = reinterpret_cast<float*>(smem + RING);
        unsigned char* wS = smem + RING + MT * 16 * 4;
        // NORM: bf16(w * bf16(x * 1/rms)) over this wave's own units of the staged step `s` ([3P] HF LlamaRMSNorm's rounding points, as gemm_mid's consumer):
        // a lane re-reads the 16 bytes its DMA lane wrote (row lane >> 3 of the unit, the chunk that came from source chunk (lane & 7) ^ swizzle)
        auto normalise = [&](int s, int stage) {
            const int sc = s < steps ? s : steps - 1;
#pragma unroll
            for (int a = 0; a < AU; ++a) {
                const int un = lw * AU + a;
                const int rim = (un & 1) * 8 + (lane >> 3);
                unsigned char* px = smem + stage * ABUF + un * 1024 + lane * 16;
                const u32x4_t xv = *reinterpret_cast<const u32x4_t*>(px);
                const u32x4_t nwv = *reinterpret_cast<const u32x4_t*>(wS + (((long)step0 + sc) * WIDE_TK + (((lane & 7) ^ ((rim >> 1) & 7)) << 3)) * 2);
                const float rs = rsS[un * 8 + (lane >> 3)];
                u32x4_t o;
                o.x = pack_bf(lo_bf(nwv.x) * bfr(lo_bf(xv.x) * rs), hi_bf(nwv.x) * bfr(hi_bf(xv.x) * rs));
                o.y = pack_bf(lo_bf(nwv.y) * bfr(lo_bf(xv.y) * rs), hi_bf(nwv.y) * bfr(hi_bf(xv.y) * rs));
                o.z = pack_bf(lo_bf(nwv.z) * bfr(lo_bf(xv.z) * rs), hi_bf(nwv.z) * bfr(hi_bf(xv.z) * rs));
                o.w = pack_bf(lo_bf(nwv.w) * bfr(lo_bf(xv.w) * rs), hi_bf(nwv.w) * bfr(hi_bf(xv.w) * rs));
                *reinterpret_cast<u32x4_t*>(px) = o;
            }
        };
        if constexpr (NORM) {
            // 1/rms of every row of the tile from the producer's per-32-column sums of squares: 4 lanes per row, the SAME summation order as gemm_mid's
            // consumer (16 floats in flight per lane, then the fixed tree) -- a row gives the same 1/rms through either kernel.  These are ordinary loads:
            // hipcc drains the DMAs above in front of their first use, which costs nothing here (steps 0 and 1 have to have landed anyway)
            const int lt = tid - 256;
            for (int row = lt >> 2; row < MT * 16; row += 64) {
                const int q = lt & 3, grow = min(m0 + row, g.M - 1);
                const int per = g.ssq_n >> 2;
                const float* sp = g.ssq + (long)grow * g.ssq_n + q * per;
                float pq = 0.f;
                if ((per & 15) == 0) {
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
                const float o1 = __shfl_xor(pq, 1, WAVE);
                const float s2 = (q & 1) ? o1 + pq : pq + o1;   // (even + odd), the same operand order in both lanes
                const float o2 = __shfl_xor(s2, 2, WAVE);
                const float tot = (q & 2) ? o2 + s2 : s2 + o2;
                if (q == 0) rsS[row] = rsqrtf(tot / g.K + g.norm_eps);
            }
            for (int i = lt; i < (g.K >> 3); i += 256) *reinterpret_cast<u32x4_t*>(wS + i * 16) = *reinterpret_cast<const u32x4_t*>(g.norm_w + i * 8);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // 1/rms and the norm weight are in LDS for all four loader waves (the consumers pass this barrier too)
            normalise(0, 0);
            normalise(1, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            wide_wait_vmcnt<(NS - 3) * AU>();
        }
        __builtin_amdgcn_s_barrier();
        int stage = NS - 1;  // stage of step t - 1 + NS
        int stage2 = 2 % NS; // stage of step t + 2
        const int T = (steps + U - 1) / U * U;  // the consumers' loop is padded to its unroll period: the same number of barriers on both sides
        for (int t = 0; t < T; ++t) {
            // stage (t-1) % NS was read for the last time in step t-1 (every consumer passed barrier t-1 behind its reads): refill it with step t-1+NS;
            // then all but the NS-3 youngest steps have landed, i.e. step t+2 is complete before barrier t -- the consumers read it from step t+1 on
            dma(t - 1 + NS, stage);
            stage = stage + 1 == NS ? 0 : stage + 1;
            wide_wait_vmcnt<(NS - 3) * AU>();
            if constexpr (NORM) {
                normalise(t + 2, stage2);
                stage2 = stage2 + 1 == NS ? 0 : stage2 + 1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rewritten rows are in LDS before the barrier publishes them
            }
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued DMAs have landed ...
        __builtin_amdgcn_s_barrier();                      // ... before the consumers reuse the ring for their epilogue
        return;
    }

    // ================= consumer waves: the W ring in registers, fragments from LDS, MFMA =================
    WIDE_STAMP_RT(240); WIDE_STAMP(241);
    if (WIDE_CONSUMER_PRIO) __builtin_amdgcn_s_setprio(WIDE_CONSUMER_PRIO);
    const int nt0 = blockIdx.x * WIDE_NT + wave * 2;
    f32x4_t acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acc[mt][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[mt][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }

    // W: one descriptor per n-tile over exactly this slice's k-tiles (tiles past N: empty; steps past the slice read zeros without traffic)
    const bool nv0 = nt0 < NTILES, nv1 = nt0 + 1 < NTILES;
    __amdgpu_buffer_rsrc_t wrs[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const bool v = nb ? nv1 : nv0;
        wrs[nb] = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(g.Wp) + ((long)(v ? nt0 + nb : 0) * KT + (long)step0 * 2) * 512, 0, (v && !(dbg & 1)) ? steps * 2048 : 0, 0x00020000);  // (dbg: timing-only runs with a stream emptied, profiles/gemm_wide_probe.py)
    }
    const int woff = lane * 16;
    auto load_w = [&](int t, int kk, int nb) -> u32x4_t { return __builtin_amdgcn_raw_buffer_load_b128(wrs[nb], woff + (t * 2 + kk) * 1024, 0, 2 /* nt: read once */); };
    u32x4_t wreg[DW][2][2];   // W steps t .. t+DW-1 (slot = step % DW): [k-step of the step][n-tile]
#pragma unroll
    for (int s = 0; s < DW; ++s) {  // program order = order of use, pinned (left alone hipcc fills a ring back to front and the first wait drains it)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { wreg[s][kk][0] = load_w(s, kk, 0); wreg[s][kk][1] = load_w(s, kk, 1); }
        __builtin_amdgcn_sched_barrier(0);
    }
    // fragment (m-tile mt, k-step ks) of a stage: lane (fr, fq) reads 16 B at mt*2048 + fr*128 + (((ks*4 + fq) ^ ((fr >> 1) & 7)) << 4)
    const int rd0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4);
    const int rd1 = fr * 128 + (((4 + fq) ^ ((fr >> 1) & 7)) << 4);
    // A fragments travel in GROUPS of 8 (one k-step x 8 m-tiles = 16 MFMAs = 256 matrix-pipe cycles) through two register sets: the next group -- of
    // this step, or the FIRST group of the next step, whose image the previous barrier already published -- is requested under the MFMAs of the
    // current one.
    constexpr int GS = MT >= 8 ? 8 : MT;       // fragments per group (MT = 4: 64 rows, 8 MFMAs per group)
    constexpr int MH = MT / GS, NG = 2 * MH;
    static_assert(GS >= 4 && MT % GS == 0, "a group carries its GS fragment reads and the 4 ring loads as fillers");
    u32x4_t af[2][GS];
    if constexpr (NORM) __builtin_amdgcn_s_barrier();  // (the loaders' 1/rms + norm-weight staging barrier)
    __builtin_amdgcn_s_barrier();  // steps 0 and 1 are in LDS
#pragma unroll
    for (int j = 0; j < GS; ++j) af[0][j] = *reinterpret_cast<const u32x4_t*>(smem + rd0 + j * 2048);
    WIDE_STAMP_RT(0); WIDE_STAMP(1);
    int st0 = 0, st1 = ABUF;  // LDS offsets of the stages of steps t and t+1
    for (int t0 = 0; t0 < steps; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u;
            // NO condition on t: an iteration past the last step (steps % DW != 0) multiplies the slice's last rows (the loaders re-read them) by zero
            // weights (past the descriptor) -- every path through the loop issues the same loads in the same order, which is what lets hipcc count
            // its waits (a skipped iteration makes the back edge's wait a drain), and both roles pass the same number of barriers
            const int sw = u % DW;
            if (t == 8) WIDE_STAMP(205);
            // ONE MFMA instruction stream per SIMD: whatever is issued between two MFMAs runs in the shadow of the first (an MFMA holds the issue port for
            // 8 of its 16 cycles), whatever is issued in a block of its own stalls the matrix pipe (trace build: 8 ds_read_b128 128 cycles, 4 buffer
            // loads 150).  So every MFMA is followed by at most one filler: the next group's 8 fragment reads, then the ring refill.  Pinned pair by pair.
#pragma unroll
            for (int gi = 0; gi < NG; ++gi) {
                const int kk = gi / MH, mh = gi % MH;
                const int ngi = (gi + 1) % NG;
                const unsigned char* nrd = smem + ((gi + 1 < NG) ? st0 : st1) + ((ngi / MH) ? rd1 : rd0) + (ngi % MH) * GS * 2048;
#pragma unroll
                for (int k = 0; k < 2 * GS; ++k) {
                    wide_mfma(acc[mh * GS + (k >> 1)][k & 1], af[gi & 1][k >> 1], wreg[sw][kk][k & 1]);
                    if (k < GS) af[(gi + 1) & 1][k] = *reinterpret_cast<const u32x4_t*>(nrd + k * 2048);
                    else if (gi == NG - 1 && k >= 2 * GS - 4) wreg[sw][(k - (2 * GS - 4)) >> 1][(k - (2 * GS - 4)) & 1] = load_w(t + DW, (k - (2 * GS - 4)) >> 1, (k - (2 * GS - 4)) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (t == 8) WIDE_STAMP(210 + gi);
            }
            if (t == 8) WIDE_STAMP(206);
            // the stage of step t is free once every consumer's reads of it have returned: all LDS reads but the GS youngest (the first group of step
            // t+1) -- LDS operations of a wave complete in order
            if constexpr (GS == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t < 200) WIDE_STAMP(4 + t);
            st0 = st1;
            st1 = st1 + ABUF == NS * ABUF ? 0 : st1 + ABUF;
        }
    }
    WIDE_STAMP(2); WIDE_STAMP_RT(3);
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // the last MFMAs have left the pipe before the epilogue reads the accumulators (wide_mfma)
    __builtin_amdgcn_s_barrier();  // every DMA has landed, every fragment read has returned: the ring is free (the loaders leave here)
    // ---- epilogue through LDS: every value is written once into a [rows][workgroup columns] image of the idle ring -- after the bias / GELU / SwiGLU
    //      arithmetic, i.e. at the reference's rounding point in front of the residual add -- and read back as 16 bytes per lane: a store instruction then
    //      covers whole 128 / 256 / 512-byte row segments instead of 4 rows x 32 (bf16) or 64 (fp32) bytes of the accumulator layout, and the residual is
    //      read the same way (the straight-from-accumulator form below took 9200 cycles = 4.4 us of a 46 us gate/up launch: profiles/r04/wide_trace_v3*).
    //      The 16-byte chunk index of a row is XORed with its fq (rows 4 apart alias on the banks).  Same arithmetic, same bits.
    {
        constexpr bool F32_OUT = EPI == EPI_PARTIAL || EPI == EPI_F32;
        constexpr int WC = EPI == EPI_SWIGLU ? 64 : 128;     // output columns of the workgroup
        constexpr int ES = F32_OUT ? 4 : 2;
        constexpr int RS = WC * ES, CPR = RS / 16, EPC = 16 / ES;  // image row stride (bytes), chunks per row, elements per chunk
        constexpr int SWS = F32_OUT ? 2 : 1;                  // chunk XOR = fq << SWS: the 2 (bf16) / 4 (fp32) chunks a wave-store spans per row, times 4 row groups
        static_assert(MT * 16 * RS <= NS * ABUF, "the image fits the ring");
        unsigned char* outp = reinterpret_cast<unsigned char*>(g.out) + (EPI == EPI_PARTIAL ? (long)slice * g.out_batch * 4 : 0);
        const bool fast = (g.n_valid % EPC) == 0 && (g.ldo % EPC) == 0 && (reinterpret_cast<uintptr_t>(outp) & 15) == 0 &&
                          ((EPI != EPI_RES && EPI != EPI_BIAS_RES) || ((g.ldres & 7) == 0 && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0));
        if (fast) {
            const int wcol = (EPI == EPI_SWIGLU ? wave * 16 : wave * 32) + fr;  // column inside the workgroup's image (+ nb * 16)
            float bv[2] = {0.f, 0.f};
            if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = (nt0 + nb) * 16 + fr;
                    bv[nb] = col < g.n_valid ? bf2f(g.bias[col]) : 0.f;
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned char* rowp = smem + (mt * 16 + fq * 4 + r) * RS;
                    if constexpr (EPI == EPI_SWIGLU) {
                        const float v = bfr(silu(bfr(acc[mt][0][r]))) * bfr(acc[mt][1][r]);
                        *reinterpret_cast<bf16_t*>(rowp + (((wcol >> 3) ^ (fq << SWS)) << 4) + (wcol & 7) * 2) = f2bf(v);
                    } else {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const int c = wcol + nb * 16;
                            const float sacc = acc[mt][nb][r];
                            if constexpr (EPI == EPI_PARTIAL) {
                                *reinterpret_cast<float*>(rowp + (((c >> 2) ^ (fq << SWS)) << 4) + (c & 3) * 4) = sacc;
                            } else if constexpr (EPI == EPI_F32) {
                                *reinterpret_cast<float*>(rowp + (((c >> 2) ^ (fq << SWS)) << 4) + (c & 3) * 4) = bfr(sacc);
                            } else {
                                float v;
                                if constexpr (EPI == EPI_NONE || EPI == EPI_RES) v = sacc;
                                else if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES) v = sacc + bv[nb];
                                else v = gelu_erf(bfr(sacc + bv[nb]));
                                *reinterpret_cast<bf16_t*>(rowp + (((c >> 3) ^ (fq << SWS)) << 4) + (c & 7) * 2) = f2bf(v);
                            }
                        }
                    }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's image writes
            __builtin_amdgcn_s_barrier();        // (the four consumer waves: the loaders have exited)
            const int col_base = blockIdx.x * WC;
#pragma unroll 4
            for (int cid = tid; cid < MT * 16 * CPR; cid += 256) {
                const int lrow = cid / CPR, c = cid % CPR;
                const int row = m0 + lrow, gcol = col_base + c * EPC;
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(smem + lrow * RS + ((c ^ (((lrow >> 2) & 3) << SWS)) << 4));
                if (row < g.M && gcol < g.n_valid) {
                    if constexpr (EPI == EPI_RES || EPI == EPI_BIAS_RES) {
                        const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(g.res + (long)row * g.ldres + gcol);
                        v.x = pack_bf(lo_bf(rv.x) + lo_bf(v.x), hi_bf(rv.x) + hi_bf(v.x));
                        v.y = pack_bf(lo_bf(rv.y) + lo_bf(v.y), hi_bf(rv.y) + hi_bf(v.y));
                        v.z = pack_bf(lo_bf(rv.z) + lo_bf(v.z), hi_bf(rv.z) + hi_bf(v.z));
                        v.w = pack_bf(lo_bf(rv.w) + lo_bf(v.w), hi_bf(rv.w) + hi_bf(v.w));
                    }
                    *reinterpret_cast<u32x4_t*>(outp + ((long)row * g.ldo + gcol) * ES) = v;
                }
            }
            WIDE_STAMP(242); WIDE_STAMP_RT(243);
            return;
        }
    }
    // ---- general form (ragged n_valid, unaligned rows), straight from the accumulators (gemm_tiled.hip's arithmetic, rounding point for rounding point):
    //      acc[mt][nb][r] = C[m0 + mt*16 + 4 fq + r][(nt0 + nb)*16 + fr] ----
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + mt * 16 + fq * 4 + r;
            if (row >= g.M) continue;
            if constexpr (EPI == EPI_SWIGLU) {
                const int col = (nt0 >> 1) * 16 + fr;
                if (nv0 && col < g.n_valid)
                    reinterpret_cast<bf16_t*>(g.out)[(long)row * g.ldo + col] = f2bf(bfr(silu(bfr(acc[mt][0][r]))) * bfr(acc[mt][1][r]));
            } else {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int col = (nt0 + nb) * 16 + fr;
                    if (!(nb ? nv1 : nv0) || col >= g.n_valid) continue;
                    const float s = acc[mt][nb][r];
                    if constexpr (EPI == EPI_PARTIAL) {
                        reinterpret_cast<float*>(g.out)[(long)slice * g.out_batch + (long)row * g.ldo + col] = s;
                    } else if constexpr (EPI == EPI_F32) {
                        reinterpret_cast<float*>(g.out)[(long)row * g.ldo + col] = bfr(s);
                    } else {
                        float v;
                        if constexpr (EPI == EPI_NONE) v = s;
                        else if constexpr (EPI == EPI_BIAS) v = s + bf2f(g.bias[col]);
                        else if constexpr (EPI == EPI_BIAS_GELU) v = gelu_erf(bfr(s + bf2f(g.bias[col])));
                        else if constexpr (EPI == EPI_RES) v = bf2f(g.res[(long)row * g.ldres + col]) + bfr(s);
                        else v = bf2f(g.res[(long)row * g.ldres + col]) + bfr(s + bf2f(g.bias[col]));
                        reinterpret_cast<bf16_t*>(g.out)[(long)row * g.ldo + col] = f2bf(v);
                    }
                }
            }
        }
    }
    WIDE_STAMP(242); WIDE_STAMP_RT(243);
}

// tuning hooks (gemm_wide_set): mode 0 = never, 1 = heuristic, 2 = wherever supported; variant = ring depths (profiles/gemm_wide_probe.py)
static int g_wide_mode = 1, g_wide_variant = 0, g_wide_max_rows = 256, g_wide_min_rows = 65, g_wide_dbg = 0;
bool gemm_wide_enabled() { return g_wide_mode != 0; }
void gemm_wide_set(int mode, int variant) { g_wide_dbg = (mode / 10) % 10; g_wide_mode = mode % 10; g_wide_variant = variant; g_wide_min_rows = mode >= 100 ? 17 : 65; }  // (mode + 100: also 17..64 rows, probes)

// rows: 65..256 for every long weight stream; from 49 rows for the WIDEST ones when they normalise on stage (gate/up and lm_head of a 49..64-stream decode
// pass: 41.0 against gemm_mid's 45.1 us at 64 rows, profiles/r04/gemm_wide_rows64_probe.txt -- the narrow projections tie gemm_mid there and stay on it)
static bool wide_rows_ok(const GemmArgs& g) {
    if (g.M >= g_wide_min_rows && g.M <= g_wide_max_rows) return true;
    return g.norm_w && g.ssq && g.M >= 49 && g.M <= 64 && (long)g.N * g.K >= (64L << 20);
}
bool gemm_wide_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    if (g.epi == EPI_PARTIAL ? g.ksplit < 1 : g.ksplit > 1) return false;
    if (g.norm_w && !(g.ssq && g.ssq_n * 32 == g.K && g.ssq_n % 4 == 0 && ks == 1 && g.K <= 8192 && (g.epi == EPI_NONE || g.epi == EPI_SWIGLU || g.epi == EPI_F32))) return false;
    return g_wide_mode != 0 && g.batch == 1 && wide_rows_ok(g) && g.K % (WIDE_TK * ks) == 0 && g.K / (WIDE_TK * ks) >= 2 && g.N % 16 == 0 && g.lda % 8 == 0 &&
           !g.attn_partial && !g.tickets && (g.epi != EPI_SWIGLU || g.N % 32 == 0) && ((long)g.M - 1) * g.lda + g.K < (1L << 29);
}
// worth it where the weight stream is long (as gemm_mid_preferred): the encoder's 2-8 MB projections at 96 rows are latency-bound
bool gemm_wide_preferred(const GemmArgs& g) { return g_wide_mode == 2 || (long)g.N * g.K >= (8L << 20); }

template <int MT, int DW, int NS, int EPI, bool NORM>
static int launch_wide_cfg2(const GemmArgs& g, hipStream_t stream) {
    const int ks = g.epi == EPI_PARTIAL ? (g.ksplit > 1 ? g.ksplit : 1) : 1;
    const int NTILES = g.N / 16;
    dim3 grid((NTILES + WIDE_NT - 1) / WIDE_NT, ks, (g.M + MT * 16 - 1) / (MT * 16)), block(512);
    const size_t lds = (size_t)NS * MT * 2048 + (NORM ? (size_t)MT * 16 * 4 + (size_t)g.K * 2 : 0);  // NORM: + 1/rms of the rows + the norm weight
    static size_t attr_lds = 0;  // (NORM: the size depends on K)
    if (lds >= 64 * 1024 && lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<MT, DW, NS, EPI, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ISST_ERR_HIP;
        attr_lds = lds;
    }
    hipLaunchKernelGGL((gemm_wide_kernel<MT, DW, NS, EPI, NORM>), grid, block, lds, stream, g, g.K / WIDE_TK / ks, g_wide_dbg);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
template <int MT, int DW, int NS, int EPI>
static int launch_wide_cfg(const GemmArgs& g, hipStream_t stream) {
    if constexpr (EPI == EPI_NONE || EPI == EPI_SWIGLU || EPI == EPI_F32) {
        if (g.norm_w) return launch_wide_cfg2<MT, DW, NS, EPI, true>(g, stream);
    }
    return launch_wide_cfg2<MT, DW, NS, EPI, false>(g, stream);
}

// ring depths (profiles/gemm_wide_probe.py, r04/wide_probe_v4*): W ring 4 K-steps / A ring 6 stages up to 128 rows (deeper rings measured equal or slower);
// 256 rows: the 128 accumulator registers of a consumer leave room for a W ring of 2 steps, and the 32 KB stages for 4 of them (a 4-step W ring with
// fragment groups of 4 instead of 8 fits too -- 234 VGPRs, no spill -- and measures the same: gate/up 63.6 / 65.1 us, profiles/r04/wide_probe_row_blocks.txt)
// (round 5: a FIFTH 32 KB stage -- all 160 KB of LDS, one more K-step of activations in flight -- measures the same too: gate/up 63.5 / 63.5 us, 64 x 4 beams
//  101.1 / 100.9 ms per step: profiles/r05/wide_256_rows_5_stage_ring_no_gain.txt)
template <int EPI>
static int launch_wide_epi(const GemmArgs& g, hipStream_t stream) {
    if (g.M <= 64) return launch_wide_cfg<4, 4, 6, EPI>(g, stream);
    if (g.M <= 128) return launch_wide_cfg<8, 4, 6, EPI>(g, stream);
    // 129..256 rows: one 256-row workgroup per (column block, K slice) -- or two 128-row ones where BOTH row blocks still fit the chip in one round
    // (o_proj: 32 column blocks x 4 slices x 2 = 256 workgroups).  A 256-row workgroup is bound by its own matrix pipes (one consumer wave per SIMD:
    // 16 x 8 MFMAs per k-step), and with 128 workgroups half the CUs idle; the twin row blocks read the same weights at the same time on the same XCD
    // (their linear ids differ by a multiple of 8).  GEMM + reducing norm, us, one 256-row / two 128-row workgroups: o_proj 27.5 / 23.7 (192 rows: 26.2 /
    // 22.8); q/k/v 30.1 / 31.6 and down_proj 46.0 / 46.1 at their best slice counts -- no gain, they keep the 256-row form (profiles/r04/wide_probe_row_blocks.txt).
    // (variant bit 0: two row blocks for every K-sliced launch, bit 1: for every launch, bit 2: never -- A/B runs)
    const int ks = g.epi == EPI_PARTIAL && g.ksplit > 1 ? g.ksplit : 1;
    const bool twin = (g_wide_variant & 2) || ((g_wide_variant & 1) && EPI == EPI_PARTIAL) ||
                      (!(g_wide_variant & 4) && EPI == EPI_PARTIAL && (long)((g.N / 16 + WIDE_NT - 1) / WIDE_NT) * ks * 2 <= 256);
    if (twin) return launch_wide_cfg<8, 4, 6, EPI>(g, stream);
    return launch_wide_cfg<16, 2, 4, EPI>(g, stream);
}

int launch_gemm_wide(const GemmArgs& g, hipStream_t stream) {
    if (!gemm_wide_supported(g)) return ISST_ERR_ARG;
    switch (g.epi) {
        case EPI_NONE: return launch_wide_epi<EPI_NONE>(g, stream);
        case EPI_SWIGLU: return launch_wide_epi<EPI_SWIGLU>(g, stream);
        case EPI_PARTIAL: return launch_wide_epi<EPI_PARTIAL>(g, stream);
        case EPI_F32: return launch_wide_epi<EPI_F32>(g, stream);
        case EPI_BIAS: return g.bias ? launch_wide_epi<EPI_BIAS>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_GELU: return g.bias ? launch_wide_epi<EPI_BIAS_GELU>(g, stream) : ISST_ERR_ARG;
        case EPI_RES: return g.res ? launch_wide_epi<EPI_RES>(g, stream) : ISST_ERR_ARG;
        case EPI_BIAS_RES: return (g.res && g.bias) ? launch_wide_epi<EPI_BIAS_RES>(g, stream) : ISST_ERR_ARG;
    }
    return ISST_ERR_ARG;
}
