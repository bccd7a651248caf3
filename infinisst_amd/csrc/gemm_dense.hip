// 256 x 256-tile MFMA GEMM for the many-row (dense) shapes of the InfiniSST path on gfx950:
//   out[M,N] = epi(A[M,K] @ W[N,K]^T): the Llama prefill of many streams (64 streams x 22 prompt rows = 1408 rows against q/k/v, o_proj,
//   gate/up, down_proj -- reference patch_llm.py:260-262,334 and HF LlamaMLP [3P]), the speech encoder's projections at many streams
//   (patch_speech_encoder.py:741-743,923,586-589) and the conv stack as implicit GEMM.  Below 65 rows the weight-streaming kernels run
//   (gemm.hip, gemm_mid.hip); gemm_tiled.hip (128 x 128) keeps the shapes this kernel does not fit (K not a multiple of 64, batched).
//
// Structure (cdna_hip_programming.md section 5, "the 256^2 8-phase template", re-derived for the fragment-major packed weights):
//   * 8 waves = 2 (M) x 4 (N); wave (wr, wc) owns 128 rows x 64 columns of the tile = 2 x 2 QUADRANTS of 64 x 32 (4 m-tiles x 2 n-tiles of
//     v_mfma_f32_16x16x32_bf16), 32 accumulators = 128 VGPRs;
//   * a K-TILE is 64 deep; both operands of a K-tile go through LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
//     pass), 2 K-tile buffers x 4 HALF-TILE slots of 16 KiB = 128 KiB, one workgroup per CU:
//       A0 / A1 = rows {wr*128 + mh*64 + 0..63} of both wave rows: 16 units of 8 rows x 128 B (full lines), the 16-byte chunk index XORed with
//                 ((row >> 1) & 7) on the SOURCE side (the DMA destination is lane-linear) and on the read side: conflict-free ds_read_b128;
//       B0 / B1 = n-tiles {wc*4 + nh*2 + 0..1} of all four wave columns x 2 k-steps: 16 fragments of 1 KiB exactly as the packed weights lie
//                 in memory ([n-tile][k-tile][lane] x 16 B): lane-linear reads, conflict-free by construction;
//   * a PHASE = one quadrant x one K-tile = 16 MFMAs (256 matrix-pipe cycles).  Quadrant walk (0,0) (0,1) (1,1) (1,0): phase 1 reads A0 + B0
//     (12 ds_read_b128), phase 2 B1 (4), phase 3 A1 (8), phase 4 nothing -- B0 stays in registers for its second use, so every half-tile slot
//     is read in exactly one phase and each of the 24 fragment reads of a K-tile happens once;
//   * every phase issues ONE half-tile of DMA (2 instructions per wave) into the slot whose last read lies two phases back:
//       phase 1 of K-tile t: B1 of t+1, phase 2: A1 of t+1, phase 3: A0 of t+2, phase 4: B0 of t+2  (needed 5-6 phases later),
//     then s_waitcnt vmcnt(8): all but the four youngest half-tiles of this wave have landed -- never a drain; the barrier that follows makes
//     them visible to the other waves, and a slot is read one phase or more after the wait that retired it;
//   * the two wave rows run the phases half a phase apart (wr = 1 passes one extra barrier before the loop): between two consecutive
//     barriers one wave of every SIMD issues its 16 MFMAs while its partner issues its reads and DMAs ("ping-pong"), so the matrix pipe of a
//     SIMD always has exactly one wave feeding it; s_setprio(1) around the MFMA cluster keeps hipcc from moving MFMAs across the barriers;
//   * K order per accumulator: ascending, two k-steps per K-tile -- the same order as gemm_tiled.hip, so results are bit-identical to it.
// Scheduling: one workgroup per tile in an XCD-aware order (as gemm_tiled.hip); narrow outputs split K over blockIdx.z into fp32 slabs
// (EPI_PARTIAL) that the residual + RMSNorm kernel sums.
//
// Round 6: HALF TILES (MH = 1).  1408 rows are 5.5 row blocks of 256: the sixth block computes on 128 repeated rows, and 6 x 112 = 672 gate/up tiles are 2.625
// rounds of the 256 CUs -- three in practice (profiles/r05/trace_busy_prof64.txt: 277 us = 3 x 92).  The same body therefore also exists for a 128-row x 256-column
// tile: each wave owns 64 rows x 64 columns = the two quadrants (0, 0) (0, 1), a K-tile is two phases, and its three half-tile slots A | B0 | B1 (48 KiB) form a
// ring of THREE K-tile buffers (144 KiB), so that a DMA still has three phases to land although a K-tile now lasts two: phase 1 of K-tile t issues A(t+2) and the
// first half of B0(t+2), phase 2 the rest of B0(t+2) and B1(t+2) (three instructions per wave and phase), waits vmcnt(9) / vmcnt(8).  Same K order per
// accumulator: bit-identical to the 256-row form and to gemm_tiled.hip.  The launcher mixes the two sizes in ONE launch -- row blocks of 256 first, then row blocks
// of 128, longest first inside every XCD's share of the columns -- wherever its list-scheduling model says the last round gets shorter (dense_pick_mix).
#include "common.h"

// diagnostic builds only (make ablate; profiles/dense_ablate_probe.py): DENSE_ABLATE bit 0 = no MFMAs (fragments kept live), bit 1 = no DMAs inside the K loop
// (the prologue's stay), bit 2 = no fragment reads inside the K loop -- what each of the three streams of the loop costs alone.  Results are garbage.
#ifndef DENSE_ABLATE
#define DENSE_ABLATE 0
#endif
#ifndef DENSE_RING10
#define DENSE_RING10 1   // 256-row tiles: one half-tile per phase over a ring of slots instead of two K-tile buffers (A/B aid: -DDENSE_RING10=0)
#endif
#ifndef DENSE_MFMA_PRIO
#define DENSE_MFMA_PRIO 1   // s_setprio inside the MFMA cluster of a phase / in the load segment (reads, DMA issue, wait) around it (A/B aids)
#endif
#ifndef DENSE_LOAD_PRIO
#define DENSE_LOAD_PRIO 0
#endif
#ifndef DENSE_SLOTS
#define DENSE_SLOTS 10   // half-tile slots of that ring (10 = all 160 KiB of the CU)
#endif

#define DT_M 256
#define DT_NT 16          // n-tiles per tile (256 columns)
#define DT_K 64
#define DT_HALF 16384     // bytes per half-tile slot
#define DT_BUF 65536      // bytes per K-tile buffer: A0 | A1 | B0 | B1

typedef __attribute__((address_space(3))) void* dlds_ptr;

// tile mix of a launch (dense_pick_mix): row blocks 0 .. n_full-1 are 256 rows, then n_half blocks of 128 rows; mixed launches are 1-D grids that also fold the K
// slices in (blockIdx.x -> XCD share, tile, slice), so that every XCD walks ITS tiles longest first
struct DenseSched {
    int n_full, n_half;
    int X;               // column blocks
    int ks;              // K slices folded into the grid (folded launches)
    int folded;          // 0: the plain grid of rounds 1-5 (256-row tiles, raster flag, blockIdx.y / .z); 1: the 1-D grid below
};

template <int EPI, int MH>
__device__ __forceinline__ void dense_tile(const GemmArgs& g, unsigned char* smem, const int m0, const int nt0, const int slice, const int T, const long k0) {
    constexpr int NBUF = MH == 2 ? 2 : 3;            // K-tile buffers
    constexpr int BUFSZ = (MH + 2) * DT_HALF;        // A0 (| A1) | B0 | B1
    constexpr int WROWS = 64 * MH;                   // rows of a wave row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int KT = g.K >> 5, NTILES = g.N >> 4;

    // ---- DMA sources of this wave.  A half mh: units 2*wave, 2*wave + 1 of the slot = m-tile (wave & 3) of wave row (wave >> 2);
    //      lane -> (row of the 8-row unit = lane >> 3, destination chunk = lane & 7), source chunk = destination chunk ^ ((row in m-tile >> 1) & 7) ----
    const bf16_t* asrc[MH][2];  // [mh][unit]
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int rim = u * 8 + (lane >> 3);                                   // row inside the 16-row m-tile
            const int row = min(m0 + (wave >> 2) * WROWS + mh * 64 + (wave & 3) * 16 + rim, g.M - 1);  // rows past M repeat the last row (never stored)
            const int chunk = (lane & 7) ^ ((rim >> 1) & 7);
            asrc[mh][u] = g.A + (long)row * g.lda + k0 + chunk * 8;
        }
    //      B half nh: fragments 2*wave, 2*wave + 1 of the slot = n-tile (wave >> 1) * 4 + nh * 2 + (wave & 1), k-steps 0 and 1
    const bf16_t* bsrc[2];     // [nh], k-step 0 of K-tile 0; k-step 1 is 512 elements further
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        const int nt = min(nt0 + (wave >> 1) * 4 + nh * 2 + (wave & 1), NTILES - 1);
        bsrc[nh] = g.Wp + ((long)nt * KT + (k0 >> 5)) * 512 + lane * 8;
    }
    // (past the last K-tile the DMAs re-read K-tile T - 1 into the slot the running index names -- a slot nobody reads any more: the issue
    //  and wait counts then stay the same in every phase)
    // one DMA instruction: unit u (0 / 1) of this wave's share of a half-tile of K-tile t, into the 16 KiB slot at byte offset `soff`
    auto dmaA = [&](int soff, int t, int mh, int u) {
        unsigned char* dst = smem + soff + (2 * wave + u) * 1024;
        const long ko = (long)(t < T ? t : T - 1) * DT_K;
        if ((DENSE_ABLATE & 2) && t >= 2) return;
        __builtin_amdgcn_global_load_lds((const void*)(asrc[mh][u] + ko), (dlds_ptr)dst, 16, 0, 0);
    };
    auto dmaB = [&](int soff, int t, int nh, int u) {
        unsigned char* dst = smem + soff + (2 * wave + u) * 1024;
        const long ko = (long)(t < T ? t : T - 1) * 1024;
        if ((DENSE_ABLATE & 2) && t >= 2) return;
        __builtin_amdgcn_global_load_lds((const void*)(bsrc[nh] + ko + u * 512), (dlds_ptr)dst, 16, 0, 0);
    };
    auto dma_a1 = [&](int boff, int t, int mh, int u) { dmaA(boff + mh * DT_HALF, t, mh, u); };          // buffer-relative forms (K-tile buffers A0 (| A1) | B0 | B1)
    auto dma_b1 = [&](int boff, int t, int nh, int u) { dmaB(boff + (MH + nh) * DT_HALF, t, nh, u); };
    auto dma_a = [&](int t, int mh) {  // (MH == 2) A half mh of K-tile t -> buffer t & 1
        dma_a1((t & 1) * BUFSZ, t, mh, 0);
        dma_a1((t & 1) * BUFSZ, t, mh, 1);
    };
    auto dma_b = [&](int t, int nh) {
        dma_b1((t & 1) * BUFSZ, t, nh, 0);
        dma_b1((t & 1) * BUFSZ, t, nh, 1);
    };

    // ---- fragment reads ----
    const int a_rd = wr * 8192 + (fr >> 3) * 1024 + (fr & 7) * 128;     // + mt * 2048 + (((ks * 4 + fq) ^ ((fr >> 1) & 7)) << 4)
    const int a_sw = (fr >> 1) & 7;
    const int b_rd = wc * 4096 + lane * 16;                              // + (nb * 2 + ks) * 1024
    u32x4_t fa[4][2], fb0[2][2], fb1[2][2];                              // [m-tile][k-step], [n-tile][k-step]
    bool first_reads = true;
    auto rdA = [&](int soff) {  // the A half-tile in the slot at byte offset soff -> fa
        if ((DENSE_ABLATE & 4) && !first_reads) return;
        const unsigned char* base = smem + soff + a_rd;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int k = 0; k < 2; ++k) fa[mt][k] = *reinterpret_cast<const u32x4_t*>(base + mt * 2048 + (((k * 4 + fq) ^ a_sw) << 4));
    };
    auto rdB = [&](int soff, u32x4_t (&fb)[2][2]) {
        if ((DENSE_ABLATE & 4) && !first_reads) return;
        const unsigned char* base = smem + soff + b_rd;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int k = 0; k < 2; ++k) fb[nb][k] = *reinterpret_cast<const u32x4_t*>(base + (nb * 2 + k) * 1024);
    };
    auto read_a = [&](int boff, int mh) { rdA(boff + mh * DT_HALF); };
    auto read_b = [&](int boff, int nh, u32x4_t (&fb)[2][2]) { rdB(boff + (MH + nh) * DT_HALF, fb); };

    f32x4_t acc[MH][2][4][2];  // [mh][nh][m-tile][n-tile]
#pragma unroll
    for (int a = 0; a < MH; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[a][b][mt][nb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    auto quadrant = [&](f32x4_t (&c)[4][2], const u32x4_t (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(DENSE_MFMA_PRIO);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    if constexpr (DENSE_ABLATE & 1) asm volatile("" ::"v"(fa[mt][k]), "v"(fb[nb][k]));
                    else c[mt][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa[mt][k]), __builtin_bit_cast(bf16x8_t, fb[nb][k]), c[mt][nb], 0, 0, 0);
                }
        __builtin_amdgcn_s_setprio(DENSE_LOAD_PRIO);
    };
#define DT_WAIT_BARRIER_N(N)                                \
    asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_sched_barrier(0)
#define DT_WAIT_BARRIER() DT_WAIT_BARRIER_N(8)
#define DT_END_PHASE()                                      \
    __builtin_amdgcn_sched_barrier(0);                      \
    __builtin_amdgcn_s_barrier();                           \
    __builtin_amdgcn_sched_barrier(0)

    if constexpr (MH == 2 && DENSE_RING10) {
        // ---- ONE HALF-TILE PER PHASE over a ring of DENSE_SLOTS half-tile slots (round 6; ten slots = all 160 KiB of the CU).  The two-buffer form below reads
        //      12 / 4 / 8 / 0 fragments in the phases of a K-tile: in phase 1 the four waves of a wave row put 4 x 12 x 4 = 192 LDS cycles of reads beside the
        //      phase's 64 cycles of DMA writes -- the whole 256-cycle MFMA cluster of their partners -- and the matrix pipe waits for them.  Here B0 of K-tile
        //      t + 1 is read in phase 4 of K-tile t, into the registers B1(t) has just left (the two B register sets swap roles every K-tile: the loop is
        //      unrolled by two), so the phases read 8 / 4 / 8 / 4 fragments, and every phase consumes exactly one half-tile of the sequence
        //          B0(0) | A0(t) B1(t) A1(t) B0(t+1) | ...        (sequence index j: phase p = 4 t + k, k = 0 .. 3, reads j = p + 1)
        //      and issues exactly one (j = p + 1 + LOOK, LOOK = DENSE_SLOTS - 2) into slot j mod DENSE_SLOTS, whose last occupant was read two phases back.
        //      A wait leaves the LOOK - 1 youngest half-tiles flying; the half-tile a phase reads was retired by the wait of the phase before. ----
        constexpr int LOOK = DENSE_SLOTS - 2;
        constexpr int RING = DENSE_SLOTS * DT_HALF;
        auto adv = [](int off, int n) { const int x = off + n * DT_HALF; return x >= RING ? x - RING : x; };
        // sequence index j >= 1 -> (kind, K-tile): (j - 1) % 4 = 0: A0, 1: B1, 2: A1, 3: B0 of the NEXT K-tile
        auto issue = [&](int soff, int tt, int kind) {  // both instructions of this wave's share
            if (kind == 0) { dmaA(soff, tt, 0, 0); dmaA(soff, tt, 0, 1); }
            else if (kind == 1) { dmaB(soff, tt, 1, 0); dmaB(soff, tt, 1, 1); }
            else if (kind == 2) { dmaA(soff, tt, 1, 0); dmaA(soff, tt, 1, 1); }
            else { dmaB(soff, tt + 1, 0, 0); dmaB(soff, tt + 1, 0, 1); }
        };
        dmaB(0, 0, 0, 0); dmaB(0, 0, 0, 1);  // j = 0: B0(0)
#pragma unroll
        for (int j = 1; j <= LOOK; ++j) issue((j % DENSE_SLOTS) * DT_HALF, (j - 1) / 4, (j - 1) % 4);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOOK) : "memory");  // B0(0) has landed
        __builtin_amdgcn_s_barrier();
        u32x4_t fbx[2][2], fby[2][2];
        rdB(0, fbx);                                      // "phase -1": B0 of K-tile 0
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LOOK - 1)) : "memory");  // A0(0) has landed
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs half a phase behind
        int rs = DT_HALF, ws = adv(DT_HALF, LOOK);        // slots of the half-tile this phase reads (j = p + 1) / issues (j = p + 1 + LOOK)
#define DT_WAIT_RING()                                                              \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LOOK - 1)) : "memory");           \
    __builtin_amdgcn_sched_barrier(0);                                              \
    __builtin_amdgcn_s_barrier();                                                   \
    __builtin_amdgcn_sched_barrier(0)
        auto ktile = [&](int t, u32x4_t (&b0)[2][2], u32x4_t (&bo)[2][2]) {  // b0 = B0(t) in registers; bo receives B1(t), then B0(t + 1)
            const int tl = t + LOOK / 4;  // K-tile of the half-tiles issued below (LOOK = 8: t + 2; kinds follow the phase)
            // phase 1: quadrant (0, 0)
            rdA(rs);
            issue(ws, tl, (0 + LOOK) % 4);
            DT_WAIT_RING();
            quadrant(acc[0][0], b0);
            DT_END_PHASE();
            // phase 2: quadrant (0, 1)
            rdB(adv(rs, 1), bo);
            issue(adv(ws, 1), tl, (1 + LOOK) % 4);
            DT_WAIT_RING();
            quadrant(acc[0][1], bo);
            DT_END_PHASE();
            // phase 3: quadrant (1, 1)
            rdA(adv(rs, 2));
            issue(adv(ws, 2), tl, (2 + LOOK) % 4);
            DT_WAIT_RING();
            quadrant(acc[MH - 1][1], bo);
            DT_END_PHASE();
            // phase 4: quadrant (1, 0) -- B0(t) is still in registers; B0 of the next K-tile takes B1's place
            rdB(adv(rs, 3), bo);
            issue(adv(ws, 3), tl, (3 + LOOK) % 4);
            DT_WAIT_RING();
            quadrant(acc[MH - 1][0], b0);
            DT_END_PHASE();
            first_reads = false;
            rs = adv(rs, 4);
            ws = adv(ws, 4);
        };
        static_assert(LOOK % 4 == 0, "the kinds issued by the four phases follow the phase only when the look-ahead is a whole number of K-tiles");
        int t = 0;
        for (; t + 1 < T; t += 2) {
            ktile(t, fbx, fby);
            ktile(t + 1, fby, fbx);
        }
        if (t < T) ktile(t, fbx, fby);
#undef DT_WAIT_RING
    } else if constexpr (MH == 2) {
        // ---- prologue: K-tile 0 whole, then A0 / B0 of K-tile 1 (the order the phases below continue: B1(t+1), A1(t+1), A0(t+2), B0(t+2)) ----
        dma_a(0, 0); dma_b(0, 0); dma_b(0, 1); dma_a(0, 1);
        dma_a(1, 0); dma_b(1, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K-tile 0 has landed (the two youngest half-tiles may fly)
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs half a phase behind

        for (int t = 0; t < T; ++t) {
            const int bo = (t & 1) * BUFSZ;
            // phase 1: quadrant (0, 0)
            read_a(bo, 0);
            read_b(bo, 0, fb0);
            dma_b(t + 1, 1);
            DT_WAIT_BARRIER();
            quadrant(acc[0][0], fb0);
            DT_END_PHASE();
            // phase 2: quadrant (0, 1)
            read_b(bo, 1, fb1);
            dma_a(t + 1, 1);
            DT_WAIT_BARRIER();
            quadrant(acc[0][1], fb1);
            DT_END_PHASE();
            // phase 3: quadrant (1, 1)
            read_a(bo, 1);
            dma_a(t + 2, 0);
            DT_WAIT_BARRIER();
            quadrant(acc[MH - 1][1], fb1);
            DT_END_PHASE();
            // phase 4: quadrant (1, 0) -- B0 is still in registers
            dma_b(t + 2, 0);
            DT_WAIT_BARRIER();
            quadrant(acc[MH - 1][0], fb0);
            DT_END_PHASE();
            first_reads = false;
        }
    } else {
        // ---- 128-row tile: two phases per K-tile, ring of three K-tile buffers.  Issue order (per wave): A.0 A.1 B0.0 | B0.1 B1.0 B1.1 of K-tile t + 2 in the
        //      phases 1 | 2 of K-tile t; the prologue issues K-tiles 0 and 1 in that same order.  A slot is refilled one phase or more after its last read
        //      (A, B0 of K-tile t - 1: phase 1; B1: phase 2 -- both behind the barrier that ends phase 2 of t - 1) and read one phase after the wait that
        //      retired it: phase 1 reads A, B0 of t (retired by phase 2 of t - 1: everything up to B0.1(t), i.e. all but B1(t) + the six of t + 1 = vmcnt(8))
        //      and phase 2 reads B1(t) (retired by phase 1 of t: all but the six of t + 1 and the three just issued = vmcnt(9)) ----
        dma_a1(0, 0, 0, 0); dma_a1(0, 0, 0, 1); dma_b1(0, 0, 0, 0); dma_b1(0, 0, 0, 1); dma_b1(0, 0, 1, 0); dma_b1(0, 0, 1, 1);
        dma_a1(BUFSZ, 1, 0, 0); dma_a1(BUFSZ, 1, 0, 1); dma_b1(BUFSZ, 1, 0, 0); dma_b1(BUFSZ, 1, 0, 1); dma_b1(BUFSZ, 1, 1, 0); dma_b1(BUFSZ, 1, 1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A, B0 of K-tile 0 have landed
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();        // wave row 1 runs half a phase behind
        int bo = 0, bn = 2 * BUFSZ;                       // byte offsets of the buffers of K-tiles t and t + 2
        for (int t = 0; t < T; ++t) {
            // phase 1: quadrant (0, 0)
            read_a(bo, 0);
            read_b(bo, 0, fb0);
            dma_a1(bn, t + 2, 0, 0); dma_a1(bn, t + 2, 0, 1); dma_b1(bn, t + 2, 0, 0);
            DT_WAIT_BARRIER_N(9);
            quadrant(acc[0][0], fb0);
            DT_END_PHASE();
            // phase 2: quadrant (0, 1) -- A is still in registers
            read_b(bo, 1, fb1);
            dma_b1(bn, t + 2, 0, 1); dma_b1(bn, t + 2, 1, 0); dma_b1(bn, t + 2, 1, 1);
            DT_WAIT_BARRIER_N(8);
            quadrant(acc[0][1], fb1);
            DT_END_PHASE();
            first_reads = false;
            bo = bo == (NBUF - 1) * BUFSZ ? 0 : bo + BUFSZ;
            bn = bn == (NBUF - 1) * BUFSZ ? 0 : bn + BUFSZ;
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();        // balances wave row 1's extra barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the over-issued DMAs must not land in LDS after the workgroup has gone
#undef DT_WAIT_BARRIER_N
#undef DT_WAIT_BARRIER
#undef DT_END_PHASE

    // ---- epilogue.  acc[mh][nh][mt][nb][r] = C[m0 + wr*WROWS + mh*64 + mt*16 + 4 fq + r][(nt0 + wc*4 + nh*2 + nb)*16 + fr] ----
    // bf16 outputs go through the wave's own 16 KiB of the (now idle) LDS: each value is written once as bf16 -- after the bias / GELU / SwiGLU
    // arithmetic, i.e. at the reference's rounding point in front of the residual add -- and read back as 16 bytes per lane, so that a store
    // instruction covers 8 rows x 128 contiguous bytes instead of 4 rows x 32 (the accumulator layout) and the residual is read the same way.
    constexpr bool BF16_OUT = EPI != EPI_PARTIAL && EPI != EPI_F32;
    constexpr int WCOLS = EPI == EPI_SWIGLU ? 32 : 64;   // output columns of a wave
    constexpr int RSTRIDE = WCOLS * 2;                   // LDS row stride in bytes: 128 rows x 128 B = the wave's 16 KiB exactly; the 16-byte chunk index of a
                                                         // row is XORed with its fq (rows 4 apart alias on the banks: the four fq groups of a write then spread)
    constexpr int SWS = WCOLS / 32 - 1;                  // chunk XOR = fq << SWS (8 chunks per row: fq * 2, 4 chunks: fq)
    const int col0 = EPI == EPI_SWIGLU ? (nt0 + wc * 4) * 8 : (nt0 + wc * 4) * 16;  // first output column of the wave
    const bool fast = BF16_OUT && (g.n_valid & 7) == 0 && (g.ldo & 7) == 0 && (reinterpret_cast<uintptr_t>(g.out) & 15) == 0 &&
                      (!g.res || ((g.ldres & 7) == 0 && (reinterpret_cast<uintptr_t>(g.res) & 15) == 0));
    if constexpr (BF16_OUT) {
        if (fast) {
            __builtin_amdgcn_s_barrier();  // every wave has finished its last fragment reads (and all DMAs have landed): LDS is free
            unsigned char* mine = smem + wave * 16384;
#pragma unroll
            for (int mh = 0; mh < MH; ++mh)
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    float bv[2] = {0.f, 0.f};
                    if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_GELU || EPI == EPI_BIAS_RES) {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const int col = (nt0 + wc * 4 + nh * 2 + nb) * 16 + fr;
                            bv[nb] = col < g.n_valid ? bf2f(g.bias[col]) : 0.f;
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int lrow = mh * 64 + mt * 16 + fq * 4 + r;
                            if constexpr (EPI == EPI_SWIGLU) {
                                const float v = bfr(silu(bfr(acc[mh][nh][mt][0][r]))) * bfr(acc[mh][nh][mt][1][r]);
                                *reinterpret_cast<bf16_t*>(mine + lrow * RSTRIDE + (((nh * 2 + (fr >> 3)) ^ (fq << SWS)) << 4) + (fr & 7) * 2) = f2bf(v);
                            } else {
#pragma unroll
                                for (int nb = 0; nb < 2; ++nb) {
                                    const float sacc = acc[mh][nh][mt][nb][r];
                                    float v;
                                    if constexpr (EPI == EPI_NONE || EPI == EPI_RES) v = sacc;
                                    else if constexpr (EPI == EPI_BIAS || EPI == EPI_BIAS_RES) v = sacc + bv[nb];
                                    else v = gelu_erf(bfr(sacc + bv[nb]));
                                    *reinterpret_cast<bf16_t*>(mine + lrow * RSTRIDE + (((nh * 4 + nb * 2 + (fr >> 3)) ^ (fq << SWS)) << 4) + (fr & 7) * 2) = f2bf(v);
                                }
                            }
                        }
                }
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's own writes (no other wave touches this region)
            constexpr int CH = WCOLS / 8;        // 16-byte chunks per row
            constexpr int RPI = 64 / CH;         // rows per store instruction
            const int lr = lane / CH, lc = lane % CH;
            const int gcol = col0 + lc * 8;
            const bool cvalid = gcol < g.n_valid;
#pragma unroll
            for (int it = 0; it < WROWS / RPI; ++it) {
                const int lrow = it * RPI + lr;
                const int row = m0 + wr * WROWS + lrow;
                u32x4_t v = *reinterpret_cast<const u32x4_t*>(mine + lrow * RSTRIDE + ((lc ^ (((lrow >> 2) & 3) << SWS)) << 4));
                if (row < g.M && cvalid) {
                    if constexpr (EPI == EPI_RES || EPI == EPI_BIAS_RES) {
                        const u32x4_t rv = *reinterpret_cast<const u32x4_t*>(g.res + (long)row * g.ldres + gcol);
                        v.x = pack_bf(lo_bf(rv.x) + lo_bf(v.x), hi_bf(rv.x) + hi_bf(v.x));
                        v.y = pack_bf(lo_bf(rv.y) + lo_bf(v.y), hi_bf(rv.y) + hi_bf(v.y));
                        v.z = pack_bf(lo_bf(rv.z) + lo_bf(v.z), hi_bf(rv.z) + hi_bf(v.z));
                        v.w = pack_bf(lo_bf(rv.w) + lo_bf(v.w), hi_bf(rv.w) + hi_bf(v.w));
                    }
                    *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(g.out) + (long)row * g.ldo + gcol) = v;
                }
            }
            return;
        }
    }
    // general form (fp32 outputs, ragged n_valid / unaligned rows): straight from the accumulators
#pragma unroll
    for (int mh = 0; mh < MH; ++mh)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * WROWS + mh * 64 + mt * 16 + fq * 4 + r;
                if (row >= g.M) continue;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int nt = nt0 + wc * 4 + nh * 2;
                    if constexpr (EPI == EPI_SWIGLU) {  // (gate, up) = the quadrant's two n-tiles
                        const int col = (nt >> 1) * 16 + fr;
                        if (nt < NTILES && col < g.n_valid)
                            reinterpret_cast<bf16_t*>(g.out)[(long)row * g.ldo + col] = f2bf(bfr(silu(bfr(acc[mh][nh][mt][0][r]))) * bfr(acc[mh][nh][mt][1][r]));
                    } else {
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) {
                            const int col = (nt + nb) * 16 + fr;
                            if (nt + nb >= NTILES || col >= g.n_valid) continue;
                            const float s = acc[mh][nh][mt][nb][r];
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
}


template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_dense_kernel(GemmArgs g, int raster, DenseSched sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 x 64 KiB (256-row tiles) or 3 x 48 KiB (launches with 128-row tiles): all LDS of the kernel (a second object would make hipcc drain the DMAs)
    const int NTILES = g.N >> 4;
    const int KTILES = g.K / DT_K;
    if (!sc.folded) {  // ---- the plain grid: 256-row tiles only ----
        int bx = blockIdx.x, by = blockIdx.y;
        if (raster) {  // XCD-aware tile order: the workgroups an XCD runs at a time form a patch of 8 column blocks x all row blocks (gemm_tiled.hip)
            const int X = (NTILES + DT_NT - 1) / DT_NT, Y = (g.M + DT_M - 1) / DT_M;
            const int n_per = gridDim.x >> 3;
            const int S = (blockIdx.x & 7) * n_per + (blockIdx.x >> 3);
            if (S >= X * Y) return;
            const int patch = S / (8 * Y), r = S - patch * (8 * Y);
            const int pw = min(8, X - patch * 8);
            by = r / pw;
            bx = patch * 8 + r % pw;
        }
        const int ks = (EPI == EPI_PARTIAL && g.ksplit > 1) ? g.ksplit : 1;
        const int slice = (EPI == EPI_PARTIAL) ? blockIdx.z : 0;
        const int T = KTILES / ks;                  // K-tiles of this workgroup
        dense_tile<EPI, 2>(g, smem, by * DT_M, bx * DT_NT, slice, T, (long)slice * T * DT_K);
        return;
    }
    // ---- folded launch: the tiles of a class (256-row, then 128-row) form ONE sequence -- (slice, patch of 8 column blocks x all row blocks of the class, column
    //      fastest), the plain grid's order -- and XCD c (= blockIdx.x & 7: the workgroups that share an L2) takes the c-th eighth of each sequence: it walks, in
    //      dispatch order, its 256-row tiles and then its 128-row tiles, the longest work first, so that the short tiles fill the last round.  (A first form
    //      dealt whole COLUMN blocks to the XCDs: with 4 column blocks -- the encoder's out_proj / fc2 -- half the chip had no work.) ----
    const int xcd = blockIdx.x & 7;
    const int local = blockIdx.x >> 3;
    const int ks = sc.ks;
    const int NF = sc.n_full * sc.X * ks, NH = sc.n_half * sc.X * ks;
    const int f0 = (int)((long)xcd * NF / 8), f1 = (int)((long)(xcd + 1) * NF / 8);
    const int h0 = (int)((long)xcd * NH / 8), h1 = (int)((long)(xcd + 1) * NH / 8);
    const bool half = local >= f1 - f0;
    const int idx = half ? h0 + local - (f1 - f0) : f0 + local;
    if (half && idx >= h1) return;
    const int Y = half ? sc.n_half : sc.n_full;
    const int slice = idx / (Y * sc.X);
    const int S = idx - slice * (Y * sc.X);
    const int patch = S / (8 * Y), r = S - patch * (8 * Y);
    const int pw = min(8, sc.X - patch * 8);
    const int by = r / pw, bx = patch * 8 + r % pw;
    // K-tiles of slice s: KTILES / ks, the first KTILES % ks slices one more (slab s holds the partial sum over exactly these K-tiles)
    const int tq = KTILES / ks, trem = KTILES - tq * ks;
    const int T = tq + (slice < trem ? 1 : 0);
    const long k0 = ((long)slice * tq + min(slice, trem)) * DT_K;
    if (half) dense_tile<EPI, 1>(g, smem, sc.n_full * DT_M + by * (DT_M / 2), bx * DT_NT, slice, T, k0);
    else dense_tile<EPI, 2>(g, smem, by * DT_M, bx * DT_NT, slice, T, k0);
}

static int g_dense_mode = 1;  // tuning hook (gemm_dense_set): 0 = never, 1 = heuristic, 2 = wherever supported
static int g_dense_mix = 0;   // ... / 10: 0 = the model below picks the tile mix, 1 = 256-row tiles only (rounds 1-5), 2 = 128-row tiles only, 100 + h = exactly h row blocks of 128
void gemm_dense_set(int mode) { g_dense_mode = mode % 10; g_dense_mix = mode / 10; }

// ---- tile mix.  List-scheduling model of ONE XCD (32 CUs, its share of the column blocks, tiles dealt in dispatch order to the CU that frees up first): a
//      256-row tile over all of K costs 1, a 128-row tile DENSE_HALF_COST (measured: profiles/r06/dense_half_tiles_probe.txt -- more than 0.5, the half tile
//      streams the same weight bytes for half the MFMAs), every workgroup DENSE_WG_COST on top (DMA ring fill + epilogue). ----
#ifndef DENSE_HALF_COST
#define DENSE_HALF_COST 0.62
#endif
#define DENSE_WG_COST 0.03
static double dense_makespan(int n_full, int n_half, int X, int ks) {
    double cu[32];
    for (int i = 0; i < 32; ++i) cu[i] = 0.0;
    auto run = [&](int count, double cost) {
        for (int k = 0; k < count; ++k) {
            int best = 0;
            for (int i = 1; i < 32; ++i)
                if (cu[i] < cu[best]) best = i;
            cu[best] += cost;
        }
    };
    run((n_full * X * ks + 7) / 8, 1.0 / ks + DENSE_WG_COST);   // (the XCD with the largest eighth)
    run((n_half * X * ks + 7) / 8, DENSE_HALF_COST / ks + DENSE_WG_COST);
    double m = 0.0;
    for (int i = 0; i < 32; ++i) m = cu[i] > m ? cu[i] : m;
    return m;
}
// row blocks of a launch: n_full of 256 rows, then n_half of 128 (n_half == 0: the plain grid).  Returns the model's makespan in units of one 256-row tile over all of K.
double gemm_dense_pick_mix(int M, int N, int ks, int* n_full, int* n_half) {
    const int X = (N / 16 + DT_NT - 1) / DT_NT, cx = X;  // (dense_makespan takes the column blocks of the launch)
    const int blocks128 = (M + 127) / 128;
    int bf = (M + 255) / 256, bh = 0;
    double best = dense_makespan(bf, 0, cx, ks);
    if (g_dense_mix == 1) { *n_full = bf; *n_half = 0; return best; }
    if (g_dense_mix == 2 || g_dense_mix >= 100) {
        const int h = g_dense_mix == 2 ? blocks128 : (g_dense_mix - 100 < blocks128 ? g_dense_mix - 100 : blocks128);
        *n_half = h;
        *n_full = (M - 128 * h + 255) / 256 > 0 ? (M - 128 * h + 255) / 256 : 0;
        return dense_makespan(*n_full, h, cx, ks);
    }
    // Mixes only where the 256-row tiles alone make MORE THAN TWO rounds of the chip: a 128-row tile streams the same weight bytes for half the MFMAs, and inside a
    // step (weights from HBM, not from the Infinity Cache as in a back-to-back probe) it is bound by the L2 -> LDS path -- measured in situ, same box, kernel traces
    // (profiles/r06/trace_same_box_80001*.txt): gate/up at 1408 rows 279.4 -> 264.7 us with 4 x 256 + 3 x 128 rows, but every one-round launch that took half
    // tiles lost (o_proj unsplit 63.7 + 6.0 us against ~50 + 16.5 in two slices; the encoder's out_proj / fc2 likewise)
    if ((long)bf * X * ks <= 2 * 256) { *n_full = bf; *n_half = 0; return best; }
    for (int h = 1; h <= blocks128; ++h) {
        const int f = M - 128 * h > 0 ? (M - 128 * h + 255) / 256 : 0;
        if (f * 256 + h * 128 >= M + 128) continue;  // (a mix that covers a whole spare 128-row block)
        const double c = dense_makespan(f, h, cx, ks);
        if (c < best - 0.02) { best = c; bf = f; bh = h; }  // (a mixed launch has to win by more than its own uncertainty)
    }
    *n_full = bf; *n_half = bh;
    return best;
}

bool gemm_dense_supported(const GemmArgs& g) {
    const int ks = g.ksplit > 1 ? g.ksplit : 1;
    return g.batch == 1 && g.K % DT_K == 0 && (g.K / DT_K) / ks >= 2 && g.lda % 8 == 0 && !g.norm_w && !g.attn_partial && g.M > 64 &&
           (g.epi != EPI_SWIGLU || g.N % 32 == 0);
}
// where the 256-row tiles pay (profiles/dense_probe.py, dense_split_probe.py; gemm_tiled's 128 x 128 tiles keep the rest): from 640 rows on everything
// (704 rows: o_proj + norm 42.8 against 44.6 us, down_proj 94 / 120; 1408: 62 / 82, 161 / 202, gate/up 286 / 372, q/k/v 86 / 100), from 320 rows the
// widest and the deepest projection (352 rows: gate/up 109 / 119, down_proj 60 / 75; o_proj 35 / 31 and q/k/v 70 / 55 stay)
bool gemm_dense_would_run(int M, int N, int K) {
    if (g_dense_mode == 0 || K % 64 != 0 || K < 128) return false;
    if (g_dense_mode == 2) return M > 64;
    return M >= 640 || (M >= 320 && (N >= 16384 || K >= 8192));
}
bool gemm_dense_preferred(const GemmArgs& g) { return gemm_dense_would_run(g.M, g.N, g.K); }

int launch_gemm_dense(const GemmArgs& g, hipStream_t stream) {
    const int NTILES = g.N / 16;
    const int ks = g.epi == EPI_PARTIAL ? (g.ksplit > 1 ? g.ksplit : 1) : 1;
    DenseSched sc{};
    sc.X = (NTILES + DT_NT - 1) / DT_NT;
    sc.ks = ks;
    gemm_dense_pick_mix(g.M, g.N, ks, &sc.n_full, &sc.n_half);
    sc.folded = (sc.n_half > 0 || (g.K / DT_K) % ks != 0) ? 1 : 0;  // (K slices of unequal length exist in the folded grid only)
    dim3 grid(sc.X, (g.M + DT_M - 1) / DT_M, ks), block(512);
    int raster = (grid.x * grid.y > 256 && grid.y > 1) ? 1 : 0;
    size_t lds = DENSE_RING10 ? 10 * DT_HALF : 2 * DT_BUF;
    if (sc.folded) {
        int n_per = 0;
        const int NF = sc.n_full * sc.X * ks, NH = sc.n_half * sc.X * ks;
        for (int c = 0; c < 8; ++c) {
            const int cnt = (int)((long)(c + 1) * NF / 8 - (long)c * NF / 8) + (int)((long)(c + 1) * NH / 8 - (long)c * NH / 8);
            n_per = cnt > n_per ? cnt : n_per;
        }
        grid = dim3(8 * n_per, 1, 1);
        raster = 0;
        if (sc.n_half > 0 && !DENSE_RING10) lds = 3 * 3 * DT_HALF;
    } else if (raster) {
        const unsigned tiles = grid.x * grid.y;
        grid.x = ((tiles + 7) / 8) * 8;
        grid.y = 1;
    }
#define LAUNCH_D(E)                                                                                                                                  \
    do {                                                                                                                                             \
        static bool attr = false;                                                                                                                    \
        if (!attr) {                                                                                                                                 \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dense_kernel<E>), hipFuncAttributeMaxDynamicSharedMemorySize, 10 * DT_HALF) != hipSuccess) return ISST_ERR_HIP; \
            attr = true;                                                                                                                             \
        }                                                                                                                                            \
        hipLaunchKernelGGL(gemm_dense_kernel<E>, grid, block, lds, stream, g, raster, sc);                                                           \
    } while (0)
    switch (g.epi) {
        case EPI_NONE: LAUNCH_D(EPI_NONE); break;
        case EPI_BIAS: LAUNCH_D(EPI_BIAS); break;
        case EPI_BIAS_GELU: LAUNCH_D(EPI_BIAS_GELU); break;
        case EPI_RES: LAUNCH_D(EPI_RES); break;
        case EPI_BIAS_RES: LAUNCH_D(EPI_BIAS_RES); break;
        case EPI_SWIGLU: LAUNCH_D(EPI_SWIGLU); break;
        case EPI_F32: LAUNCH_D(EPI_F32); break;
        case EPI_PARTIAL: LAUNCH_D(EPI_PARTIAL); break;
        default: return ISST_ERR_ARG;
    }
#undef LAUNCH_D
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
