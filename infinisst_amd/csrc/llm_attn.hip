// Llama attention over an UNROTATED KV arena with RoPE applied on read (gfx950, both products on MFMA).
//
// Reference: llama_sdpa_attention_new_forward (model/patches/patch_llm.py:231-336): cache.update(unrotated K, V)
// (:280-284); q rotated at positions past..total-1 and the ENTIRE K cache at 0..total-1 (:286-299); repeat_kv
// (:304-305); causal SDPA (:320-329).  Because the cache is unrotated, dropping the oldest chunks re-indexes the
// remaining keys (reference agents/infinisst.py:340-361).  Here the arena is a per-stream
// [pinned system prompt][ring] and eviction is a ring-start advance: a key's rotation angle is its LOGICAL index
// at read time, so nothing is copied and nothing is re-rotated in memory.
//
// Layout in HBM, per stream, layer and kv head:  K [slots][128] bf16 (row per key, unrotated) and V [slots][128] (row per key: a
// 16-key tile is one contiguous 4 KiB run for either; the transposed [128][slots] plane this kernel used first streamed at 3.4 TB/s
// against 5.6 TB/s for rows, profiles/probes/vt_load_probe.hip), slots = sys_cap + ring_cap (multiples of 64).  Logical position p lives in slot p
// (p < sys_len) or sys_cap + (ring_start + p - sys_len) mod ring_cap.  Attention is a sum over keys, so the kernel
// walks PHYSICAL 16-slot tiles and derives each slot's logical index (rotation angle, causal mask); dead slots
// get probability 0.
//
// RoPE [3P HF apply_rotary_pos_emb, half-split]: out = bf16(bf16(x*cos) + bf16(rotate_half(x)*sin)) with the bf16
// cos/sin table (cos[d+64] == cos[d]).  An MFMA k-step covers 32 dims, so the partner of dim d (d +- 64) sits two
// k-steps away IN THE SAME LANE: the rotation needs no cross-lane traffic.
//
// 4 waves per workgroup, grid = (slot splits, kv heads, row groups); a wave takes every 4th 16-key tile of the workgroup's
// slot span (one tile at one stream, several with a flash-style running softmax at many streams).  A row
// group is a run of consecutive query rows of one stream (1 row in a decode step, up to 16/G rows of a prefill);
// its columns c = (row, head of the kv group) fill the N dimension:
//   S^T[key][c]  = K_rot[key][:] . Q_rot[c][:]        v_mfma_f32_16x16x32_bf16, A = K tile, B = Q^T
//   P = exp(S^T - max_c)                              C layout: lane holds 4 keys of ONE column -> max/sum are
//                                                     in-lane + 2 cross-row shuffles per tile (not per key)
//   O[c][dim]   += P[c][key] V[key][dim]              v_mfma_f32_16x16x16_bf16: its A layout (4 consecutive k per
//                                                     lane) IS the C layout of S^T, so P never leaves registers;
//                                                     B = V through a wave-private LDS image read with ds_read_b64_tr_b16
// Keys written by this launch (logical position >= new_start: the prompt rows of a prefill, the row itself in a
// decode step) are read from the qkv rows; the wave that meets a row's own key appends it (K row + V row) to
// the arena, so every new key is stored exactly once and nobody reads a slot another workgroup writes.
// The 4 waves' partials meet in LDS; one (m, l, o[128]) slab per (row, head, split) goes to global and a small
// second kernel combines the splits (a fused last-arriver combine measured slower, see profiles/r01).
#include <type_traits>
#include "common.h"
#include "kernels.h"

#define HD 128
#define LLM_ATTN_TARGET_WGS 512   // workgroups wanted chip-wide before slot spans grow beyond 64 (swept 256..4096 at 4..64 streams: fewer, longer
                                  // spans win; at 64 streams every (stream, kv head) is ONE workgroup and no combine pass is needed)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// logical position of physical slot t (or -1 for a slot that holds nothing visible)
__device__ __forceinline__ int llm_logical(const LlmStreamView& v, const LlmAttnDims& d, int t, int total) {
    int j;
    if (t < d.sys_cap) {
        j = t < v.sys_len ? t : -1;
    } else {
        int x = t - d.sys_cap - v.ring_start;
        if (x < 0) x += d.ring_cap;
        j = v.sys_len + x;
    }
    return (j >= 0 && j < total) ? j : -1;
}

// rotate the four 8-dim chunks a lane holds of one 128-dim row (dims 8fq + 32s, s = 0..3) with the table entries tab = (cos, sin of dims 8 fq .. + 7, cos, sin of
// dims 32 + 8 fq .. + 7) of its position
__device__ __forceinline__ void rope_row_chunks_tab(const u32x4_t* raw, const u32x4_t* tab, u32x4_t* out) {
    // y[d] = bf16(bf16(x[d] c) - bf16(x[d + 64] s)),  y[d + 64] = bf16(bf16(x[d + 64] c) + bf16(x[d] s))  (HF apply_rotary_pos_emb in bf16: every torch op rounds).
    // Word by word (two dims per 32-bit word): the four products of a dim are rounded by TWO v_cvt_pk_bf16_f32 (one instruction rounds two floats) and the two results
    // of a word are rounded and packed by one -- the element-wise form (a cast per value, then a second cast in pack8) issued ~1.6 x the instructions for the same
    // bits; this function is the key loaders' whole job in llm_attn_prefill_kernel's arena fill (round 6).
#pragma clang fp contract(off)
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // chunk pairs (s = h, s = h + 2): dims d and d + 64
        const u32x4_t cw = tab[2 * h], sw = tab[2 * h + 1];
        u32x4_t o1, o2;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t x1 = raw[h][w], x2 = raw[h + 2][w], c = cw[w], sn = sw[w];
            const uint32_t lo_a = pack_bf(lo_bf(x1) * lo_bf(c), lo_bf(x2) * lo_bf(sn));   // (bf16(x1 c), bf16(x2 s)) of the word's low dim
            const uint32_t lo_b = pack_bf(lo_bf(x2) * lo_bf(c), lo_bf(x1) * lo_bf(sn));   // (bf16(x2 c), bf16(x1 s))
            const uint32_t hi_a = pack_bf(hi_bf(x1) * hi_bf(c), hi_bf(x2) * hi_bf(sn));
            const uint32_t hi_b = pack_bf(hi_bf(x2) * hi_bf(c), hi_bf(x1) * hi_bf(sn));
            o1[w] = pack_bf(lo_bf(lo_a) - hi_bf(lo_a), lo_bf(hi_a) - hi_bf(hi_a));
            o2[w] = pack_bf(lo_bf(lo_b) + hi_bf(lo_b), lo_bf(hi_b) + hi_bf(hi_b));
        }
        out[h] = o1;
        out[h + 2] = o2;
    }
}
__device__ __forceinline__ void rope_tab_load(int pos, int fq, const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin, u32x4_t* tab) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        tab[2 * h] = *reinterpret_cast<const u32x4_t*>(rope_cos + (long)pos * 64 + 32 * h + 8 * fq);
        tab[2 * h + 1] = *reinterpret_cast<const u32x4_t*>(rope_sin + (long)pos * 64 + 32 * h + 8 * fq);
    }
}
// ... at position pos (table entries loaded here)
__device__ __forceinline__ void rope_row_chunks(const u32x4_t* raw, int pos, int fq, const bf16_t* __restrict__ rope_cos,
                                                const bf16_t* __restrict__ rope_sin, u32x4_t* out) {
    u32x4_t tab[4];
    rope_tab_load(pos, fq, rope_cos, rope_sin, tab);
    rope_row_chunks_tab(raw, tab, out);
}

// MULTI: a wave may take several tiles (running softmax, next tile's keys prefetched); false: exactly one tile per wave,
// the lean one-stream form (fewer registers, 3 waves per SIMD)
// tiles a wave keeps in flight ahead of the one it works on (MULTI form).  Measured, same box, 64 streams x 10 passes: 1 -> 93.4 / 93.7 ms per
// step, 2 -> 94.2 / 94.5 (256 VGPRs, no spill): more bytes in flight per wave do not raise the rate the launch streams the caches at
// -DISST_ATTN_TRACE (make trace): thread 0 of every workgroup of the decode attention stamps the 100 MHz wall clock: entry / queries rotated /
// tiles done / slab stored (profiles/attn_trace_probe.py)
#ifdef ISST_ATTN_TRACE
__device__ unsigned long long g_attn_trace[8192 * 8];
#define ATTN_STAMP_W4(i) do { if (threadIdx.x == 256) { const int wg_ = blockIdx.x; if (wg_ < 8192) g_attn_trace[wg_ * 8 + (i)] = wall_clock64(); } } while (0)
#define ATTN_STAMP(i) do { if (threadIdx.x == 0) { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (wg_ < 8192) g_attn_trace[wg_ * 8 + (i)] = wall_clock64(); } } while (0)
extern "C" int isst_debug_attn_trace_read(void* dst, long bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_trace), (size_t)bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define ATTN_STAMP(i) do {} while (0)
#define ATTN_STAMP_W4(i) do {} while (0)
#endif
// exp of the prefill kernel's online softmax: the hardware exp2 on x * log2(e) (two instructions; libm's expf is ~10 with its range handling, and the
// loop is VALU-issue-bound).  Arguments are <= 0 and results feed a bf16 rounding; PF_EXP_LIBM=1 restores expf for A/B runs.
#ifndef PF_EXP_LIBM
#define PF_EXP(x) __builtin_amdgcn_exp2f((x) * 1.44269504088896340736f)
#else
#define PF_EXP(x) expf(x)
#endif
#ifndef ATTN_SLAB_SC1
#define ATTN_SLAB_SC1 0  // 1: partial slabs always stored write-through (experiment: a cheaper end-of-kernel write-back?)
#endif
#ifndef ATTN_KV_NT
#define ATTN_KV_NT 0
#endif
// four waves of a workgroup meet at an LDS counter (zero before the first arrival): each wave's LDS writes are complete before its add, LDS
// executes in order, so a wave that reads 4 finds all of them.  No global-memory wait (a workgroup barrier through __syncthreads has one).
__device__ __forceinline__ void lds_sync4(int* cnt, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

// The body of the decode / small-group attention as a device function: llm_attn_partial_kernel is a thin wrapper, and llm_attn_oproj_kernel (one
// stream's decode step fused with the combine and o_proj, below) runs it on the first four waves of its first workgroups.
//   sp, kvh, zgrp: the (slot split, kv head, row group) this workgroup works on (the wrapper passes blockIdx)
//   slab_sc1: store the partial slabs write-through (another workgroup of the SAME launch will read them)
//   sync4: null -- the four waves meet at the workgroup barrier; else an LDS counter (zeroed by the caller) that they meet at instead, so that
//   further waves of the workgroup (the fused launch's weight-only waves) stay out of it
//   after_fetch(): called once, right after the wave's first key / value loads have been issued (the fused launch puts its o_proj weight loads
//   behind them: vmcnt counts in order, so waiting for the keys leaves the weights in flight)
template <int G, int CT, bool MULTI, class Hook>  // G q-heads per kv head, CT column tiles (16 columns each) per workgroup
__device__ __forceinline__ void llm_attn_partial_body(const bf16_t* __restrict__ qkv, const int* __restrict__ row_stream,
                                                      const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
                                                      const int2* __restrict__ groups, const bf16_t* __restrict__ rope_cos,
                                                      const bf16_t* __restrict__ rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool,
                                                      float* __restrict__ partial, const LlmAttnDims& d, int layer, int n_splits,
                                                      int tiles_per_split, const LlmAttnOne& one, bf16_t* __restrict__ out_direct, int n_extra,
                                                      bf16_t* __restrict__ out_final, int* arrive, int sp, int kvh, int zgrp, bool slab_sc1,
                                                      Hook&& after_fetch, int* sync4 = nullptr) {
    // every mul and add below is rounded on its own, as written: the body is compiled into two kernels whose surrounding code differs, and with
    // contraction left to the compiler the two copies fused different mul/add pairs (a 1-ulp logit now and then between the two paths)
#ifndef ATTN_CONTRACT_ON
#pragma clang fp contract(off)
#endif
    // arrive != null (one row group per launch, more than one split): the last workgroup of a kv head to arrive combines the splits itself
    // n_splits: partial slabs per (row, head) = gridDim.x; the last n_extra (> 0) of them are the per-beam workgroups of the shared-prefix form
    __shared__ float mS[4][CT * 16], lS[4][CT * 16];
    __shared__ float oS[4][CT * 16][HD + 4];  // +4: rows shift by 4 banks
    __shared__ int s_ticket;
    ATTN_STAMP(0);
    const int2 grp = one.enabled ? one.grp : groups[zgrp];
    const int r0 = grp.x, nrows = grp.y, ncols = nrows * G;
    const LlmStreamView v = one.enabled ? one.v : sv[row_stream[r0]];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform: the tile walk below stays in scalar registers)
    const int fr = lane & 15, fq = lane >> 4;
    const int H = d.heads, KV = d.kv_heads;
    const long ldq = (long)(H + 2 * KV) * HD;
    const int slots = d.sys_cap + d.ring_cap;
    // shared-prefix beams, two forms: n_extra > 0 -- the last n_extra workgroups of the x dimension each take ONE beam's own keys (few streams: more
    // workgroups in flight); n_extra < 0 ("folded", one prefix workgroup per (stream, kv head)) -- wave w walks its share of the prefix tiles and then
    // the own tiles of beams w, w + 4, ..: the running softmax carries on (a beam's own keys only count for that beam's columns), the four waves meet
    // in LDS as always and the workgroup writes the attention output itself: no per-beam workgroups, no slabs, no combine launch.  (Measured at
    // 64 streams x 4 beams: the 2 048 short per-beam workgroups interleaved with the 512 long ones in dispatch order kept the long ones from
    // starting together -- 98.6 us per launch against 52 for the same keys without beams, profiles/r04/trace_busy_prof64x4.txt.)
    const bool shared = n_extra != 0 && v.n_beams > 1;
    const bool fold = shared && n_extra < 0;
    const bool tailwg = shared && !fold && sp >= n_splits - n_extra;  // this workgroup: the per-beam keys of beam `wg_beam`
    const int wg_beam = tailwg ? sp - (n_splits - n_extra) : 0;
    const long base = v.kv_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;  // beam 0's arena; beam b's: + b * v.beam_stride
    bf16_t* const kb = kpool + base;    // [slots][128]
    bf16_t* const kr = krpool + base;   // [slots][128] the same keys rotated at their logical position of this chunk (LlmStreamView::rot_keys)
    const bool rot = v.rot_keys != 0;
    bf16_t* const vb = vtpool + base;   // [slots][128] row per key, like K
    const int total = (one.enabled ? one.pos0 + (nrows - 1) * one.pos_step : row_pos[r0 + nrows - 1]) + 1;  // keys visible to the last row of the group
    const float scale = 0.08838834764831845f;       // 1/sqrt(128)
    // Tiles are walked in a COMPACT index space that holds only the tiles with a live slot: [pinned prefix tiles | ring tiles from the one that
    // holds ring_start on, as many as the live span touches].  The arena is sized for max_llm_cache_size + a chunk + slack, the steady-state
    // cache fills ~85 % of it, and a dead tile costs the same 8 KB of loads as a live one (64 streams: 318 MB per launch instead of 272).
    // this wave's prefix tiles: compact indices tile_begin + wave, + 4, ... below tile_end
    const int sys_tiles = (v.sys_len + 15) >> 4, ring_tiles = d.ring_cap >> 4, ring_tile0 = v.ring_start >> 4;
    const int ring_len = total - v.sys_len;
    const int ring_live = ring_len > 0 ? min(ring_tiles, ((v.ring_start & 15) + ring_len + 15) >> 4) : 0;
    const int live_tiles = sys_tiles + ring_live;
    const int tile_begin = sp * tiles_per_split, tile_end = tailwg ? tile_begin : min(tile_begin + tiles_per_split, live_tiles);
    const int n_pref = tile_end > tile_begin + wave ? (tile_end - tile_begin - wave + 3) >> 2 : 0;  // (compact spans need not be multiples of 4)
    // the (<= 4 in the per-beam-workgroup form, host-checked) ring tiles that hold logical positions tail_start .. total-1 of a beam's own arena
    int tail_first = 0, n_tail = 0;
    if (shared) {
        const int s_first = (v.ring_start + (v.tail_start - v.sys_len)) % d.ring_cap, s_last = (v.ring_start + (total - 1 - v.sys_len)) % d.ring_cap;
        tail_first = s_first >> 4;
        n_tail = ((s_last >> 4) - tail_first + ring_tiles) % ring_tiles + 1;
    }
    // own-key tiles of this wave: per-beam workgroup -- the wave-th tile of its beam; folded -- all of them for beams wave, wave + 4, ..
    const int n_own = tailwg ? (wave < n_tail ? 1 : 0) : fold ? n_tail * ((v.n_beams - wave + 3) >> 2) : 0;
    const int n_it = n_pref + n_own;
    // the wave's i-th tile: (physical tile, beam whose arena holds it, index -- compact prefix index or own-tile number --, own-key tile?), advanced
    // incrementally with selects, no branches: the walk is wave-uniform and must not sit between a wave and its next loads
    auto phys_prefix = [&](int tc) -> int {  // compact index -> physical 16-slot tile
        int r = ring_tile0 + (tc - sys_tiles);
        r = r >= ring_tiles ? r - ring_tiles : r;
        return tc < sys_tiles ? tc : (d.sys_cap >> 4) + r;
    };
    auto phys_own = [&](int idx) -> int {
        int r = tail_first + idx;
        r = r >= ring_tiles ? r - ring_tiles : r;
        return (d.sys_cap >> 4) + r;
    };
    const int own_idx0 = tailwg ? wave : 0, own_beam0 = tailwg ? wg_beam : wave;
    auto tile_next = [&](int& tp, int& beam, int& idx, bool& own, int i_next) {  // -> the wave's i_next-th tile (i_next < n_it), from its predecessor
        const bool to_own = i_next >= n_pref;
        const bool wrap = idx + 1 >= n_tail;
        const int oidx = own ? (wrap ? 0 : idx + 1) : own_idx0;
        const int obeam = own ? (wrap ? beam + 4 : beam) : own_beam0;
        const int pidx = idx + 4;
        tp = to_own ? phys_own(oidx) : phys_prefix(pidx);
        beam = to_own ? obeam : 0;
        idx = to_own ? oidx : pidx;
        own = to_own;
    };

    // ---- rotated query fragments: B[k = dim][n = column c], c = ct*16 + fr -> (row r0 + c / G, head kvh*G + c % G) ----
    u32x4_t qf[CT][4];
    int cpos[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = ct * 16 + fr;
        const bool cv = c < ncols;
        const int row = r0 + (cv ? c / G : 0);
        cpos[ct] = cv ? (one.enabled ? one.pos0 + (row - r0) * one.pos_step : row_pos[row]) : -1;
        const bf16_t* qh = qkv + (long)row * ldq + (long)(kvh * G + (cv ? c % G : 0)) * HD;
        u32x4_t qraw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[s] = *reinterpret_cast<const u32x4_t*>(qh + 32 * s + 8 * fq);
        rope_row_chunks(qraw, cv ? cpos[ct] : 0, fq, rope_cos, rope_sin, qf[ct]);
    }

#ifdef ISST_ATTN_TRACE
    asm volatile("s_nop 0" :: "v"(qf[0][0].x), "v"(qf[0][3].w));
#endif
    ATTN_STAMP(1);
    // running softmax state of this wave (flash-style, fp32): per column fr its max and sum; O in the C layout
    float m_run[CT], l_run[CT];
    f32x4_t o[CT][8];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        m_run[ct] = -INFINITY;
        l_run[ct] = 0.f;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) o[ct][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    // what a wave needs to put a tile in flight: this lane's key of the tile as an A-operand row (slot 16 tp + fr of the arena of `beam`) and its
    // value row -- from the arena, or from the qkv rows for keys written by this launch -- and the key's logical position.  Computed one tile AHEAD
    // of the loads (the walk is latency-bound with one tile in flight: every instruction between "the tile in hand has landed" and "the next
    // loads are out" is paid once per tile; with the address arithmetic in that gap the 64-stream launch took 63.5 us instead of 52)
    auto prep = [&](int tp, int beam, const bf16_t*& ks, const bf16_t*& vs, int& jk, bool& k_new) {
        jk = llm_logical(v, d, tp * 16 + fr, total);
        k_new = jk >= 0 && jk >= v.new_start;
        const long boff = (long)beam * v.beam_stride + (long)(tp * 16 + fr) * HD;
        const long qoff = (long)(v.row0 + beam + (jk - v.new_start)) * ldq + (long)(H + kvh) * HD;
        ks = k_new ? qkv + qoff : (rot ? kr : kb) + boff;
        vs = k_new ? qkv + qoff + (long)KV * HD : vb + boff;
    };
    // V tile staging: the lane's value row (16 B x 4, same lane -> (key, dims) map as K) goes to a wave-private LDS image
    // [16 keys][128 dims] (256-byte rows, 16-byte chunks XOR-swizzled so that row writes and transposed reads both spread over the
    // banks) and comes back through ds_read_b64_tr_b16 as the P.V B operand: lane (fr, fq) gets V[key 4fq + j][dim 16nt + fr].
    // The image lives inside this wave's part of the merge buffer oS (free until the end).
    unsigned char* vimg = reinterpret_cast<unsigned char*>(&oS[wave][0][0]);
    auto v_off = [](int row, int ch) -> int { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
    int vw_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) vw_off[s] = v_off(fr, 4 * s + fq);
    const int tq = (lane >> 2) & 3, tp = lane & 3;  // transposed read: lane 4q + p of its 16-lane group addresses row 4g + q, 4 dims at 4p
    int vr_off[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) vr_off[nt] = v_off(4 * fq + tq, 2 * nt + (tp >> 1)) + 8 * (tp & 1);

    // MULTI: the next tile's key and value rows go in flight before this tile's arithmetic (one tile ahead; two measured no faster, see above).
    // The prefetch is unconditional, clamped to the wave's last tile: a conditional load makes hipcc branch around it and drain vmcnt(0).
    int jk_n = -1;
    bool knew_n = false;
    u32x4_t kraw_n[4], vraw_n[4];
    auto fetch = [&](const bf16_t* ks, const bf16_t* vs) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kraw_n[s] = (MULTI && ATTN_KV_NT) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(ks + 32 * s + 8 * fq)) : *reinterpret_cast<const u32x4_t*>(ks + 32 * s + 8 * fq);
#pragma unroll
        for (int s = 0; s < 4; ++s) vraw_n[s] = (MULTI && ATTN_KV_NT) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(vs + 32 * s + 8 * fq)) : *reinterpret_cast<const u32x4_t*>(vs + 32 * s + 8 * fq);
    };
    // cur_*: the tile whose rows are in (or on their way to) the prefetch registers; nxt_*: the tile to request next, with its sources
    const bool pre0 = n_pref > 0;
    int cur_tp = pre0 ? phys_prefix(tile_begin + wave) : phys_own(own_idx0), cur_beam = pre0 ? 0 : own_beam0, cur_idx = pre0 ? tile_begin + wave : own_idx0;
    bool cur_own = !pre0;
    const bf16_t *nks = nullptr, *nvs = nullptr;
    int njk = -1;
    bool nknew = false;
    prep(cur_tp, cur_beam, nks, nvs, jk_n, knew_n);
    if (n_it > 0) fetch(nks, nvs);
    // One tile per wave (the one-stream forms): the rotary table entries the tile's rotation will need are requested HERE, with its key / value rows.  With the
    // rotated-key arena only the tile that holds the launch's own key(s) rotates anything; asked for inside the tile body its table loads sat behind the
    // append stores of that key -- vmcnt counts in order -- and the wave waited ~2.8 us for the stores' acknowledgements: the ONE workgroup per kv head that meets
    // the step's own key had its tile done at 5.8 us where every other one had at 3.0, and the launch waits for its last workgroup
    // (profiles/r06/attn_oproj_trace_per_wg.txt).  The lanes whose rotation is dropped (cached keys, already rotated) read the row of the launch's first own
    // position, which the query rotation has just read.
    u32x4_t tab0[4];
    bool tab0_have = false;
    if constexpr (!MULTI) {
        tab0_have = n_it > 0 && (!rot || __any(knew_n));
        if (tab0_have) rope_tab_load((rot && !knew_n) ? v.new_start : (jk_n >= 0 ? jk_n : 0), fq, rope_cos, rope_sin, tab0);
    }
    after_fetch();
    int nxt_tp = cur_tp, nxt_beam = cur_beam, nxt_idx = cur_idx;
    bool nxt_own = cur_own;
    if constexpr (MULTI) {
        if (n_it > 1) tile_next(nxt_tp, nxt_beam, nxt_idx, nxt_own, 1);
        prep(nxt_tp, nxt_beam, nks, nvs, njk, nknew);
    }
    auto tile_body = [&](int i) {  // the wave's i-th tile (= cur, already in the prefetch registers)
        const int t0 = cur_tp * 16;
        const int beam = cur_beam;
        const bool own = cur_own;
        const long boff = (long)beam * v.beam_stride;
        const int jk = jk_n;
        const bool k_new = knew_n;
        u32x4_t kraw[4], vraw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) { kraw[s] = kraw_n[s]; vraw[s] = vraw_n[s]; }
        if constexpr (MULTI) {
            fetch(nks, nvs);  // (addresses ready since the previous tile)
            jk_n = njk;
            knew_n = nknew;
            __builtin_amdgcn_sched_barrier(0);
            cur_tp = nxt_tp; cur_beam = nxt_beam; cur_idx = nxt_idx; cur_own = nxt_own;
            if (i + 2 < n_it) tile_next(nxt_tp, nxt_beam, nxt_idx, nxt_own, i + 2);
            prep(nxt_tp, nxt_beam, nks, nvs, njk, nknew);
        }
        // One tile per wave: EVERY load of the wave -- key rows, value rows, table entries -- has landed before anything else happens, on every path.  The appends
        // below are conditional stores; hipcc counts such a store as "maybe not issued", so a later wait for an OLDER load is written with a count that, when
        // the stores were issued, also waits for their acknowledgements: ~2.8 us for the one wave per kv head that holds the step's own key, with a one-stream
        // launch waiting for it (profiles/r06/attn_oproj_trace_per_wg.txt).  (Waiting for the value rows here costs the other tiles nothing measurable: they
        // land right behind the key rows.)
        if constexpr (!MULTI) asm volatile("" :: "v"(kraw[3].x), "v"(vraw[0].x), "v"(vraw[1].x), "v"(vraw[2].x), "v"(vraw[3].x), "v"(tab0[0].x), "v"(tab0[1].x), "v"(tab0[2].x), "v"(tab0[3].x));
        const bool tile_live = __any(jk >= 0);
        if (!tile_live) return;
        const bool tile_has_new = __any(k_new);
        // ---- the 4 keys this lane holds in the C layout: slots t0 + 4 fq + r ----
        int jc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) jc[r] = llm_logical(v, d, t0 + 4 * fq + r, total);

        // ---- append this group's own new keys (unrotated K row, V row) ----
        // (the multi-tile walk keeps its appends where the tile is met: compiled out altogether -- timing only -- the 64-stream launch took 52.65 instead of 53.05 us,
        //  64 x 4 beams 63.3 instead of 63.9: profiles/r06/attn_walk_without_appends_timing_only.txt)
        if (k_new && (!shared || own)) {  // (shared-prefix beams: whoever walks beam b's own tiles appends beam b's key to arena b)
            const int krow = v.row0 + beam + (jk - v.new_start);
            if (krow >= r0 && krow < r0 + nrows) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    *reinterpret_cast<u32x4_t*>(kb + boff + (long)(t0 + fr) * HD + 32 * s + 8 * fq) = kraw[s];
                    *reinterpret_cast<u32x4_t*>(vb + boff + (long)(t0 + fr) * HD + 32 * s + 8 * fq) = vraw[s];
                }
            }
        }
        u32x4_t kf[4];
        if (!rot || tile_has_new) {
            if constexpr (!MULTI) {
                rope_row_chunks_tab(kraw, tab0, kf);  // (requested with the tile's rows: tab0_have == this condition)
            } else {
                rope_row_chunks(kraw, jk >= 0 ? jk : 0, fq, rope_cos, rope_sin, kf);
            }
            if (rot && !k_new) {
#pragma unroll
                for (int s = 0; s < 4; ++s) kf[s] = kraw[s];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = kraw[s];
        }
        if (rot && k_new && (!shared || own)) {
            const int krow = v.row0 + beam + (jk - v.new_start);
            if (krow >= r0 && krow < r0 + nrows) {
#pragma unroll
                for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(kr + boff + (long)(t0 + fr) * HD + 32 * s + 8 * fq) = kf[s];
            }
        }
        u32x2_t vf[8];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[s]), __builtin_bit_cast(bf16x8_t, qf[ct][s]), st, 0, 0, 0);
            // lane holds S^T[key 4fq + r][column fr]: causal mask against the column's own position
            float sc[4], mx = -INFINITY;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bool ok = jc[r] >= 0 && jc[r] <= cpos[ct];
                if (shared) ok = ok && (own ? (jc[r] >= v.tail_start && (ct * 16 + fr) / G == beam) : jc[r] < v.tail_start);
                sc[r] = ok ? st[r] * scale : -INFINITY;
                mx = fmaxf(mx, sc[r]);
            }
            // (the two column reductions of a tile as v_permlane16_swap / v_permlane32_swap instead of ds_bpermute: measured, no gain at 1 / 64 streams, greedy
            //  or beam 4 -- profiles/r05/ab_attention_column_reductions_permlane_swap_no_gain.txt; the walk waits for memory, not for these)
            mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
            mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
            const float m_new = fmaxf(m_run[ct], mx);
            // rescale of what this wave has accumulated so far for column fr (1 when the max did not move)
            const float resc = (m_run[ct] == -INFINITY) ? 0.f : expf(m_run[ct] - m_new);
            float p[4], ls = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = (sc[r] == -INFINITY) ? 0.f : expf(sc[r] - m_new);
                ls += p[r];
            }
            ls += __shfl_xor(ls, 16, WAVE);
            ls += __shfl_xor(ls, 32, WAVE);
            l_run[ct] = l_run[ct] * resc + ls;
            m_run[ct] = m_new;
            // O rows are columns 4 fq + r: fetch their factors from the lanes that hold those columns' statistics
            if constexpr (MULTI) {  // rescale only when some column's running max moved (wave-uniform; x * 1.0f is exact: no bit changes)
                if (__any(resc != 1.0f)) {
                    float rs[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) rs[r] = __shfl(resc, 4 * fq + r, WAVE);
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[ct][nt][r] *= rs[r];
                }
            }
            // P (bf16) in the C layout == A operand of the 16x16x16 product: A[row = column fr][k = key 4fq + j]
            u32x2_t pp;
            pp.x = pack_bf(p[0], p[1]);
            pp.y = pack_bf(p[2], p[3]);
            const s16x4_t pa = __builtin_bit_cast(s16x4_t, pp);
            if (ct == 0) {  // values through the wave's LDS image (written as rows, read transposed); EXEC is full here
#pragma unroll
                for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(vimg + vw_off[s]) = vraw[s];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt)
                    vf[nt] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                             (__attribute__((address_space(3))) s16x4_t*)(vimg + vr_off[nt])));
            }
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
                o[ct][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, __builtin_bit_cast(s16x4_t, vf[nt]), o[ct][nt], 0, 0, 0);
        }
    };
    if constexpr (MULTI) {
        for (int i = 0; i < n_it; ++i) tile_body(i);
    } else {
        if (n_it > 0) tile_body(0);
    }
#ifdef ISST_ATTN_TRACE
    asm volatile("s_nop 0" :: "v"(o[0][0][0]), "v"(o[0][7][3]));
#endif
    ATTN_STAMP(2);
    // ---- the 4 waves' partials meet in LDS.  o[ct][nt][r] is O[column ct*16 + 4fq + r][dim 16nt + fr] ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        if (fq == 0) { mS[wave][ct * 16 + fr] = m_run[ct]; lS[wave][ct * 16 + fr] = l_run[ct]; }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oS[wave][ct * 16 + 4 * fq + r][16 * nt + fr] = o[ct][nt][r];
    }
    // (LDS only: __syncthreads() would also wait for every outstanding global load)
    if (sync4) lds_sync4(sync4, lane);
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // four dims per thread: a slab leaves the workgroup as whole 16-byte stores
    const bool inline_combine = arrive != nullptr && !out_direct;
    for (int e = tid; e < ncols * (HD / 4); e += 256) {
        const int c = e / (HD / 4), d4 = (e % (HD / 4)) * 4;
        float M = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, mS[w][c]);
        float L = 0.f;
        f32x4_t O = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = mS[w][c];
            const float f = (mw == -INFINITY) ? 0.f : expf(mw - M);
            L += lS[w][c] * f;
            const f32x4_t ov = *reinterpret_cast<const f32x4_t*>(&oS[w][c][d4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) O[q] += ov[q] * f;
        }
        const int row = r0 + c / G, head = kvh * G + c % G;
        if (out_direct) {  // a single split: this IS the attention output (the combine's arithmetic for one slab)
            u32x2_t pk;
            pk.x = pack_bf(O[0] / L, O[1] / L);
            pk.y = pack_bf(O[2] / L, O[3] / L);
            *reinterpret_cast<u32x2_t*>(out_direct + ((long)row * H + head) * HD + d4) = pk;
        } else {
            float* dst = partial + (((long)row * H + head) * n_splits + sp) * ATTN_SLAB;
            const u32x4_t ob = __builtin_bit_cast(u32x4_t, O);
            u32x4_t sb;
            sb.x = __float_as_uint(M); sb.y = __float_as_uint(L); sb.z = 0u; sb.w = 0u;
            if (inline_combine || slab_sc1 || ATTN_SLAB_SC1) {  // write-through (sc1): the reducing workgroup may sit on another XCD, whose L2 never sees this one's lines
                const __amdgpu_buffer_rsrc_t ds = __builtin_amdgcn_make_buffer_rsrc(dst, 0, ATTN_SLAB * 4, 0x00020000);
                __builtin_amdgcn_raw_buffer_store_b128(ob, ds, (unsigned)d4 * 4u, 0, 16);
                if (d4 == 0) __builtin_amdgcn_raw_buffer_store_b128(sb, ds, HD * 4u, 0, 16);
            } else {
                *reinterpret_cast<u32x4_t*>(dst + d4) = ob;
                if (d4 == 0) *reinterpret_cast<u32x4_t*>(dst + HD) = sb;
            }
        }
    }
    ATTN_STAMP(3);
    if (!inline_combine) return;
    // ---- in-launch combine (cdna_hip_programming.md Guideline 16, the counter form; MI355X_MICROARCH.md visibility table, row 1): every slab byte
    //      was stored sc1; every storing wave drains its stores; after the workgroup's barrier ONE lane adds to the (kv head's) arrival counter at
    //      agent scope; the workgroup whose add returns n_splits - 1 is the last arriver: it alone reads all the slabs of its columns, with sc1
    //      loads only (nothing here reads them through L1), merges them (common.h attn_merge_*: the combine kernel's arithmetic) and writes the
    //      attention output.  It also re-arms the counter: launches are stream-ordered, the next one finds 0 ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_ticket = __hip_atomic_fetch_add(arrive + kvh, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_ticket != n_splits - 1) return;
    for (int c = wave; c < ncols; c += 4) {
        const int row = r0 + c / G, head = kvh * G + c % G;
        AttnMergeLoads<ATTN_MERGE_MAX_SPLITS> ld;
        attn_merge_issue<ATTN_MERGE_MAX_SPLITS, 16>(partial + ((long)row * H + head) * n_splits * ATTN_SLAB, n_splits, lane, ld);
        *reinterpret_cast<uint32_t*>(out_final + ((long)row * H + head) * HD + 2 * lane) = attn_merge_finish<ATTN_MERGE_MAX_SPLITS>(ld, n_splits, lane);
    }
    if (tid == 0) __hip_atomic_store(arrive + kvh, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int G, int CT, bool MULTI>
__global__ __launch_bounds__(256, MULTI ? 2 : 1) void llm_attn_partial_kernel(const bf16_t* __restrict__ qkv, const int* __restrict__ row_stream,
                                                               const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
                                                               const int2* __restrict__ groups, const bf16_t* __restrict__ rope_cos,
                                                               const bf16_t* __restrict__ rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool,
                                                               float* __restrict__ partial, LlmAttnDims d, int layer, int n_splits,
                                                               int tiles_per_split, LlmAttnOne one, bf16_t* __restrict__ out_direct, int n_extra,
                                                               bf16_t* __restrict__ out_final, int* arrive) {
    llm_attn_partial_body<G, CT, MULTI>(qkv, row_stream, row_pos, sv, groups, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer, n_splits,
                                        tiles_per_split, one, out_direct, n_extra, out_final, arrive, blockIdx.x, blockIdx.y, blockIdx.z, false, [] {});
}

// ------------------------------------------------------------------------------------------------------------------------
// One stream's decode step: attention + split-KV combine + o_proj (+ residual) as ONE launch (round 4).
//
// Three dependent launches cost ~19 us per layer at one stream (attention 5.8, combine 4.7, the o_proj GEMV 7.2, two boundaries) for 4 MB of keys and
// 33.5 MB of weights, and the weights only start streaming when the attention output exists.  Here N / 16 workgroups of 8 waves (256 for
// Llama-3.1-8B: one per CU, all resident) do all three:
//   A  every wave fetches ITS k-tiles of the workgroup's 16 o_proj columns into registers (K = 4096: 16 b128 loads, 64 VGPRs: the whole 33.5 MB
//      matrix ends up in the chip's register files).  The first n_splits x kv_heads workgroups run llm_attn_partial_body on waves 0-3 and store
//      their slabs write-through; those four waves ask for their weights only AFTER their slab stores (vmcnt counts in order: s_waitcnt
//      vmcnt(KPW) then means "slabs in memory" while the weights stay in flight) and meet each other at LDS counters, not at the workgroup barrier.
//      All other waves hold their loads back by `delay` x ~0.4 us: 33.5 MB requested at t = 0 put a 4 us queue in front of the attention's
//      dependent round trips (queries rotated at 3.5 us instead of 1.0, profiles/r04/attn_oproj_trace_v*.txt).
//   B  hand-off 1: an attention workgroup adds to ITS kv head's arrival count; workgroup h < heads (x rows: one per (row, head)) polls the count of head h's
//      kv head (on wave 4, whose own loads have landed: a poll is a load, and waiting for it waits for every older load), merges head h's slabs (common.h
//      attn_merge_*, the combine kernel's arithmetic) and stores the 128 outputs write-through as 8-byte words {two bf16 outputs, tag};
//   C  hand-off 2 carries its own validity: tag = the handle's running total of merges after this launch, so a word is either this launch's or it is read
//      again -- no store acknowledgement, no counter.  Wave 4 of every workgroup polls ONE word as a gate, then all threads read the row (8 KB of outputs
//      = 16 KB of words) until every word carries the tag, stage it in LDS and run the skinny GEMV's arithmetic from registers: wave w multiplies
//      k-tiles w, w + 8, .. in ascending order, the eight partial tiles are summed in wave order, out = res + bf16(sum) -- the operations of
//      gemm_skinny_kernel<1, 1, EPI_RES, nt, AMODE 0> at 8 waves in the same order, so the result is bit-identical to the three-launch path
//      (tests/test_gpu_engine.py, test_gpu_fullsize.py).
// Memory ordering across workgroups is the protocol of the in-launch combine above: sc1 stores, a counted wait for them, relaxed agent-scope
// counter adds, sc1 loads on the reading side.  The launch needs every workgroup resident at once (the host checks N / 16 <= CU count; the device must
// not be shared with another process' kernels); every wait is bounded and raises *err (pinned host memory) instead of hanging.
// Measured (profiles/r04/attn_oproj_trace_v4.txt, _v5_tagged_row.txt, fused_attn_oproj_ab_v4.txt, _v5_tagged_row.txt): the launch ends 11.8 us after its first wave
// starts (12.9 before the tagged row; rocprofv3: 15.2 us in its first form against 17.7 us + two boundaries for the three launches); one stream 31.8 -> 30.9 ms per
// chunk.  What is left is a chain of memory round trips behind the attention (store ack, count, slabs, tagged row), each 0.6-1 us while the weights stream.
// bar: 40 x 128-byte lines of unsigned -- [0..31] one arrival count per kv head (the attention workgroups of that head) ([32..39]: unused since the merged row
// carries tags).  The counts only grow (wrap-safe compares): the host keeps the totals every enqueued launch will have brought them to (the handle's launches
// are stream-ordered) and passes a launch its own: arrive_target = arrivals per kv head so far + this launch's splits; merge_target = merges so far + heads x
// rows, used as the tag of this launch's row words.
// delay: x ~0.4 us that the waves WITHOUT attention work hold their weight loads back -- all 33.5 MB requested at t = 0 put the attention's
// dependent round trips (queries, keys, values) behind a 4 us queue (profiles/r04/attn_oproj_trace_v2.txt)
// ------------------------------------------------------------------------------------------------------------------------
#define FUSED_WAVES 8
template <int N>
__device__ __forceinline__ void fused_wait_vmcnt() { __builtin_amdgcn_s_waitcnt((N & 15) | 0x70 | 0xf00 | ((N >> 4) << 14)); }  // s_waitcnt vmcnt(N) alone
#define FUSED_SPIN (1 << 18)
__device__ __forceinline__ unsigned fused_peek(const unsigned* bar, int line) {
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(bar), 0, 40 * 128, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b32(br, (unsigned)line * 128u, 0, 16);
}
// wait until replica `line` has reached `target` (wrap-safe); false: timed out or another workgroup gave up
__device__ __forceinline__ bool fused_wait(const unsigned* bar, int line, unsigned target, int* err) {
    for (int it = 0; it < FUSED_SPIN; ++it) {
        if ((int)(fused_peek(bar, line) - target) >= 0) return true;
        if ((it & 63) == 63 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}
#define FUSED_MAX_ROWS 4
template <int G, int KPW>  // KPW: k-tiles (32 deep) per wave: K = 32 * FUSED_WAVES * KPW
__global__ __launch_bounds__(FUSED_WAVES * 64, 1) void llm_attn_oproj_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ rope_cos,
                                                                              const bf16_t* __restrict__ rope_sin, bf16_t* kpool, bf16_t* krpool,
                                                                              bf16_t* vtpool, float* __restrict__ partial, LlmAttnDims d, int layer,
                                                                              int n_splits, int n_extra, int tiles_per_split, LlmAttnOne one,
                                                                              const bf16_t* __restrict__ Wp, int n_valid, const bf16_t* res,
                                                                              bf16_t* out, int ld, bf16_t* attn_row, unsigned* trow, unsigned* bar, int* err, int mode,
                                                                              unsigned arrive_target, unsigned merge_target, int delay) {
    // n_splits: slabs per (row, head) = the attention workgroups of a kv head (n_extra of them the per-beam workgroups of a shared-prefix beam group);
    // res / out / attn_row: row one.grp.x of the hidden state (row stride ld) and of the attention output (row stride K)
    constexpr int K = 32 * FUSED_WAVES * KPW, KT = K / 32;
    __shared__ __attribute__((aligned(16))) bf16_t xs[FUSED_MAX_ROWS * K];
    __shared__ float red[FUSED_WAVES][FUSED_MAX_ROWS][16];
    __shared__ int s_sync[2];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = d.heads, r0 = one.grp.x, rows = one.grp.y;
    ATTN_STAMP(0);
    if (tid == 0) { s_sync[0] = 0; s_sync[1] = 0; }
    __syncthreads();
    // ---- A: this wave's o_proj weights -> registers ----
    u32x4_t wreg[KPW];
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Wp) + (long)b * KT * 512, 0, KT * 1024, 0x00020000);
    auto issue_w = [&]() {
#pragma unroll
        for (int j = 0; j < KPW; ++j) wreg[j] = __builtin_amdgcn_raw_buffer_load_b128(wrs, (unsigned)lane * 16u, (wave + FUSED_WAVES * j) * 1024, 2);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto hold = [&]() { for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(16); };
    const int n_attn = n_splits * d.kv_heads;
    if (b < n_attn) {
        const int kvh = b / n_splits;
        if (wave < 4) {
            // (these waves ask for their weights only once their slabs are on the way: a wait for the slab stores -- vmcnt counts in order -- then
            //  leaves exactly the KPW weight loads outstanding; asked for earlier, every wait of the attention itself would also wait for them.
            //  They meet each other at LDS counters, not at the workgroup barrier: waves 4-7 are busy getting their loads issued)
            llm_attn_partial_body<G, 1, false>(qkv, nullptr, nullptr, nullptr, nullptr, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer,
                                               n_splits, tiles_per_split, one, nullptr, n_extra, nullptr, nullptr, b % n_splits, kvh, 0, true, [] {}, &s_sync[0]);
            // (compiler fence: fused_wait_vmcnt<KPW> below reads "all but the KPW youngest vector-memory operations are done" as "the slab stores have
            //  landed", which holds only if the KPW weight loads are issued AFTER the body's last slab store and nothing else in between.  sched_barrier
            //  pins the machine schedule, not IR-level motion of a load through another descriptor; this does.)
            asm volatile("" ::: "memory");
            issue_w();
            fused_wait_vmcnt<KPW>();  // this wave's slab stores (write-through) have completed
            lds_sync4(&s_sync[1], lane);
        } else {
            hold();
            issue_w();
        }
        ATTN_STAMP(4);
        // ---- hand-off 1: this workgroup's slabs are in memory ----
        if (tid == 0) __hip_atomic_fetch_add(bar + kvh * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        hold();
        issue_w();
    }
    // hand-off 2 carries its own validity: the merged row travels as 8-byte words {two bf16 outputs, tag}, tag = merge_target (unique per launch: the
    // running total of merges, which also makes stale words from any earlier launch recognisable).  A reader needs no count and the writer no store
    // acknowledgement and no atomic: a word is either this launch's or it is read again (saves the ack wait + the counter's propagation, ~1 us).
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(trow, 0, rows * K * 4, 0x00020000);  // [rows][K / 2] words
    const unsigned tag = merge_target;
    if (wave == 4) {  // (a wave whose weights went out early and have landed: every poll below is a load, and waiting for it waits for all older loads)
        // ---- B: workgroup b merges (row r0 + b / heads, head b % heads), once the workgroups of that head's kv head have arrived ----
        if (b < H * rows) {
            const int mr = b / H, mh = b - mr * H;
            if (fused_wait(bar, mh / G, arrive_target, err)) {
                AttnMergeLoads<ATTN_MERGE_MAX_SPLITS> ldm;
                attn_merge_issue<ATTN_MERGE_MAX_SPLITS, 16>(partial + ((long)(r0 + mr) * H + mh) * n_splits * ATTN_SLAB, n_splits, lane, ldm);
                u32x2_t wv;
                wv.x = attn_merge_finish<ATTN_MERGE_MAX_SPLITS>(ldm, n_splits, lane);
                wv.y = tag;
                __builtin_amdgcn_raw_buffer_store_b64(wv, trs, (unsigned)(mr * (K / 2) + mh * 64 + lane) * 8u, 0, 16);
            }
            ATTN_STAMP_W4(5);
        }
        // ---- C: a cheap gate before everybody reads: ONE word of head b % heads (the heads finish within ~0.4 us of each other) ----
        {
            bool seen = false;
            for (int it = 0; it < FUSED_SPIN && !seen; ++it) {
                const u32x2_t gw = __builtin_amdgcn_raw_buffer_load_b64(trs, (unsigned)((b % H) * 64) * 8u, 0, 16);
                seen = gw.y == tag;
                if (!seen) {
                    if ((it & 63) == 63 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
    }
    __syncthreads();
    ATTN_STAMP(6);
    // the merged rows -> LDS: every thread takes runs of four words (= 8 consecutive outputs) and re-reads a run until all four carry this launch's tag
    for (int p4 = tid; p4 < rows * K / 8; p4 += FUSED_WAVES * 64) {
        u32x4_t a, c;
        bool ok = false;
        for (int it = 0; it < FUSED_SPIN && !ok; ++it) {
            a = __builtin_amdgcn_raw_buffer_load_b128(trs, (unsigned)p4 * 32u, 0, 16);
            c = __builtin_amdgcn_raw_buffer_load_b128(trs, (unsigned)p4 * 32u + 16u, 0, 16);
            ok = a.y == tag && a.w == tag && c.y == tag && c.w == tag;
            if (!ok && (it & 63) == 63 && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) break;
        }
        if (!ok) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const u32x4_t dv = {a.x, a.z, c.x, c.z};
        *reinterpret_cast<u32x4_t*>(xs + p4 * 8) = dv;
    }
    __syncthreads();
    if (mode == 1) {  // (bisecting aid, ISST_FUSE_ATTN_OPROJ=2: attention + combine only; the caller launches the o_proj GEMV on the plain rows)
        if (b == 0)
            for (int c8 = tid * 8; c8 < rows * K; c8 += FUSED_WAVES * 64 * 8) *reinterpret_cast<u32x4_t*>(attn_row + c8) = *reinterpret_cast<const u32x4_t*>(xs + c8);
        return;
    }
    const int arow = lane & 15, kq = (lane >> 4) * 8;
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        const int kt = wave + FUSED_WAVES * j;
        const u32x4_t a = arow < rows ? *reinterpret_cast<const u32x4_t*>(xs + arow * K + kt * 32 + kq) : zero4;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, wreg[j]), acc, 0, 0, 0);
    }
    if (lane < 16) {  // rows 0..3 of the tile: lanes 0..15, registers 0..3
#pragma unroll
        for (int r = 0; r < FUSED_MAX_ROWS; ++r) red[wave][r][lane] = acc[r];
    }
    __syncthreads();
    if (tid < 16 * rows) {
        const int r = tid >> 4, c = tid & 15;
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < FUSED_WAVES; ++w) sum += red[w][r][c];
        const int col = b * 16 + c;
        if (col < n_valid) out[(long)r * ld + col] = f2bf(bf2f(res[(long)r * ld + col]) + bfr(sum));
    }
    ATTN_STAMP(7);
}

// ------------------------------------------------------------------------------------------------------------------------
// Prefill form: one workgroup per (slot split, kv head, UNIT), a unit = up to PREFILL_MAX_WAVES consecutive row groups of one
// stream (a 22-row prompt = 6 groups of 4 rows x G heads = one unit).  The per-group kernel above makes every row group stream the
// stream's whole K/V again (6x the bytes: 270 us per layer at 64 streams, bandwidth-bound); here a key tile is loaded ONCE -- wave 0
// brings the 16 key rows (rotating them if they do not come from the rotated-key arena), wave 1 the 16 value rows -- into
// double-buffered LDS images, and every wave of the unit (one per row group) consumes it for its own 16 columns: keys as A-operand
// rows (ds_read_b128), values transposed (ds_read_b64_tr_b16).  A wave covers all keys of the span for its columns, so there is
// no cross-wave merge; with a single split it writes the attention output itself.
// ------------------------------------------------------------------------------------------------------------------------
#define PREFILL_MAX_WAVES 8
#define PREFILL_MAX_GROUPS 6  // row groups of a unit = consumer waves (the other two waves stage the keys)
#define PF_ST 4  // key / value tiles per staged step (2 loader waves each)
#define LLM_ATTN_PREFILL_TARGET_WGS 512
template <int G>
__global__ __launch_bounds__(PREFILL_MAX_WAVES * 64, 4) void llm_attn_prefill_kernel(  // (2 workgroups per CU: 128 VGPRs, 2 x 64 KB of LDS)
    const bf16_t* __restrict__ qkv, const int* __restrict__ row_stream, const int* __restrict__ row_pos, const LlmStreamView* __restrict__ sv,
    const int2* __restrict__ groups, const int2* __restrict__ units, const bf16_t* __restrict__ rope_cos, const bf16_t* __restrict__ rope_sin,
    bf16_t* kpool, bf16_t* krpool, bf16_t* vpool, float* __restrict__ partial, bf16_t* __restrict__ out_direct, LlmAttnDims d, int layer,
    int n_splits, int tiles_per_split) {
    // [buffer][tile of the stage][16 keys][128 dims], swizzled 16-byte chunks.  A STAGE is PF_ST consecutive live tiles: every one of the 8 waves is a
    // loader (wave w: keys (even w) or values (odd w) of the stage's tile w >> 1), so a workgroup keeps 8 x 4 KB of loads in flight -- with one tile
    // per barrier (2 loader waves, 8 KB in flight per workgroup) 512 workgroups had 4 MB in flight chip-wide and the kernel sat on the memory
    // LATENCY: 272 MB in 197 us = 1.4 TB/s at 64 streams (profiles/r02/bench_kernel_stats_prof64.csv)
    __shared__ __attribute__((aligned(16))) unsigned char kimg[2][PF_ST][4096], vimg[2][PF_ST][4096];
    const int sp = blockIdx.x, kvh = blockIdx.y;
    const int2 unit = units[blockIdx.z];  // (first group, number of groups)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int H = d.heads, KV = d.kv_heads;
    const long ldq = (long)(H + 2 * KV) * HD;
    const int slots = d.sys_cap + d.ring_cap;
    const int2 g_first = groups[unit.x], g_last = groups[unit.x + unit.y - 1];
    const LlmStreamView v = sv[row_stream[g_first.x]];
    const int unit_r0 = g_first.x, unit_rows = g_last.x + g_last.y - g_first.x;
    const int total_u = row_pos[g_last.x + g_last.y - 1] + 1;  // keys visible to the unit's last row
    const long base = v.kv_offset + (long)layer * d.layer_stride + (long)kvh * slots * HD;
    bf16_t* kb = kpool + base;
    bf16_t* kr = krpool + base;
    bf16_t* vb = vpool + base;
    // rot_keys 1: cached keys are read already rotated (the rotated-key arena was filled for this chunk).  2 ("fill", round 5): THIS launch fills it -- the
    // loader waves read the unrotated keys, rotate them for the scores as the rotate-on-read schedule does, and store the rotated tile to the arena on their
    // way to LDS, so the chunk needs no pre-pass over every layer's keys (llm_rope_cache_kernel: 8.4 GB of traffic, 1.8-1.95 ms per 64-stream chunk).  Every
    // live cached tile of the stream is staged by exactly one loader wave per unit of the stream (same bits from every unit); the decode passes then read 1.
    const bool rot = v.rot_keys == 1, fill = v.rot_keys == 2;
    const float scale = 0.08838834764831845f;
    const float c2 = scale * 1.44269504088896340736f;  // 1/sqrt(128) x log2(e): the exponentials below are exp2(s c2 - m c2)
    // tiles are walked in the compact index space of the tiles that hold a live slot (see llm_attn_partial_kernel)
    const int sys_tiles = (v.sys_len + 15) >> 4, ring_tiles = d.ring_cap >> 4, ring_tile0 = v.ring_start >> 4;
    const int ring_len = total_u - v.sys_len;
    const int ring_live = ring_len > 0 ? min(ring_tiles, ((v.ring_start & 15) + ring_len + 15) >> 4) : 0;
    const int tile_begin = sp * tiles_per_split, tile_end = min(tile_begin + tiles_per_split, sys_tiles + ring_live);
    auto phys = [&](int tc) -> int {
        if (tc < sys_tiles) return tc;
        int r = ring_tile0 + (tc - sys_tiles);
        if (r >= ring_tiles) r -= ring_tiles;
        return (d.sys_cap >> 4) + r;
    };
    const bool active = wave < unit.y;  // waves beyond the unit's groups only take part in the barriers
    const int2 grp = groups[unit.x + (active ? wave : 0)];
    const int r0 = grp.x, nrows = grp.y, ncols = nrows * G;

    auto v_off = [](int row, int ch) -> int { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
    int rw_off[4];  // row access: this lane's key fr, chunks 4s + fq (writes by the loader waves, K reads by everyone)
#pragma unroll
    for (int s = 0; s < 4; ++s) rw_off[s] = v_off(fr, 4 * s + fq);
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    int vr_off[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) vr_off[nt] = v_off(4 * fq + tq, 2 * nt + (tp >> 1)) + 8 * (tp & 1);

    // ---- roles (round 6; the launch always has 8 waves).  Waves 0 .. 5 are the CONSUMERS, one per row group of the unit (a unit holds at most
    //      PREFILL_MAX_GROUPS = 6 groups): they issue no global load inside the loop.
    //      Waves 6 and 7 are the LOADERS, two tiles of every stage each, keys and values: they rotate (rotate-on-read, or the fill of the rotated-key arena), append the
    //      unit's own keys and write the key images.  Rounds 2-5 made every wave a loader (even waves keys, odd waves values): the three consumers on even
    //      waves then carried a rotation of 32 dims per lane and stage on top of their four score / softmax / P.V tiles -- waves 0 and 4 share SIMD 0, which
    //      had 2800 issue slots per stage where SIMD 3 had 1000 (profiles/r06/prefill_attention_roles.txt) -- and the rotation's temporaries met the
    //      accumulators in ONE register budget (128 at two workgroups per CU): 25 spilled registers in the loop.  With the roles apart the kernel does not spill. ----
    // (tab: the rotary table entries the tile's rotation will use, requested WITH its rows -- a stage ahead.  Asked for inside commit_k they were a dependent
    //  round trip per tile in front of every rotation, and sat behind the arena stores of the tile before: vmcnt counts in order, and hipcc writes the wait for
    //  a load that is older than a CONDITIONAL store with a count that also waits for the store's acknowledgement once it was issued -- each loader wave spent
    //  ~2 x 2 us per stage on acknowledgements of stores nothing was waiting for, and with the arena fill the loaders are what a stage lasts.)
    struct Fetch { u32x4_t raw[4]; u32x4_t tab[4]; int t0, jk; bool is_new, valid; };
    auto fetch = [&](int stage_first, int slot, bool is_k, Fetch& f) {  // tile `slot` of the stage that starts at compact index stage_first
        const int tc = stage_first + slot;
        f.valid = tc < tile_end;
        f.t0 = phys(f.valid ? tc : tile_begin) * 16;
        f.jk = llm_logical(v, d, f.t0 + fr, total_u);
        f.is_new = f.jk >= 0 && f.jk >= v.new_start;
        const int krow = v.row0 + (f.jk - v.new_start);
        const bf16_t* src;
        if (is_k) src = f.is_new ? qkv + (long)krow * ldq + (long)(H + kvh) * HD : (rot ? kr : kb) + (long)(f.t0 + fr) * HD;
        else src = f.is_new ? qkv + (long)krow * ldq + (long)(H + KV + kvh) * HD : vb + (long)(f.t0 + fr) * HD;
#pragma unroll
        for (int s = 0; s < 4; ++s) f.raw[s] = *reinterpret_cast<const u32x4_t*>(src + 32 * s + 8 * fq);
        // (unconditional: a branch around loads costs the counted waits.  Lanes whose rotation is dropped -- rotated-key arena, cached key -- read the row of the
        //  launch's first own position, which its query rotations have just read)
        if (is_k) rope_tab_load((rot && !f.is_new) ? v.new_start : (f.jk >= 0 ? f.jk : 0), fq, rope_cos, rope_sin, f.tab);
    };
    // every load of both fetches has landed (they were issued a stage ago): from here to the next fetch the loader issues stores only
    auto landed = [&](const Fetch& a, const Fetch& b) {
        asm volatile("" :: "v"(a.raw[0].x), "v"(a.raw[1].x), "v"(a.raw[2].x), "v"(a.raw[3].x), "v"(a.tab[0].x), "v"(a.tab[1].x), "v"(a.tab[2].x), "v"(a.tab[3].x),
                           "v"(b.raw[0].x), "v"(b.raw[1].x), "v"(b.raw[2].x), "v"(b.raw[3].x), "v"(b.tab[0].x), "v"(b.tab[1].x), "v"(b.tab[2].x), "v"(b.tab[3].x));
    };
    auto commit_k = [&](int buf, int slot, const Fetch& f) {
        if (!f.valid) return;  // (wave-uniform)
        const int krow = v.row0 + (f.jk - v.new_start);
        const bool mine = f.is_new && krow >= unit_r0 && krow < unit_r0 + unit_rows;  // a key of this unit's own rows: append it
        u32x4_t kf[4];
        if (!rot || __any(f.is_new)) {
            rope_row_chunks_tab(f.raw, f.tab, kf);
            if (rot && !f.is_new) {
#pragma unroll
                for (int s = 0; s < 4; ++s) kf[s] = f.raw[s];
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) kf[s] = f.raw[s];
        }
        if (mine) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                *reinterpret_cast<u32x4_t*>(kb + (long)(f.t0 + fr) * HD + 32 * s + 8 * fq) = f.raw[s];
                if (rot || fill) *reinterpret_cast<u32x4_t*>(kr + (long)(f.t0 + fr) * HD + 32 * s + 8 * fq) = kf[s];
            }
        } else if (fill && !f.is_new && f.jk >= 0) {  // a cached key: its rotation of this chunk goes to the arena for the decode passes
#pragma unroll
            for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(kr + (long)(f.t0 + fr) * HD + 32 * s + 8 * fq) = kf[s];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(&kimg[buf][slot][rw_off[s]]) = kf[s];
    };
    int st0 = tile_begin;  // compact index of the current stage's first tile
    if (wave >= PREFILL_MAX_GROUPS) {
        // ---- key loader waves: tiles 2 (wave - 6), + 1 of every stage; both tiles' loads stay in flight for a whole stage (these waves hold no accumulators) ----
        const int sa = (wave - PREFILL_MAX_GROUPS) * 2, sb = sa + 1;
#ifndef PF_LOADER_PRIO
#define PF_LOADER_PRIO 2
#endif
        __builtin_amdgcn_s_setprio(PF_LOADER_PRIO);  // with the arena fill a stage lasts as long as its loaders' two rotations: they go first on their SIMDs (the consumers have the slack)
        // Keys: fetched into registers a whole stage ahead, rotated / appended on their way to LDS.  (The value tiles are staged by the consumer waves, by
        // LDS-DMA: below.)  These waves also append the unit's own new value rows to the arena (a register copy, last stages only).
        Fetch FA, FB;
        FA.valid = FB.valid = false; FA.t0 = FB.t0 = 0; FA.jk = FB.jk = -1; FA.is_new = FB.is_new = false;
        auto append_v = [&](int sf, int slot) {
            const int tc = sf + slot;
            if (tc >= tile_end) return;  // (wave-uniform)
            const int t0 = phys(tc) * 16;
            const int jk = llm_logical(v, d, t0 + fr, total_u);
            const bool is_new = jk >= 0 && jk >= v.new_start;
            const int krow = v.row0 + (jk - v.new_start);
            const bool mine = is_new && krow >= unit_r0 && krow < unit_r0 + unit_rows;
            if (__any(mine)) {
                const bf16_t* src = qkv + (long)(mine ? krow : unit_r0) * ldq + (long)(H + KV + kvh) * HD;
                u32x4_t raw[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) raw[s] = *reinterpret_cast<const u32x4_t*>(src + 32 * s + 8 * fq);
                if (mine) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(vb + (long)(t0 + fr) * HD + 32 * s + 8 * fq) = raw[s];
                }
            }
        };
        if (st0 < tile_end) {
            fetch(st0, sa, true, FA); fetch(st0, sb, true, FB);
            landed(FA, FB);
            commit_k(0, sa, FA); commit_k(0, sb, FB);
            append_v(st0, sa); append_v(st0, sb);
            if (st0 + PF_ST < tile_end) { fetch(st0 + PF_ST, sa, true, FA); fetch(st0 + PF_ST, sb, true, FB); }
            else FA.valid = FB.valid = false;
        }
        __syncthreads();
        for (int b = 0; st0 < tile_end; st0 += PF_ST, b ^= 1) {
            if (st0 + PF_ST < tile_end) {
                landed(FA, FB);
                commit_k(b ^ 1, sa, FA); commit_k(b ^ 1, sb, FB);
                append_v(st0 + PF_ST, sa); append_v(st0 + PF_ST, sb);
                if (st0 + 2 * PF_ST < tile_end) { fetch(st0 + 2 * PF_ST, sa, true, FA); fetch(st0 + 2 * PF_ST, sb, true, FB); }
                else FA.valid = FB.valid = false;
            }
            __syncthreads();  // (the consumers' barrier at the end of the stage; no DMA is issued by these waves, so this is a bare s_barrier behind the LDS writes)
        }
        return;
    }
    // ---- consumer waves ----
    // rotated queries of this wave's group
    u32x4_t qf[4];
    const int c = fr;
    const bool cv = active && c < ncols;
    const int crow = r0 + (cv ? c / G : 0);
    const int cpos = cv ? row_pos[crow] : -1;
    {
        const bf16_t* qh = qkv + (long)crow * ldq + (long)(kvh * G + (cv ? c % G : 0)) * HD;
        u32x4_t qraw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qraw[s] = *reinterpret_cast<const u32x4_t*>(qh + 32 * s + 8 * fq);
        rope_row_chunks(qraw, cv ? cpos : 0, fq, rope_cos, rope_sin, qf);
    }
    float m_run = -INFINITY, l_run = 0.f;
    f32x4_t o[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) o[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // smallest position among this wave's real columns (columns past ncols are never stored: they may see anything)
    int cpos_min = cv ? cpos : 0x7fffffff;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) cpos_min = min(cpos_min, __shfl_xor(cpos_min, off, WAVE));
    cpos_min = __builtin_amdgcn_readfirstlane(cpos_min);
    // Value tiles never touch a register: waves 0 .. 3 send tile `wave` of the NEXT stage straight into its LDS image by LDS-DMA (global_load_lds_dwordx4).  The
    // destination of one instruction is 64 consecutive 16-byte slots = rows 4 q .. 4 q + 3 of the tile, so lane (row, slot) fetches the chunk that the image's XOR
    // swizzle puts into that slot.  Issued at the top of a step, the DMA has the whole step to land; the step's closing __syncthreads (vmcnt(0) + barrier while a DMA
    // is in flight) makes it visible.  These waves issue no other vector memory instruction inside the loop.
    auto dma_v = [&](int sf, int buf) {
        const int tc = sf + wave;
        if (wave >= PF_ST || tc >= tile_end) return;  // (wave-uniform)
        const int t0 = phys(tc) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = 4 * q + (lane >> 4), chs = lane & 15;
            const int ch = chs ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const int jk = llm_logical(v, d, t0 + row, total_u);
            const bool is_new = jk >= 0 && jk >= v.new_start;
            const int krow = v.row0 + (jk - v.new_start);
            const bf16_t* src = (is_new ? qkv + (long)krow * ldq + (long)(H + KV + kvh) * HD : vb + (long)(t0 + row) * HD) + ch * 8;
            __builtin_amdgcn_global_load_lds((const void*)src, (__attribute__((address_space(3))) void*)(&vimg[buf][wave][q * 1024]), 16, 0, 0);
        }
    };
    // one stage: keys staged by the loader waves, values by the DMAs issued one step ago
    auto step = [&](int s0, int b) {
        if (s0 + PF_ST < tile_end) dma_v(s0 + PF_ST, b ^ 1);
        if (active) {
            // Round 6: ONE running-softmax update per PAIR of tiles (32 keys) instead of one per 16-key tile.  Per tile the update cost two cross-lane maxima, two
            // cross-lane sums (ds_bpermute round trips on the dependency chain between the two MFMA groups), the exponential of the old maximum, the wave-uniform
            // "did any maximum move" test and its branch -- ~150 issued instructions per tile of which the MFMAs are 12; three consumer waves per SIMD made the
            // launch issue-bound at ~2800 cycles per tile step.  Now the two score tiles of a pair are produced back to back, the statistics are reduced once,
            // and P.V runs on v_mfma_f32_16x16x32_bf16 over PAIRS of tiles: k-slot (fq, j) of the pair is key 4 fq + j of the first tile for j < 4 and of the
            // second for j >= 4 -- exactly the keys this lane holds of each tile in the C layout of the scores, and exactly what two transposed value reads
            // deliver -- so neither P nor V moves between lanes.  (Same arithmetic per key; the running maximum advances in steps of 32 keys, so results
            // differ from the per-tile form in the last bits of a different rounding order -- the parity tests compare against the oracle, not against it.)
            const int n_t = min(PF_ST, tile_end - s0);  // live tiles of this stage (wave-uniform)
#pragma unroll
            for (int pr = 0; pr < PF_ST / 2; ++pr) {  // pairs of tiles: 32 keys per update (all four tiles' scores at once need 16 more registers than the budget of two workgroups per CU has)
                const int n_p = min(2, n_t - 2 * pr);  // live tiles of the pair
                if (n_p <= 0) break;                    // (uniform)
                float sc[2][4];
                float mx = -INFINITY;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int sl = 2 * pr + h2;
                    if (h2 >= n_p) {  // (uniform)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sc[h2][r] = -INFINITY;
                        continue;
                    }
                    const int t0 = phys(s0 + sl) * 16;
                    u32x4_t kf[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const u32x4_t*>(&kimg[b][sl][rw_off[s]]);
                    f32x4_t st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kf[s]), __builtin_bit_cast(bf16x8_t, qf[s]), st, 0, 0, 0);
                    // logical positions of the lane's four keys (slots t0 + 4 fq + r): the tile's base is wave-uniform, so the ring arithmetic is scalar and a
                    // lane only adds its offset and wraps (llm_logical's arithmetic, once per tile instead of once per key)
                    const bool sys_tile = t0 < d.sys_cap;
                    int xb = t0 - d.sys_cap - v.ring_start;
                    if (xb < 0) xb += d.ring_cap;
                    // a tile whose 16 slots are all live and all at or before every column's own position needs no mask at all (every cached tile of a
                    // steady-state chunk but the last two or three): the scalar test below replaces 4 x (ring arithmetic + two compares + select) per lane
                    const int j_last = sys_tile ? t0 + 15 : v.sys_len + xb + 15;
                    const bool plain = (sys_tile ? t0 + 15 < v.sys_len : xb + 15 < d.ring_cap) && j_last < total_u && j_last <= cpos_min;
                    // (scores stay RAW q.k here: the 1/sqrt(d) scale and the change of base ride in ONE fma in front of v_exp_f32 below -- exp2(s c - m c), c = scale log2(e) --
                    //  where the per-tile form spent mul (scale), sub (max), mul (log2 e) per element; the running maximum is kept in raw units)
                    if (plain) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            sc[h2][r] = st[r];
                            mx = fmaxf(mx, sc[h2][r]);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int jc;
                            if (sys_tile) {
                                jc = t0 + 4 * fq + r;
                                if (jc >= v.sys_len) jc = 0x7fffffff;
                            } else {
                                int x = xb + 4 * fq + r;
                                if (x >= d.ring_cap) x -= d.ring_cap;
                                jc = v.sys_len + x;
                            }
                            const bool ok = jc < total_u && jc <= cpos;
                            sc[h2][r] = ok ? st[r] : -INFINITY;
                            mx = fmaxf(mx, sc[h2][r]);
                        }
                    }
                }
                mx = fmaxf(mx, __shfl_xor(mx, 16, WAVE));
                mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
                const float m_new = fmaxf(m_run, mx);
                const float resc = (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f((m_run - m_new) * c2);
                const float mneg = (m_new == -INFINITY) ? 0.f : -m_new * c2;  // (a column without a visible key so far: exp2(-inf c + 0) = 0, not exp2(-inf + inf))
                float ls = 0.f;
                u32x2_t pk[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    float pe[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        pe[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[h2][r], c2, mneg));  // masked keys: exp2(-inf) = 0
                        ls += pe[r];
                    }
                    pk[h2].x = pack_bf(pe[0], pe[1]);
                    pk[h2].y = pack_bf(pe[2], pe[3]);
                }
                ls += __shfl_xor(ls, 16, WAVE);
                ls += __shfl_xor(ls, 32, WAVE);
                l_run = l_run * resc + ls;
                m_run = m_new;
                // rescale of O only when some column's running max moved (wave-uniform test; x * 1.0f is exact, so skipping changes no bit)
                if (__any(resc != 1.0f)) {
                    float rs[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) rs[r] = __shfl(resc, 4 * fq + r, WAVE);
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[nt][r] *= rs[r];
                }
                const int s1 = 2 * pr, s2 = 2 * pr + 1;
                if (n_p == 2) {  // 32 keys in one v_mfma_f32_16x16x32_bf16 per 16 output dims
                    u32x4_t pa;
                    pa.x = pk[0].x; pa.y = pk[0].y; pa.z = pk[1].x; pa.w = pk[1].y;
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) {
                        const u32x2_t v1 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                           (__attribute__((address_space(3))) s16x4_t*)(&vimg[b][s1][vr_off[nt]])));
                        const u32x2_t v2 = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                           (__attribute__((address_space(3))) s16x4_t*)(&vimg[b][s2][vr_off[nt]])));
                        u32x4_t vf;
                        vf.x = v1.x; vf.y = v1.y; vf.z = v2.x; vf.w = v2.y;
                        o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa), __builtin_bit_cast(bf16x8_t, vf), o[nt], 0, 0, 0);
                    }
                } else {         // the odd tile of a span's last stage
                    const s16x4_t pa = __builtin_bit_cast(s16x4_t, pk[0]);
#pragma unroll
                    for (int nt = 0; nt < 8; ++nt) {
                        const u32x2_t vf = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                           (__attribute__((address_space(3))) s16x4_t*)(&vimg[b][s1][vr_off[nt]])));
                        o[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pa, __builtin_bit_cast(s16x4_t, vf), o[nt], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();  // the other buffer is staged, this one is free again
    };
    // ---- stage s + 1 is committed to the other buffer and stage s + 2 requested by the loader waves while stage s is consumed ----
    if (st0 < tile_end) dma_v(st0, 0);
    __syncthreads();
    for (; st0 < tile_end; st0 += 2 * PF_ST) {
        step(st0, 0);
        if (st0 + PF_ST < tile_end) step(st0 + PF_ST, 1);
    }
    if (!active) return;
    // ---- output: o[nt][r] = O[column 4fq + r][dim 16nt + fr]; the column's statistics sit in the lanes fr == column ----
    float Mr[4], Lr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { Mr[r] = __shfl(m_run, 4 * fq + r, WAVE); Lr[r] = __shfl(l_run, 4 * fq + r, WAVE); }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cc = 4 * fq + r;
        if (cc >= ncols) continue;
        const int row = r0 + cc / G, head = kvh * G + cc % G;
        if (out_direct) {
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) out_direct[((long)row * H + head) * HD + 16 * nt + fr] = f2bf(o[nt][r] / Lr[r]);
        } else {
            float* dst = partial + (((long)row * H + head) * n_splits + sp) * ATTN_SLAB;
            if (fr == 0) { dst[HD] = Mr[r] * scale; dst[HD + 1] = Lr[r]; }  // (the running maximum is kept in raw q.k units: the combine pass expects scaled scores)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) dst[16 * nt + fr] = o[nt][r];
        }
    }
}

// Combine of the split partials as a launch of its own (many rows; one decode row merges inside the o_proj GEMV instead: gemm.hip AMODE 3).
// All split loads are issued before the first use (fully unrolled, predicated): two memory round trips instead of one per split.
__global__ __launch_bounds__(64) void llm_attn_combine_kernel(const float* __restrict__ partial, bf16_t* __restrict__ out, int heads, int n_splits) {
    const int h = blockIdx.x, r = blockIdx.y, lane = threadIdx.x;
    const float* src = partial + ((long)r * heads + h) * n_splits * ATTN_SLAB;
    *reinterpret_cast<uint32_t*>(out + ((long)r * heads + h) * HD + 2 * lane) = attn_merge_pair(src, n_splits, lane);
}

int launch_llm_attn_combine(const float* partial, bf16_t* out, int heads, int rows, int n_splits, hipStream_t s) {
    if (n_splits < 1 || n_splits > ATTN_MERGE_MAX_SPLITS) return ISST_ERR_ARG;
    hipLaunchKernelGGL(llm_attn_combine_kernel, dim3(heads, rows), dim3(64), 0, s, partial, out, heads, n_splits);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ---- rotated-key arena fill (once per chunk): one wave per 16-slot tile, same lane -> key mapping and rotation as above ----
__global__ __launch_bounds__(256) void llm_rope_cache_kernel(const LlmStreamView* __restrict__ sv, const bf16_t* __restrict__ rope_cos,
                                                             const bf16_t* __restrict__ rope_sin, const bf16_t* __restrict__ kpool,
                                                             bf16_t* __restrict__ krpool, LlmAttnDims d, int layers) {
    const LlmStreamView v = sv[blockIdx.z];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int slots = d.sys_cap + d.ring_cap;
    const int t = blockIdx.x * 4 + wave;  // tile
    if (t >= (slots >> 4)) return;
    const int total = v.new_start;        // cached keys before this chunk's launches
    const int jk = llm_logical(v, d, t * 16 + fr, total);
    if (!__any(jk >= 0)) return;
    const int lk = blockIdx.y;            // layer * kv_heads + kv head
    const long base = v.kv_offset + (long)(lk / d.kv_heads) * d.layer_stride + (long)(lk % d.kv_heads) * slots * HD + (long)(t * 16 + fr) * HD;
    u32x4_t kraw[4], kf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kraw[s] = *reinterpret_cast<const u32x4_t*>(kpool + base + 32 * s + 8 * fq);
    rope_row_chunks(kraw, jk >= 0 ? jk : 0, fq, rope_cos, rope_sin, kf);
    if (jk >= 0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<u32x4_t*>(krpool + base + 32 * s + 8 * fq) = kf[s];
    }
}

int launch_llm_rope_cache(const LlmStreamView* sv, int n_streams, const bf16_t* rope_cos, const bf16_t* rope_sin, const bf16_t* kpool, bf16_t* krpool,
                          LlmAttnDims d, int layers, hipStream_t s) {
    if (n_streams <= 0) return ISST_OK;
    const int slots = d.sys_cap + d.ring_cap;
    hipLaunchKernelGGL(llm_rope_cache_kernel, dim3((slots / 16 + 3) / 4, layers * d.kv_heads, n_streams), dim3(256), 0, s, sv, rope_cos, rope_sin, kpool,
                       krpool, d, layers);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

static int g_attn_target_wgs = 0, g_attn_prefill_target_wgs = 0;  // profiling aid (isst_op_set_attn_tuning): 0 = the defaults; bits 16.. = prefill
static bool g_attn_no_fold = false;  // A/B aid: target_wgs < 0 keeps the per-beam workgroups of the shared-prefix form at any stream count
void llm_attn_set_tuning(int target_wgs) {
    g_attn_no_fold = target_wgs < 0;
    if (target_wgs < 0) target_wgs = -target_wgs - 1;
    g_attn_target_wgs = target_wgs & 0xffff;
    g_attn_prefill_target_wgs = target_wgs >> 16;
}

template <int G>
static int launch_g(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv, const int2* groups,
                    int n_groups, int max_group_rows, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool,
                    float* partial, LlmAttnDims d, int layer, int n_splits, int tiles_per_split, hipStream_t s, const LlmAttnOne& one, bf16_t* out_direct,
                    int n_extra, bf16_t* out_final, int* arrive) {
    dim3 grid(n_splits, d.kv_heads, n_groups), block(256);  // n_splits includes the n_extra per-beam workgroups
    if (max_group_rows * G > 16) return ISST_ERR_ARG;  // one 16-column tile per workgroup (LLM_ATTN_GROUP_ROWS(G) rows)
    if (tiles_per_split > 4)
        hipLaunchKernelGGL((llm_attn_partial_kernel<G, 1, true>), grid, block, 0, s, qkv, row_stream, row_pos, sv, groups, rope_cos, rope_sin, kpool,
                           krpool, vtpool, partial, d, layer, n_splits, tiles_per_split, one, out_direct, n_extra, out_final, arrive);
    else
        hipLaunchKernelGGL((llm_attn_partial_kernel<G, 1, false>), grid, block, 0, s, qkv, row_stream, row_pos, sv, groups, rope_cos, rope_sin, kpool,
                           krpool, vtpool, partial, d, layer, n_splits, tiles_per_split, one, out_direct, n_extra, out_final, arrive);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

template <int G>
static int launch_prefill_g(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv, const int2* groups, const int2* units,
                            int n_units, int waves, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vpool,
                            float* partial, bf16_t* out_direct, LlmAttnDims d, int layer, int n_splits, int tiles_per_split, hipStream_t s) {
    hipLaunchKernelGGL((llm_attn_prefill_kernel<G>), dim3(n_splits, d.kv_heads, n_units), dim3(64 * waves), 0, s, qkv, row_stream, row_pos, sv, groups,
                       units, rope_cos, rope_sin, kpool, krpool, vpool, partial, out_direct, d, layer, n_splits, tiles_per_split);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

int launch_llm_attention(const bf16_t* qkv, const int* row_stream, const int* row_pos, const LlmStreamView* sv, const int2* groups,
                         int n_groups, int max_group_rows, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool,
                         float* partial, bf16_t* out, LlmAttnDims d, int layer, int rows, hipStream_t s, const LlmAttnOne* one, const int2* units,
                         int n_units, int max_unit_groups, int n_beam_wgs, int* defer_combine, int* arrive_counters) {
    if (rows <= 0 || n_groups <= 0) return ISST_OK;
    LlmAttnOne one1{};
    if (one && one->enabled && n_groups == 1) one1 = *one;
    const int slots = d.sys_cap + d.ring_cap;
    if (slots % 64 != 0 || d.sys_cap % 16 != 0) return ISST_ERR_ARG;
    const int G = d.heads / d.kv_heads;
    const int total_tiles = slots / 16;
    // (up to 4 groups -- 4 streams, or the 4 beams of one -- still prefer one 64-slot span per workgroup: beam 4 36.4 -> 36.2 ms per chunk,
    //  4 greedy streams 40.25 -> 39.97; from 8 groups on the longer spans win: 45.8 vs 46.7 ms)
    const int target = g_attn_target_wgs > 0 ? g_attn_target_wgs : (n_groups <= 4 ? 2 * LLM_ATTN_TARGET_WGS : LLM_ATTN_TARGET_WGS);
    int rc;
    int n_splits;
    bool inline_combine = false;
    if (units && n_units > 0 && max_group_rows > 1) {
        // prefill: a unit (<= 8 row groups of one stream) shares every key tile through LDS (llm_attn_prefill_kernel)
        if (max_unit_groups < 1 || max_unit_groups > PREFILL_MAX_GROUPS || max_group_rows * G > 16) return ISST_ERR_ARG;
        const int ptarget = g_attn_prefill_target_wgs > 0 ? g_attn_prefill_target_wgs : LLM_ATTN_PREFILL_TARGET_WGS;
        n_splits = (ptarget + d.kv_heads * n_units - 1) / (d.kv_heads * n_units);
        n_splits = n_splits < 1 ? 1 : (n_splits > slots / 64 ? slots / 64 : n_splits);
        if (n_splits > ATTN_MERGE_MAX_SPLITS) n_splits = ATTN_MERGE_MAX_SPLITS;  // (very long caches: longer spans instead of more slabs)
        const int tiles_per_split = (total_tiles + n_splits - 1) / n_splits;
        n_splits = (total_tiles + tiles_per_split - 1) / tiles_per_split;
        const int waves = PREFILL_MAX_WAVES;  // every wave is a loader (2 per staged tile); waves beyond the unit's row groups do not consume
        bf16_t* od = n_splits == 1 ? out : nullptr;
        switch (G) {
            case 1: rc = launch_prefill_g<1>(qkv, row_stream, row_pos, sv, groups, units, n_units, waves, rope_cos, rope_sin, kpool, krpool, vtpool, partial, od, d, layer, n_splits, tiles_per_split, s); break;
            case 2: rc = launch_prefill_g<2>(qkv, row_stream, row_pos, sv, groups, units, n_units, waves, rope_cos, rope_sin, kpool, krpool, vtpool, partial, od, d, layer, n_splits, tiles_per_split, s); break;
            case 4: rc = launch_prefill_g<4>(qkv, row_stream, row_pos, sv, groups, units, n_units, waves, rope_cos, rope_sin, kpool, krpool, vtpool, partial, od, d, layer, n_splits, tiles_per_split, s); break;
            default: return ISST_ERR_ARG;
        }
    } else {
        // slot splits: one 64-slot span per workgroup while that fills the chip (one stream: latency), longer spans -- each wave
        // then loops over several tiles with a running softmax -- once (kv heads x row groups) alone provide the workgroups
        // (many streams: the per-workgroup prologue, LDS merge and slab traffic are amortised over more keys)
        n_splits = (target + d.kv_heads * n_groups - 1) / (d.kv_heads * n_groups);
        n_splits = n_splits < 1 ? 1 : (n_splits > slots / 64 ? slots / 64 : n_splits);
        // shared-prefix beam groups from 16 streams on: one span per (stream, kv head), which takes the beams' own keys along (below).  Measured, ms per
        // step at 4 beams, slot splits + per-beam workgroups against one folded span: 8 streams 54.3 / 56.2, 16: 68.8 / 67.1, 32: 90.7 / 83.1,
        // 48: 116.6 / 102.5, 64: 131.6 / 116.1, 128: 221.8 / 198.8 (profiles/r04/beam_attention_fold_{ab,sweep}.txt)
        if (n_beam_wgs > 0 && n_groups >= 16 && g_attn_target_wgs == 0 && !g_attn_no_fold) n_splits = 1;
        if (n_splits + n_beam_wgs > ATTN_MERGE_MAX_SPLITS) n_splits = ATTN_MERGE_MAX_SPLITS - n_beam_wgs;
        const int tiles_per_split = ((total_tiles + n_splits - 1) / n_splits + 3) / 4 * 4;
        n_splits = (total_tiles + tiles_per_split - 1) / tiles_per_split;
        // shared-prefix beam groups (LlmStreamView::n_beams): one more workgroup (and partial slab) per beam (each of its waves takes one tile,
        // in either form of the kernel)
        // ... unless one workgroup per (stream, kv head) already walks the whole prefix: its waves then take the beams' own tiles too ("folded", see the
        // kernel) and the workgroup writes the attention output itself
        const bool fold = n_beam_wgs > 0 && n_splits == 1 && tiles_per_split > 4 && !g_attn_no_fold;
        const int n_extra = fold ? -n_beam_wgs : n_beam_wgs;
        if (!fold) n_splits += n_beam_wgs;
        // one row group (one stream's decode step) whose workgroups all fit the chip at one per CU: the last workgroup of a kv head to arrive
        // combines the splits inside this launch -- no combine launch, no boundary (the hand-off form this is measured for: one workgroup per CU)
        int hip_cus = 256;
        inline_combine = arrive_counters != nullptr && n_groups == 1 && n_beam_wgs == 0 && n_splits > 1 && d.kv_heads <= 64 && n_splits * d.kv_heads <= hip_cus;
        bf16_t* od = n_splits == 1 ? out : nullptr;
        int* arr = inline_combine ? arrive_counters : nullptr;
        switch (G) {
            case 1: rc = launch_g<1>(qkv, row_stream, row_pos, sv, groups, n_groups, max_group_rows, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer, n_splits, tiles_per_split, s, one1, od, n_extra, out, arr); break;
            case 2: rc = launch_g<2>(qkv, row_stream, row_pos, sv, groups, n_groups, max_group_rows, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer, n_splits, tiles_per_split, s, one1, od, n_extra, out, arr); break;
            case 4: rc = launch_g<4>(qkv, row_stream, row_pos, sv, groups, n_groups, max_group_rows, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer, n_splits, tiles_per_split, s, one1, od, n_extra, out, arr); break;
            default: return ISST_ERR_ARG;
        }
    }
    if (rc != ISST_OK) return rc;
    if (defer_combine) *defer_combine = 0;
    if (n_splits == 1 || inline_combine) return ISST_OK;  // the attention kernel wrote the output itself
    if (n_splits > ATTN_MERGE_MAX_SPLITS) return ISST_ERR_ARG;  // (<= 2048 slots)
    if (defer_combine) {  // the caller's o_proj GEMV merges the partials while it stages its A row (gemm.hip AMODE 3): no combine launch
        *defer_combine = n_splits;
        return ISST_OK;
    }
    hipLaunchKernelGGL(llm_attn_combine_kernel, dim3(d.heads, rows), dim3(64), 0, s, partial, out, d.heads, n_splits);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// One stream's decode step (one row, or the <= 4 beams of a shared-prefix beam group): attention + combine + o_proj + residual in one launch
// (llm_attn_oproj_kernel).  > 0: the slot splits of the prefix it will use; 0: the shape is outside what the launch covers (the caller runs the three launches).
int llm_attn_oproj_supported(const LlmAttnDims& d, int rows, int n_groups, const LlmAttnOne* one, int N, int K, int n_cus, int n_beam_wgs) {
    if (!one || !one->enabled || n_groups != 1 || rows < 1 || rows > FUSED_MAX_ROWS || one->grp.y != rows) return 0;
    const int G = d.heads / d.kv_heads;
    if (G != 1 && G != 2 && G != 4) return 0;
    if (n_beam_wgs > 0 ? (n_beam_wgs != rows || rows * G > 16 || one->v.n_beams != rows || one->pos_step != 0) : rows != 1) return 0;
    if (K != d.heads * HD || (K != 32 * FUSED_WAVES * 16 && K != 32 * FUSED_WAVES * 2) || N % 128 != 0) return 0;  // (4096: Llama-3.1-8B; 512: the test configs)
    const int nwg = N / 16;
    if (nwg > n_cus || nwg < d.heads * rows || d.kv_heads > 32) return 0;  // (one merging workgroup per (row, head))
    const int slots = d.sys_cap + d.ring_cap;
    if (slots % 64 != 0 || d.sys_cap % 16 != 0) return 0;
    int n_splits = nwg / d.kv_heads - n_beam_wgs;  // 64-slot spans (one tile per wave) while they fit the launch
    if (n_splits > slots / 64) n_splits = slots / 64;
    if (n_splits < 1 || n_splits + n_beam_wgs > ATTN_MERGE_MAX_SPLITS) return 0;
    const int tiles_per_split = ((slots / 16 + n_splits - 1) / n_splits + 3) / 4 * 4;
    return tiles_per_split <= 4 ? n_splits : 0;  // (longer caches than 64 slots x the workgroups available: the three-launch path)
}

// arrive_total / merge_total: the handle's running totals of arrivals per kv head and of merges (the device-side counts only grow); advanced here once the
// launch is enqueued
int launch_llm_attn_oproj(const bf16_t* qkv, const bf16_t* rope_cos, const bf16_t* rope_sin, bf16_t* kpool, bf16_t* krpool, bf16_t* vtpool, float* partial,
                          LlmAttnDims d, int layer, const LlmAttnOne& one, int n_beam_wgs, const bf16_t* Wp, int N, int K, int n_valid, const bf16_t* res,
                          bf16_t* out, int ld, bf16_t* attn_row, unsigned* trow, unsigned* bar, int* err, int n_cus, hipStream_t s, unsigned* arrive_total,
                          unsigned* merge_total, int mode, int delay, unsigned arrive_bias) {
    const int rows = one.grp.y;
    const int n_prefix = llm_attn_oproj_supported(d, rows, 1, &one, N, K, n_cus, n_beam_wgs);
    if (n_prefix <= 0) return ISST_ERR_ARG;
    const int slots = d.sys_cap + d.ring_cap;
    const int tiles_per_split = ((slots / 16 + n_prefix - 1) / n_prefix + 3) / 4 * 4;
    const int n_splits = n_prefix + n_beam_wgs;
    const unsigned arrive_target = *arrive_total + (unsigned)n_splits + arrive_bias, merge_target = *merge_total + (unsigned)(d.heads * rows);  // (arrive_bias: test aid -- a count that is never reached)
    const int G = d.heads / d.kv_heads;
    dim3 grid(N / 16), block(FUSED_WAVES * 64);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, block, 0, s, qkv, rope_cos, rope_sin, kpool, krpool, vtpool, partial, d, layer, n_splits, n_beam_wgs, tiles_per_split, one, Wp,
                           n_valid, res, out, ld, attn_row, trow, bar, err, mode, arrive_target, merge_target, delay);
    };
    const bool big = K == 32 * FUSED_WAVES * 16;
    switch (G) {
        case 1: if (big) go(llm_attn_oproj_kernel<1, 16>); else go(llm_attn_oproj_kernel<1, 2>); break;
        case 2: if (big) go(llm_attn_oproj_kernel<2, 16>); else go(llm_attn_oproj_kernel<2, 2>); break;
        case 4: if (big) go(llm_attn_oproj_kernel<4, 16>); else go(llm_attn_oproj_kernel<4, 2>); break;
        default: return ISST_ERR_ARG;
    }
    if (hipGetLastError() != hipSuccess) return ISST_ERR_HIP;
    *arrive_total = arrive_target - arrive_bias;
    *merge_total = merge_target;
    return ISST_OK;
}
