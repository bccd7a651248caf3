// Epilogue shared by the 13..64-row packed-weight GEMMs (gemm_mid.hip, gemm_ring.hip): the per-wave partial sums sit in LDS as
// red[wk][(mt * WN + n) * 4 + r][lane] (fp32, KW k-step waves); this sums them in fixed order and applies bias / GELU / residual / SwiGLU with the
// reference's bf16 rounding points, or -- EPI_PARTIAL -- writes the K slice's fp32 slab and, with GemmArgs::tickets, performs the launch-free
// residual (+ sums of squares for the consumer's RMSNorm) as the last-arriving K-slice workgroup of its column block.
#pragma once
#include "common.h"

template <int MT, int WN, int KW, int NW, int EPI>
__device__ __forceinline__ void mid_epilogue(const GemmArgs& g, const float* red, int tid, int m0, int* s_tk_p) {
    constexpr int TILES = MT * WN;
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
            if (tid == 0) *s_tk_p = __hip_atomic_fetch_add(g.tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*s_tk_p != (int)gridDim.y - 1) return;
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
