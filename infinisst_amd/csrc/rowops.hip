// Row-wise HBM-bound kernels of the InfiniSST hot path (gfx950): first conv layer, LayerNorm(+GELU),
// RMSNorm, embedding gather + speech splice, audio cast.  One wave (or block) per row, 16-byte accesses.
#include "common.h"
#include "kernels.h"

// ------------------------------------------------------------------------------------------------
// fp32 PCM -> bf16 (reference agents/infinisst.py:222 casts the waveform to the model dtype)
// ------------------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2bf(src[i]);
}
// all streams of a call at once: window[i] = [history of stream sids[i] | bf16(pcm[i])]  (one launch instead of a copy + a cast per stream:
// at 64 streams the per-stream form was 192 dependent operations, ~20 us of queue turnaround each)
// (ptrs != null: the samples of stream i are at ptrs[i] -- audio the caller already holds in HBM; else at pcm + i * n_samples)
__global__ void audio_window_kernel(const float* __restrict__ pcm, const float* const* __restrict__ ptrs, const int* __restrict__ sids,
                                    const bf16_t* __restrict__ hist_pool, long histp, bf16_t* __restrict__ window, long winp, int hist, int n_samples) {
    const int i = blockIdx.y;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= hist + n_samples) return;
    const float* src = ptrs ? ptrs[i] : pcm + (long)i * n_samples;
    window[i * winp + j] = j < hist ? hist_pool[sids[i] * histp + j] : f2bf(src[j - hist]);
}
int launch_audio_window(const float* pcm, const float* const* ptrs, const int* sids, const bf16_t* hist_pool, long histp, bf16_t* window, long winp,
                        int hist, int n_samples, int n, hipStream_t s) {
    if (n <= 0) return ISST_OK;
    hipLaunchKernelGGL(audio_window_kernel, dim3((unsigned)((hist + n_samples + 255) / 256), n), dim3(256), 0, s, pcm, ptrs, sids, hist_pool, histp,
                       window, winp, hist, n_samples);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
// history for the next chunk: the last `hist` samples of every stream's window (the window is not modified, so no ordering hazard)
__global__ void audio_hist_save_kernel(const bf16_t* __restrict__ window, long winp, const int* __restrict__ sids, bf16_t* __restrict__ hist_pool,
                                       long histp, int hist, int win) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < hist) hist_pool[sids[i] * histp + j] = window[i * winp + (win - hist) + j];
}
int launch_audio_hist_save(const bf16_t* window, long winp, const int* sids, bf16_t* hist_pool, long histp, int hist, int win, int n, hipStream_t s) {
    if (n <= 0 || hist <= 0) return ISST_OK;
    hipLaunchKernelGGL(audio_hist_save_kernel, dim3((unsigned)((hist + 255) / 256), n), dim3(256), 0, s, window, winp, sids, hist_pool, histp, hist, win);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
// --rope 0 (patch_speech_encoder.py:448-461, :488-493): x[s][t][:] += sinusoid(prefix_s + t), bf16 + bf16 -> bf16.  The reference keeps the positions
// themselves in bf16, so above 256 several frames share one row: the table holds one row per bf16 integer -- rows 0..255 = positions 0..255, row
// 256 + i = the bf16 value with bits 0x4380 + i (256, 258, ..., 512, 516, ...) -- built on the host with the reference's own arithmetic (rope.py)
__global__ void __launch_bounds__(128) enc_add_position_kernel(bf16_t* __restrict__ x, const EncStreamView* __restrict__ ev, const bf16_t* __restrict__ table,
                                                                int table_rows, int Q, int D) {
    const int t = blockIdx.x, s = blockIdx.y;
    const int p = ev[s].prefix + t;
    int row = p;
    if (p >= 256) row = 256 + ((int)f2bf((float)p) - 0x4380);
    row = row < table_rows ? row : table_rows - 1;
    bf16_t* xr = x + ((size_t)s * Q + t) * D;
    const bf16_t* tr = table + (size_t)row * D;
    for (int c = threadIdx.x * 8; c < D; c += 128 * 8) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(xr + c), b = *reinterpret_cast<const u32x4_t*>(tr + c);
        float fa[8], fb[8];
        unpack8(a, fa);
        unpack8(b, fb);
        u32x4_t o;
        o.x = pack_bf(fa[0] + fb[0], fa[1] + fb[1]);
        o.y = pack_bf(fa[2] + fb[2], fa[3] + fb[3]);
        o.z = pack_bf(fa[4] + fb[4], fa[5] + fb[5]);
        o.w = pack_bf(fa[6] + fb[6], fa[7] + fb[7]);
        *reinterpret_cast<u32x4_t*>(xr + c) = o;
    }
}
int launch_enc_add_position(bf16_t* x, const EncStreamView* ev, const bf16_t* table, int table_rows, int n, int Q, int D, hipStream_t s) {
    if (n <= 0 || Q <= 0) return ISST_OK;
    if (D % 8 || !table || table_rows < 257) return ISST_ERR_ARG;
    hipLaunchKernelGGL(enc_add_position_kernel, dim3((unsigned)Q, (unsigned)n), dim3(128), 0, s, x, ev, table, table_rows, Q, D);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
int launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t s) {
    if (n <= 0) return ISST_OK;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// conv layer 0: Conv1d(1 -> C, k, stride, bias) + LayerNorm over channels + GELU, fused.
// [3P fairseq ConvFeatureExtractionModel(mode=layer_norm)], call site patch_speech_encoder.py:245-251.
// One wave per output frame; the k input samples are wave-uniform, each lane owns channels lane, lane+64, ...
// (C <= 64*CPL).  Rounding points: conv(+bias) -> bf16, LN -> bf16, GELU -> bf16.
// ------------------------------------------------------------------------------------------------
template <int CPL>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const bf16_t* __restrict__ audio, long audio_batch,
                                                            const bf16_t* __restrict__ w /*[C][k]*/,
                                                            const bf16_t* __restrict__ bias,
                                                            const bf16_t* __restrict__ ln_w,
                                                            const bf16_t* __restrict__ ln_b, bf16_t* __restrict__ out,
                                                            long out_batch, int T, int C, int k, int stride) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= T) return;
    const bf16_t* a = audio + (long)blockIdx.y * audio_batch + (long)t * stride;
    float x[16];
    for (int j = 0; j < k; ++j) x[j] = bf2f(a[j]);
    float v[CPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        float acc = 0.f;
        if (c < C) {
            for (int j = 0; j < k; ++j) acc += x[j] * bf2f(w[c * k + j]);
            if (bias) acc += bf2f(bias[c]);
            acc = bfr(acc);
            sum += acc;
        }
        v[i] = acc;
    }
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
        if (lane + 64 * i < C) { const float d = v[i] - mean; sq += d * d; }
    const float rstd = rsqrtf(wave_sum(sq) / C + 1e-5f);
    bf16_t* o = out + (long)blockIdx.y * out_batch + (long)t * C;
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int c = lane + 64 * i;
        if (c < C) {
            const float y = bfr((v[i] - mean) * rstd * bf2f(ln_w[c]) + bf2f(ln_b[c]));
            o[c] = f2bf(gelu_erf(y));
        }
    }
}


// The full-size form (C == 512, k == KK): a wave walks FR consecutive frames of one stream.  Each lane owns 8 CONSECUTIVE channels -- their KK x 8 weights,
// bias and LayerNorm affine stay in registers for all FR frames (the one-frame kernel above re-reads 80 two-byte weights per lane and frame: at 64 streams
// that is 16 M scattered vector loads, 594 us for 2 GFLOP) -- the frames' input samples are loaded once per wave (lane l holds samples l and l + 64 of the
// wave's window) and broadcast by v_readlane, and a frame's 8 outputs leave as one 16-byte store.  Same rounding points, taps summed in the same order.
template <int KK, int FR>
__global__ __launch_bounds__(256) void conv0_frames_kernel(const bf16_t* __restrict__ audio, long audio_batch, const bf16_t* __restrict__ w /*[512][KK]*/,
                                                           const bf16_t* __restrict__ bias, const bf16_t* __restrict__ ln_w, const bf16_t* __restrict__ ln_b,
                                                           bf16_t* __restrict__ out, long out_batch, int T, int stride) {
    constexpr int C = 512;
    static_assert((KK * 8) % 8 == 0 && (FR - 1) * 7 + KK <= 128, "window of a wave (stride <= 7): two samples per lane");
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int t0 = (blockIdx.x * 4 + wv) * FR;
    if (t0 >= T) return;
    const int nfr = min(FR, T - t0);
    const bf16_t* a = audio + (long)blockIdx.y * audio_batch + (long)t0 * stride;
    const int span = (nfr - 1) * stride + KK;  // (the launcher guarantees stride <= 7: span <= 128)
    const int s0 = lane < span ? __float_as_int(bf2f(a[lane])) : 0;
    const int s1 = lane + 64 < span ? __float_as_int(bf2f(a[lane + 64])) : 0;
    float wr[8 * KK];  // wr[i * KK + j]: tap j of channel 8 lane + i
#pragma unroll
    for (int q = 0; q < KK; ++q) {
        float f[8];
        unpack8(*reinterpret_cast<const u32x4_t*>(w + (long)lane * (8 * KK) + q * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) wr[q * 8 + e] = f[e];
    }
    float bv[8], gw[8], gb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = 0.f;
    if (bias) unpack8(*reinterpret_cast<const u32x4_t*>(bias + lane * 8), bv);
    unpack8(*reinterpret_cast<const u32x4_t*>(ln_w + lane * 8), gw);
    unpack8(*reinterpret_cast<const u32x4_t*>(ln_b + lane * 8), gb);
    bf16_t* o = out + (long)blockIdx.y * out_batch + (long)t0 * C + lane * 8;
    for (int f = 0; f < nfr; ++f) {
        float x[KK];
#pragma unroll
        for (int j = 0; j < KK; ++j) {
            const int idx = f * stride + j;  // wave-uniform
            x[j] = __int_as_float(idx < 64 ? __builtin_amdgcn_readlane(s0, idx) : __builtin_amdgcn_readlane(s1, idx - 64));
        }
        float v[8], sum = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < KK; ++j) acc += x[j] * wr[i * KK + j];
            if (bias) acc += bv[i];
            v[i] = bfr(acc);
            sum += v[i];
        }
        const float mean = wave_sum(sum) / C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[i] - mean; sq += d * d; }
        const float rstd = rsqrtf(wave_sum(sq) / C + 1e-5f);
        float y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = gelu_erf(bfr((v[i] - mean) * rstd * gw[i] + gb[i]));
        *reinterpret_cast<u32x4_t*>(o + (long)f * C) = pack8(y);
    }
}

int launch_conv0(const bf16_t* audio, long audio_batch, const bf16_t* w, const bf16_t* bias, const bf16_t* ln_w,
                 const bf16_t* ln_b, bf16_t* out, long out_batch, int T, int C, int k, int stride, int batch,
                 hipStream_t s) {
    if (T <= 0) return ISST_OK;
    if (k > 16 || C > 512) return ISST_ERR_ARG;
    if (C == 512 && k == 10 && stride <= 7 && (reinterpret_cast<uintptr_t>(w) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && out_batch % 8 == 0) {
        // wav2vec2's first layer.  Many frames (several streams): 16 per wave; one stream's 3 199 frames: 4 per wave, so that 200 workgroups remain
        if ((long)batch * T >= 32768) {
            hipLaunchKernelGGL((conv0_frames_kernel<10, 16>), dim3((T + 63) / 64, batch), dim3(256), 0, s, audio, audio_batch, w, bias, ln_w, ln_b, out, out_batch, T, stride);
        } else {
            hipLaunchKernelGGL((conv0_frames_kernel<10, 4>), dim3((T + 15) / 16, batch), dim3(256), 0, s, audio, audio_batch, w, bias, ln_w, ln_b, out, out_batch, T, stride);
        }
        return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    }
    dim3 grid((T + 3) / 4, batch), block(256);
    if (C <= 64)
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<1>, grid, block, 0, s, audio, audio_batch, w, bias, ln_w, ln_b, out, out_batch, T, C, k, stride);
    else
        hipLaunchKernelGGL(conv0_ln_gelu_kernel<8>, grid, block, 0, s, audio, audio_batch, w, bias, ln_w, ln_b, out, out_batch, T, C, k, stride);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim (+ optional GELU), one wave per row, 8 bf16 per lane per step.
// torch semantics: statistics and affine in fp32, one rounding to bf16; GELU of the bf16 value, rounded again.
// ------------------------------------------------------------------------------------------------
// Optional split-K prologue (slabs != null; encoder out_proj / fc2 at 65..1024 rows): x = bf16(x + bf16(sum of slabs + bias)) is
// written back first -- the residual update of the projection whose K slices the dense kernel left as fp32 slabs -- then normalised.
// acc[0..7] = sum over the K slices' fp32 slabs, ascending slice order (the order every reducer of this library uses), of the 8 elements at `off`.
// Up to 8 slices: every load is issued before the first add (the runtime-bounded loop below waits for each slice's round trip in turn: at 256 rows one
// workgroup per row then sits through 4-8 serial round trips); slices past n_slabs re-read slice 0 and are dropped by a select AFTER the add, so the
// arithmetic is exactly the loop's.  rmsnorm_reduce_kernel: 7.08 -> 6.05 us per launch over a 128-stream step (A/B of library builds, same box).
template <int NMAX>
__device__ __forceinline__ void slab_sum_upto(const float* __restrict__ slabs, long slab_stride, int n_slabs, long off, float (&acc)[8]) {
    f32x4_t a[NMAX], b[NMAX];
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        const float* sp = slabs + (long)(k < n_slabs ? k : 0) * slab_stride + off;
        a[k] = *reinterpret_cast<const f32x4_t*>(sp);
        b[k] = *reinterpret_cast<const f32x4_t*>(sp + 4);
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        const bool on = k < n_slabs;
        acc[0] = on ? acc[0] + a[k].x : acc[0]; acc[1] = on ? acc[1] + a[k].y : acc[1]; acc[2] = on ? acc[2] + a[k].z : acc[2]; acc[3] = on ? acc[3] + a[k].w : acc[3];
        acc[4] = on ? acc[4] + b[k].x : acc[4]; acc[5] = on ? acc[5] + b[k].y : acc[5]; acc[6] = on ? acc[6] + b[k].z : acc[6]; acc[7] = on ? acc[7] + b[k].w : acc[7];
    }
}
__device__ __forceinline__ void slab_sum8(const float* __restrict__ slabs, long slab_stride, int n_slabs, long off, float (&acc)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (n_slabs <= 4) {  // (o_proj's 4 slices at 65..256 rows: half the load instructions of the 8-wide form)
        slab_sum_upto<4>(slabs, slab_stride, n_slabs, off, acc);
        return;
    }
    if (n_slabs <= 8) {
        slab_sum_upto<8>(slabs, slab_stride, n_slabs, off, acc);
        return;
    }
    for (int k = 0; k < n_slabs; ++k) {
        const float* sp = slabs + (long)k * slab_stride + off;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(sp), b = *reinterpret_cast<const f32x4_t*>(sp + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
}

// LF: the slabs of a step are summed loads-first (slab_sum8: every slice's loads in flight before the first add).  Few rows (one stream's 48 frames: 12 workgroups,
// nothing to hide a round trip behind) are a chain of n_slabs dependent round trips otherwise; at many rows (one wave per row, thousands of waves) the plain loop
// measured faster (16.4 against 20.5 us at 128 streams), so the launcher picks by the row count.
template <int STEPS, bool LF = false>
__global__ __launch_bounds__(256) void layernorm_kernel(bf16_t* x, long ldx, const bf16_t* __restrict__ w,
                                                        const bf16_t* __restrict__ b, bf16_t* out, long ldo,
                                                        int rows, int C, float eps, int gelu, const float* __restrict__ slabs, long slab_stride,
                                                        int n_slabs, const bf16_t* __restrict__ proj_bias, const bf16_t* __restrict__ tin, long ldt) {
    // (tin != null -- the library-GEMM path: the projection's bf16 output, bias included, waits in tin: x = bf16(x + t) is written back first)
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* xr = x + row * ldx;
    float v[STEPS][8];
    float sum = 0.f;
    // weight and bias are asked for with the row (the encoder's widths, STEPS <= 2): read after the two wave reductions they were a dependent round trip
    // at the end of a 6 us launch
    constexpr bool PRE = STEPS <= 2;
    u32x4_t wraw[PRE ? STEPS : 1], braw[PRE ? STEPS : 1];
    if constexpr (PRE) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int c = (lane + 64 * s) * 8;
            const bool on = w && c < C;
            wraw[s] = on ? *reinterpret_cast<const u32x4_t*>(w + c) : (u32x4_t){0u, 0u, 0u, 0u};
            braw[s] = on ? *reinterpret_cast<const u32x4_t*>(b + c) : (u32x4_t){0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (lane + 64 * s) * 8;
        if (c < C) {
            unpack8(*reinterpret_cast<const u32x4_t*>(xr + c), v[s]);
            if (slabs) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pb[8];
                const u32x4_t praw = *reinterpret_cast<const u32x4_t*>(proj_bias + c);  // (ahead of the slices, not behind their sums)
                if constexpr (LF) slab_sum8(slabs, slab_stride, n_slabs, row * C + c, acc);
                else
                for (int k = 0; k < n_slabs; ++k) {  // (slab_sum8's loads-first form measured SLOWER here at many rows -- 16.4 -> 20.5 us at 128 streams: one wave per row, registers)
                    const float* sp = slabs + (long)k * slab_stride + row * C + c;
                    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(sp), a2 = *reinterpret_cast<const f32x4_t*>(sp + 4);
                    acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
                    acc[4] += a2.x; acc[5] += a2.y; acc[6] += a2.z; acc[7] += a2.w;
                }
                unpack8(praw, pb);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] = bfr(v[s][j] + bfr(acc[j] + pb[j]));
                *reinterpret_cast<u32x4_t*>(xr + c) = pack8(v[s]);
            } else if (tin) {
                float tv[8];
                unpack8(*reinterpret_cast<const u32x4_t*>(tin + row * ldt + c), tv);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[s][j] = bfr(v[s][j] + tv[j]);
                *reinterpret_cast<u32x4_t*>(xr + c) = pack8(v[s]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[s][j];
        }
    }
    if (!w) return;
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
        if ((lane + 64 * s) * 8 < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[s][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    bf16_t* orow = out + row * ldo;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (lane + 64 * s) * 8;
        if (c < C) {
            float wv[8], bv[8], y[8];
            if constexpr (PRE) {
                unpack8(wraw[s], wv);
                unpack8(braw[s], bv);
            } else {
                unpack8(*reinterpret_cast<const u32x4_t*>(w + c), wv);
                unpack8(*reinterpret_cast<const u32x4_t*>(b + c), bv);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                y[j] = (v[s][j] - mean) * rstd * wv[j] + bv[j];
                if (gelu) y[j] = gelu_erf(bfr(y[j]));
            }
            *reinterpret_cast<u32x4_t*>(orow + c) = pack8(y);
        }
    }
}

// Few rows (one stream's 48 frames = 12 workgroups), a projection's K slices to sum, C <= 1024, <= NMAX slices: ALL loads first.  layernorm_kernel<2, true> still walks
// the row in two steps with the x store of step 0 between them, and vmcnt counts loads and stores in order: the wait for step 1's loads is also a wait for that store's
// acknowledgement (rmsnorm_reduce_lf_kernel below has the long form of this).  Every load of the row -- x, projection bias, slices, weight, bias --
// is in flight before the first add here, and nothing is stored before the last one has landed.  Same arithmetic, same order as the stepwise kernel.
template <int NMAX>
__global__ __launch_bounds__(256) void layernorm_reduce_lf_kernel(bf16_t* x, long ldx, const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, bf16_t* out, long ldo,
                                                                  int rows, int C, float eps, int gelu, const float* __restrict__ slabs, long slab_stride,
                                                                  int n_slabs, const bf16_t* __restrict__ proj_bias) {
    constexpr int STEPS = 2;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    bf16_t* xr = x + row * ldx;
    u32x4_t wraw[STEPS], braw[STEPS], xraw[STEPS], praw[STEPS];
    f32x4_t sa[STEPS][NMAX], sb[STEPS][NMAX];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (lane + 64 * s) * 8, cl = c < C ? c : 0;
        wraw[s] = *reinterpret_cast<const u32x4_t*>((w ? w : proj_bias) + cl);
        braw[s] = *reinterpret_cast<const u32x4_t*>((w ? b : proj_bias) + cl);
        xraw[s] = *reinterpret_cast<const u32x4_t*>(xr + cl);
        praw[s] = *reinterpret_cast<const u32x4_t*>(proj_bias + cl);
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            const float* sp = slabs + (long)(k < n_slabs ? k : 0) * slab_stride + row * C + cl;
            sa[s][k] = *reinterpret_cast<const f32x4_t*>(sp);
            sb[s][k] = *reinterpret_cast<const f32x4_t*>(sp + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(sa[STEPS - 1][NMAX - 1].x), "v"(sb[STEPS - 1][NMAX - 1].x));  // (returns are counted in order: the last one in is every one in)
    float v[STEPS][8];
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (lane + 64 * s) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, pb[8];
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            const bool on = k < n_slabs;
            const f32x4_t a = sa[s][k], a2 = sb[s][k];
            acc[0] = on ? acc[0] + a.x : acc[0]; acc[1] = on ? acc[1] + a.y : acc[1]; acc[2] = on ? acc[2] + a.z : acc[2]; acc[3] = on ? acc[3] + a.w : acc[3];
            acc[4] = on ? acc[4] + a2.x : acc[4]; acc[5] = on ? acc[5] + a2.y : acc[5]; acc[6] = on ? acc[6] + a2.z : acc[6]; acc[7] = on ? acc[7] + a2.w : acc[7];
        }
        unpack8(xraw[s], v[s]);
        unpack8(praw[s], pb);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] = bfr(v[s][j] + bfr(acc[j] + pb[j]));
        if (c < C) {
            *reinterpret_cast<u32x4_t*>(xr + c) = pack8(v[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += v[s][j];
        }
    }
    if (!w) return;
    const float mean = wave_sum(sum) / C;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s)
        if ((lane + 64 * s) * 8 < C) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[s][j] - mean; sq += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(sq) / C + eps);
    bf16_t* orow = out + row * ldo;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (lane + 64 * s) * 8;
        if (c < C) {
            float wv[8], bv[8], y[8];
            unpack8(wraw[s], wv);
            unpack8(braw[s], bv);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                y[j] = (v[s][j] - mean) * rstd * wv[j] + bv[j];
                if (gelu) y[j] = gelu_erf(bfr(y[j]));
            }
            *reinterpret_cast<u32x4_t*>(orow + c) = pack8(y);
        }
    }
}

// ISST_LN_LF_ROWS=n: the all-loads-first kernel up to n rows (0: the stepwise kernel everywhere).  Default: every row count with <= 4 slices (3072 rows, 64 streams:
// 13.2 -> 12.3 us per launch), up to 256 rows with 5..8 (170 registers per wave: measured at 192 rows only) -- profiles/r06/reduce_kernels_many_rows_ab.txt
static int g_rms_lf_rows = -1, g_ln_lf_rows = -1;  // isst_op_set_reduce_tuning (test aid): >= 0 overrides the environment
void reduce_set_tuning(int rms_lf_rows, int ln_lf_rows) { g_rms_lf_rows = rms_lf_rows; g_ln_lf_rows = ln_lf_rows; }
static int ln_all_loads_first_rows(int n_slabs) {
    static const int n = [] { const char* e = getenv("ISST_LN_LF_ROWS"); return e ? atoi(e) : -1; }();
    if (g_ln_lf_rows >= 0) return g_ln_lf_rows;
    return n >= 0 ? n : (n_slabs <= 4 ? 1 << 30 : 256);
}
static int launch_layernorm_impl(bf16_t* x, long ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, long ldo, int rows, int C, float eps,
                                 int gelu, const float* slabs, long slab_stride, int n_slabs, const bf16_t* proj_bias, hipStream_t s,
                                 const bf16_t* tin = nullptr, long ldt = 0) {
    if (rows <= 0) return ISST_OK;
    if (C % 8 != 0 || C > 4096 || ldx % 8 != 0 || ldo % 8 != 0) return ISST_ERR_ARG;
    dim3 grid((rows + 3) / 4), block(256);
    if (C <= 512)
        hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias, tin, ldt);
    else if (C <= 1024 && slabs && rows <= ln_all_loads_first_rows(n_slabs) && n_slabs <= 8 && n_slabs > 4 && proj_bias && !tin)
        hipLaunchKernelGGL(layernorm_reduce_lf_kernel<8>, grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias);
    else if (C <= 1024 && slabs && rows <= ln_all_loads_first_rows(n_slabs) && n_slabs <= 4 && proj_bias && !tin)
        hipLaunchKernelGGL(layernorm_reduce_lf_kernel<4>, grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias);
    else if (C <= 1024 && slabs && rows <= 256)
        hipLaunchKernelGGL((layernorm_kernel<2, true>), grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias, tin, ldt);
    else if (C <= 1024)
        hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias, tin, ldt);
    else
        hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, ldx, w, b, out, ldo, rows, C, eps, gelu, slabs, slab_stride, n_slabs, proj_bias, tin, ldt);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

int launch_layernorm(const bf16_t* x, long ldx, const bf16_t* w, const bf16_t* b, bf16_t* out, long ldo, int rows, int C,
                     float eps, int gelu, hipStream_t s) {
    return launch_layernorm_impl(const_cast<bf16_t*>(x), ldx, w, b, out, ldo, rows, C, eps, gelu, nullptr, 0, 0, nullptr, s);
}

// x[rows][C] (in place) += projection slabs + bias, then out = LayerNorm(x) (w == null: update only)
int launch_layernorm_reduce(const float* slabs, long slab_stride, int n_slabs, const bf16_t* proj_bias, bf16_t* x, long ldx, const bf16_t* w,
                            const bf16_t* b, bf16_t* out, long ldo, int rows, int C, float eps, hipStream_t s) {
    if (!slabs || n_slabs < 1 || !proj_bias) return ISST_ERR_ARG;
    return launch_layernorm_impl(x, ldx, w, b, out, ldo, rows, C, eps, 0, slabs, slab_stride, n_slabs, proj_bias, s);
}

// ------------------------------------------------------------------------------------------------
// RMSNorm [3P HF LlamaRMSNorm]: var = mean(x^2) in fp32; (x * rsqrt(var+eps)) -> bf16; * weight -> bf16.
// One 256-thread block per row (D <= 256*8*STEPS); optional row gather (rows_idx) so that only the rows that
// need the final norm (last prompt position of each stream) are touched.
// ------------------------------------------------------------------------------------------------
template <int STEPS>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const bf16_t* __restrict__ x, long ldx, const int* __restrict__ rows_idx,
                                                      const bf16_t* __restrict__ w, bf16_t* __restrict__ out, long ldo, int D,
                                                      float eps) {
    __shared__ float part[4];
    const long row = rows_idx ? rows_idx[blockIdx.x] : blockIdx.x;
    const bf16_t* xr = x + row * ldx;
    float v[STEPS][8];
    float sq = 0.f;
    u32x4_t wraw[STEPS];  // (asked for with the row, not after the barrier)
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        wraw[s] = c < D ? *reinterpret_cast<const u32x4_t*>(w + c) : (u32x4_t){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        if (c < D) {
            unpack8(*reinterpret_cast<const u32x4_t*>(xr + c), v[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += v[s][j] * v[s][j];
        }
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float r = rsqrtf(tot / D + eps);
    bf16_t* orow = out + (long)blockIdx.x * ldo;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        if (c < D) {
            float wv[8], y[8];
            unpack8(wraw[s], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = wv[j] * bfr(v[s][j] * r);
            *reinterpret_cast<u32x4_t*>(orow + c) = pack8(y);
        }
    }
}

int launch_rmsnorm(const bf16_t* x, long ldx, const int* rows_idx, const bf16_t* w, bf16_t* out, long ldo, int rows, int D,
                   float eps, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (D % 8 != 0 || D > 8192 || ldx % 8 != 0 || ldo % 8 != 0) return ISST_ERR_ARG;
    if (D <= 2048)
        hipLaunchKernelGGL(rmsnorm_kernel<1>, dim3(rows), dim3(256), 0, s, x, ldx, rows_idx, w, out, ldo, D, eps);
    else if (D <= 4096)
        hipLaunchKernelGGL(rmsnorm_kernel<2>, dim3(rows), dim3(256), 0, s, x, ldx, rows_idx, w, out, ldo, D, eps);
    else
        hipLaunchKernelGGL(rmsnorm_kernel<4>, dim3(rows), dim3(256), 0, s, x, ldx, rows_idx, w, out, ldo, D, eps);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// split-K epilogue + RMSNorm in one pass (rows > GEMM_FUSED_NORM_MAX_M): x = bf16(x + bf16(sum_s slab[s])) is written
// back (the residual stream), then normalised as above.  The slab sum runs in slice order: deterministic.
template <int STEPS>
__global__ __launch_bounds__(256) void rmsnorm_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int n_slabs, bf16_t* x,
                                                             long ldx, const bf16_t* __restrict__ w, bf16_t* __restrict__ out, long ldo,
                                                             int D, float eps) {
    __shared__ float part[4];
    const long row = blockIdx.x;
    bf16_t* xr = x + row * ldx;
    float v[STEPS][8];
    float sq = 0.f;
    // (the norm weight is asked for FIRST: read after the row's barrier it was one more dependent round trip at the end of a 8 us launch)
    u32x4_t wraw[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        wraw[s] = (w && c < D) ? *reinterpret_cast<const u32x4_t*>(w + c) : (u32x4_t){0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        if (c < D) {
            float acc[8];
            const u32x4_t xraw = *reinterpret_cast<const u32x4_t*>(xr + c);  // (asked for ahead of the slices: behind slab_sum8's branches it was a round trip of its own)
            slab_sum8(slabs, slab_stride, n_slabs, row * D + c, acc);
            unpack8(xraw, v[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[s][j] = bfr(v[s][j] + bfr(acc[j]));
            *reinterpret_cast<u32x4_t*>(xr + c) = pack8(v[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += v[s][j] * v[s][j];
        }
    }
    if (!w) return;
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float r = rsqrtf(tot / D + eps);
    bf16_t* orow = out + row * ldo;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        if (c < D) {
            float wv[8], y[8];
            unpack8(wraw[s], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = wv[j] * bfr(v[s][j] * r);
            *reinterpret_cast<u32x4_t*>(orow + c) = pack8(y);
        }
    }
}

// The same pass with ALL loads first -- written for the decode launches of 65..256 rows (one workgroup per row on one CU each: nothing hides a round trip), and
// the faster one at the prefill's 1408 rows too.  The kernel above walks
// a row in STEPS steps of: slab loads -> wait -> the x chunk (asked for behind the sums) -> wait -> store x; and because vmcnt counts loads and stores in order, the
// wait for step 1's loads also waits for the acknowledgement of step 0's store: four dependent load round trips and a store round trip per 8 us launch.  Here every
// load of the row -- norm weight, x, every slice of every step -- is in flight before the first add, and every one has landed before the first store (DESIGN.md
// section 7, "what a conditional store does to counted waits"); loads of columns past D re-read column 0 and are dropped.  Same arithmetic, same order.
template <int STEPS, int NMAX>
__global__ __launch_bounds__(256) void rmsnorm_reduce_lf_kernel(const float* __restrict__ slabs, long slab_stride, int n_slabs, bf16_t* x, long ldx,
                                                                const bf16_t* __restrict__ w, bf16_t* __restrict__ out, long ldo, int D, float eps) {
    __shared__ float part[4];
    const long row = blockIdx.x;
    bf16_t* xr = x + row * ldx;
    u32x4_t wraw[STEPS], xraw[STEPS];
    f32x4_t sa[STEPS][NMAX], sb[STEPS][NMAX];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8, cl = c < D ? c : 0;
        wraw[s] = *reinterpret_cast<const u32x4_t*>((w ? w : xr) + cl);
        xraw[s] = *reinterpret_cast<const u32x4_t*>(xr + cl);
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            const float* sp = slabs + (long)(k < n_slabs ? k : 0) * slab_stride + row * D + cl;
            sa[s][k] = *reinterpret_cast<const f32x4_t*>(sp);
            sb[s][k] = *reinterpret_cast<const f32x4_t*>(sp + 4);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(sa[STEPS - 1][NMAX - 1].x), "v"(sb[STEPS - 1][NMAX - 1].x));  // (returns are counted in order: the last one in is every one in)
    float v[STEPS][8];
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {
            const bool on = k < n_slabs;
            const f32x4_t a = sa[s][k], b = sb[s][k];
            acc[0] = on ? acc[0] + a.x : acc[0]; acc[1] = on ? acc[1] + a.y : acc[1]; acc[2] = on ? acc[2] + a.z : acc[2]; acc[3] = on ? acc[3] + a.w : acc[3];
            acc[4] = on ? acc[4] + b.x : acc[4]; acc[5] = on ? acc[5] + b.y : acc[5]; acc[6] = on ? acc[6] + b.z : acc[6]; acc[7] = on ? acc[7] + b.w : acc[7];
        }
        unpack8(xraw[s], v[s]);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[s][j] = bfr(v[s][j] + bfr(acc[j]));
        if (c < D) {
            *reinterpret_cast<u32x4_t*>(xr + c) = pack8(v[s]);
#pragma unroll
            for (int j = 0; j < 8; ++j) sq += v[s][j] * v[s][j];
        }
    }
    if (!w) return;
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = sq;
    __syncthreads();
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float r = rsqrtf(tot / D + eps);
    bf16_t* orow = out + row * ldo;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const int c = (threadIdx.x + 256 * s) * 8;
        if (c < D) {
            float wv[8], y[8];
            unpack8(wraw[s], wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = wv[j] * bfr(v[s][j] * r);
            *reinterpret_cast<u32x4_t*>(orow + c) = pack8(y);
        }
    }
}

// slabs: fp32 [n_slabs][rows][D] (dense rows); x: residual stream, updated in place; w == null: no norm output
int launch_rmsnorm_reduce(const float* slabs, long slab_stride, int n_slabs, bf16_t* x, long ldx, const bf16_t* w, bf16_t* out, long ldo,
                          int rows, int D, float eps, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (D % 8 != 0 || D > 8192 || ldx % 8 != 0 || ldo % 8 != 0 || n_slabs < 1) return ISST_ERR_ARG;
    // (ISST_RMS_LF_ROWS=n: up to n rows, 0: the stepwise kernel everywhere.  Default: every row count with <= 4 slices -- 1408 rows, the prefill of 64 streams:
    //  16.3 -> 15.0 us per launch, profiles/r06/reduce_kernels_many_rows_ab.txt --, up to 256 rows with 5..8 slices: 150 registers per wave, measured there only)
    static const int lf_env = [] { const char* e = getenv("ISST_RMS_LF_ROWS"); return e ? atoi(e) : -1; }();
    const int lf_rows = g_rms_lf_rows >= 0 ? g_rms_lf_rows : lf_env >= 0 ? lf_env : (n_slabs <= 4 ? 1 << 30 : 256);
    if (rows <= lf_rows && n_slabs <= 8 && D <= 4096) {
        if (D <= 2048 && n_slabs <= 4)
            hipLaunchKernelGGL((rmsnorm_reduce_lf_kernel<1, 4>), dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
        else if (D <= 2048)
            hipLaunchKernelGGL((rmsnorm_reduce_lf_kernel<1, 8>), dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
        else if (n_slabs <= 4)
            hipLaunchKernelGGL((rmsnorm_reduce_lf_kernel<2, 4>), dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
        else
            hipLaunchKernelGGL((rmsnorm_reduce_lf_kernel<2, 8>), dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
        return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
    }
    if (D <= 2048)
        hipLaunchKernelGGL(rmsnorm_reduce_kernel<1>, dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
    else if (D <= 4096)
        hipLaunchKernelGGL(rmsnorm_reduce_kernel<2>, dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
    else
        hipLaunchKernelGGL(rmsnorm_reduce_kernel<4>, dim3(rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, x, ldx, w, out, ldo, D, eps);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// plain split-K epilogue: out = bf16(sum of the fp32 slabs), slice order (deterministic).  For the q/k/v projection at 129..1024
// rows, whose consumer (attention) reads bf16 rows; o_proj / down_proj use rmsnorm_reduce_kernel instead.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int n_slabs, bf16_t* __restrict__ out,
                                                          long ldo, int N) {
    const long row = blockIdx.y;
    const int c = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (c >= N) return;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < n_slabs; ++k) {  // (768+ small workgroups: the loads-first form of slab_sum8 changes nothing here, 5.1 us either way)
        const float* sp = slabs + (long)k * slab_stride + row * N + c;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(sp), b = *reinterpret_cast<const f32x4_t*>(sp + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
    *reinterpret_cast<u32x4_t*>(out + row * ldo + c) = pack8(acc);
}

int launch_slab_reduce(const float* slabs, long slab_stride, int n_slabs, bf16_t* out, long ldo, int rows, int N, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (N % 8 != 0 || ldo % 8 != 0 || n_slabs < 1) return ISST_ERR_ARG;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((N / 8 + 255) / 256, rows), dim3(256), 0, s, slabs, slab_stride, n_slabs, out, ldo, N);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// embedding gather + speech splice (reference model/llm.py:86-113): row r takes the speech feature row
// src_row[r] >= 0, else the embedding of token ids[r].  One wave per row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_splice_kernel(const int* __restrict__ ids, const int* __restrict__ speech_row,
                                                           const bf16_t* __restrict__ table, const bf16_t* __restrict__ speech,
                                                           bf16_t* __restrict__ out, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int sr = speech_row ? speech_row[r] : -1;
    const bf16_t* src = sr >= 0 ? speech + (long)sr * D : table + (long)ids[r] * D;
    bf16_t* o = out + (long)r * D;
    for (int c = lane * 8; c < D; c += 512) *reinterpret_cast<u32x4_t*>(o + c) = *reinterpret_cast<const u32x4_t*>(src + c);
}

int launch_embed_splice(const int* ids, const int* speech_row, const bf16_t* table, const bf16_t* speech, bf16_t* out,
                        int rows, int D, hipStream_t s) {
    if (rows <= 0) return ISST_OK;
    if (D % 8 != 0) return ISST_ERR_ARG;
    hipLaunchKernelGGL(embed_splice_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, ids, speech_row, table, speech, out, rows, D);
    return hipGetLastError() == hipSuccess ? ISST_OK : ISST_ERR_HIP;
}
