// Shared device helpers for the gfx950 (CDNA4) kernels of the InfiniSST hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; arithmetic is always done in fp32

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define WAVE 64

__device__ __forceinline__ float bf2f(bf16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
// round-to-nearest-even; hipcc emits v_cvt_pk_bf16_f32 (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(bf16_t, h);
}
// value after one rounding to bf16 (the reference rounds after every torch op)
__device__ __forceinline__ float bfr(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// two floats rounded (nearest even, NaN kept) and packed by ONE v_cvt_pk_bf16_f32 (two scalar casts + shift + or compile to three or four instructions for the same bits)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const u32x4_t& v, float* f) {
    f[0] = lo_bf(v.x); f[1] = hi_bf(v.x); f[2] = lo_bf(v.y); f[3] = hi_bf(v.y);
    f[4] = lo_bf(v.z); f[5] = hi_bf(v.z); f[6] = lo_bf(v.w); f[7] = hi_bf(v.w);
}
__device__ __forceinline__ u32x4_t pack8(const float* f) {
    u32x4_t v;
    v.x = pack_bf(f[0], f[1]); v.y = pack_bf(f[2], f[3]); v.z = pack_bf(f[4], f[5]); v.w = pack_bf(f[6], f[7]);
    return v;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }


// Wave-wide reductions on DPP (row rotations / broadcasts inside the VALU: ~8 cycles a step) with the result in EVERY lane.  __shfl_xor compiles to
// ds_bpermute_b32, an LDS-crossbar round trip of ~100 cycles per step: a k-round argmax over (value, index) pairs -- 12-18 dependent bpermutes per round --
// made the beam search's small top-k kernels 11-25 us of pure latency (profiles/r05/beam_tail_kernels.txt).  Sequence as rocPRIM's warp_reduce_dpp:
// quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8, row_bcast:15, row_bcast:31 -> lane 63 holds the reduction; v_readlane broadcasts it.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "the DPP reductions below use row_bcast:15 / row_bcast:31 (wave64, gfx9 family only): this library is written for gfx950"
#endif
static_assert(WAVE == 64, "row_bcast:31 and v_readlane 63 assume 64-lane waves");
template <int CTRL>
__device__ __forceinline__ int dpp_take(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }  // (lanes without a source keep their own value)
__device__ __forceinline__ float wave_max_all(float v) {
    v = fmaxf(v, __int_as_float(dpp_take<0xb1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_take<0x4e>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_take<0x124>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_take<0x128>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_take<0x142>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_take<0x143>(__float_as_int(v))));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_all(int v) {
    v = min(v, dpp_take<0xb1>(v));
    v = min(v, dpp_take<0x4e>(v));
    v = min(v, dpp_take<0x124>(v));
    v = min(v, dpp_take<0x128>(v));
    v = min(v, dpp_take<0x142>(v));
    v = min(v, dpp_take<0x143>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
// wave_sum / wave_max of every kernel of the library (norm statistics, softmax, split-KV merge): round 5 moved them from the xor butterfly over ds_bpermute
// (6 dependent LDS-crossbar round trips, ~0.3 us on a kernel's critical path) onto the same DPP tree.  The sum's association order changed with it -- lanes
// (0,1), quads, rows of 16 by rotation, then rows -- once, for every consumer alike: each path still adds in ONE fixed order, the bit-identity tests between
// launch forms compare like with like.
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(dpp_take<0xb1>(__float_as_int(v)));
    v += __int_as_float(dpp_take<0x4e>(__float_as_int(v)));
    v += __int_as_float(dpp_take<0x124>(__float_as_int(v)));
    v += __int_as_float(dpp_take<0x128>(__float_as_int(v)));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3 (the other rows add 0)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) { return wave_max_all(v); }
// the wave's best (value, index) pair under (value descending, index ascending), in every lane (NaN-free values)
__device__ __forceinline__ void wave_argmax_all(float& v, int& i) {
    const float m = wave_max_all(v);
    i = wave_min_all(v == m ? i : 0x7fffffff);
    v = m;
}

// ---- status codes of the C ABI (include/infinisst_hip.h) ----
#define ISST_OK 0
#define ISST_ERR_ARG (-1)
#define ISST_ERR_HIP (-2)
#define ISST_ERR_STATE (-3)
#define ISST_ERR_NOMEM (-4)
#define ISST_ERR_NOTFOUND (-5)

// ---- epilogues of the packed-weight GEMM ----
enum GemmEpi {
    EPI_NONE = 0,       // out = bf16(acc)
    EPI_BIAS = 1,       // out = bf16(acc + bias)
    EPI_BIAS_GELU = 2,  // out = bf16(gelu(bf16(acc + bias)))
    EPI_RES = 3,        // out = bf16(res + bf16(acc))
    EPI_BIAS_RES = 4,   // out = bf16(res + bf16(acc + bias))
    EPI_SWIGLU = 5,     // tile pairs (gate, up): out = bf16(bf16(silu(bf16 g)) * bf16 u)
    EPI_F32 = 6,        // out(fp32) = bf16-rounded acc
    EPI_PARTIAL = 7,    // split-K slab: out(fp32)[slice][M][N] = raw partial sum (gemm_mid.hip; reduced by rmsnorm_reduce)
    EPI_SWIGLU8 = 8,    // SwiGLU on SELF-PAIRED tiles (gemm.hip, one-row passes): tile t = gate rows 8t..8t+7 | up rows 8t..8t+7 (launch_pack_weight_half): out[8t + c] as EPI_SWIGLU
};

// launchers (defined in the kernel .hip files; all asynchronous on `stream`)
struct GemmArgs {
    const bf16_t* A; long lda; long a_batch;      // A[batch][M][K] with row stride lda (elements)
    const bf16_t* Wp;                              // packed weight [N/16][K/32][64][8]
    const bf16_t* bias;                            // [N] or null
    const bf16_t* res; long ldres; long res_batch; // residual, may alias out
    void* out; long ldo; long out_batch;           // bf16 (or fp32 for EPI_F32)
    int M, N, K, batch, epi;                       // N = number of packed rows (multiple of 16)
    int n_valid;                                   // output columns actually stored (<= N, or N/2 for SWIGLU)
    const bf16_t* norm_w;                          // non-null: A rows are RMS-normalised on load (HF LlamaRMSNorm) with this weight
    float norm_eps;
    int ksplit;                                    // EPI_PARTIAL: K slices over workgroups (slab stride = out_batch elements)
    // non-null (M <= ATTN_MERGE_MAX_ROWS, K = heads * 128, EPI_RES): A is NOT read; the rows are the Llama attention output merged on load from
    // the split-KV partials of llm_attn_partial_kernel, [M][K / 128][attn_splits][ATTN_SLAB] fp32 = (unnormalised O[128], running max, sum) per split
    const float* attn_partial;
    int attn_splits;
    int tune;                                      // experiment switches (gemm_set_tuning 700000 + bits), 0 in production
    // 13..64 rows (gemm_mid.hip), the residual + RMSNorm between two projections WITHOUT a launch of its own:
    //  producer (EPI_PARTIAL, tickets != null): the K-slice workgroup of a column block that arrives last sums the slabs (slice order), writes
    //    x = bf16(x + bf16(sum)) in place (x = res, ldres) and, per row and 32-column pair, the sum of squares of the new x -> ssq[row][N / 32];
    //  consumer (norm_w != null, ssq != null): A = x; rows are RMS-normalised on the way into LDS with 1/rms from the ssq partials (fixed order).
    float* ssq; int ssq_n;                         // [M][ssq_n] fp32, ssq_n = x columns / 32
    int* tickets;                                  // [column blocks] arrival counters, zero between launches (the last arriver re-arms its own)
    int reduce_plain;                              // producer: no residual -- res[row][col] = bf16(sum of the slabs) (q/k/v in K slices), ssq unused
};
#define ATTN_MERGE_MAX_ROWS 2
#define ATTN_MERGE_MAX_SPLITS 32
#define ATTN_SLAB 132  // floats per (row, head, split) slab: O[0..127], max, sum, 2 x pad -- 528 B, so that a slab is 33 whole 16-byte stores
// Merge of the split-KV attention partials of one (row, head) for the two output dims a lane owns: the ONE definition of this arithmetic
// (llm_attn_combine_kernel and the o_proj GEMV's merge-on-load prologue both call it, so a row gives the same bits through either).
// src: [n_splits][ATTN_SLAB] fp32 (O[0..127], max, sum, 2 x pad); every load is issued before the first use.  Returns bf16(O / L) of dims 2 lane, 2 lane + 1, packed.
// (split into a load half and a math half so that a caller can put several heads' loads -- and other loads -- in flight before the first use;
//  MAXS >= n_splits is the unroll bound: slabs past n_splits fall outside the descriptor's extent and read zeros at no traffic)
template <int MAXS>
struct AttnMergeLoads {
    u32x2_t st;        // lane s: (max, sum) of split s
    u32x2_t os[MAXS];  // the lane's two dims of every split's O
};
// AUX: cache policy of the loads (0 = default; 16 = sc1: bypass this CU's L1 -- slabs another workgroup of the SAME launch stored write-through)
template <int MAXS, int AUX = 0>
__device__ __forceinline__ void attn_merge_issue(const float* __restrict__ src, int n_splits, int lane, AttnMergeLoads<MAXS>& ld) {
    // one descriptor over exactly the n_splits slabs: a uniform (SGPR) slab offset plus one per-lane offset, no per-load address registers
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_splits * (ATTN_SLAB * 4), 0x00020000);
    ld.st = __builtin_amdgcn_raw_buffer_load_b64(rs, (unsigned)lane * (ATTN_SLAB * 4u) + 512u, 0, AUX);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) ld.os[s] = __builtin_amdgcn_raw_buffer_load_b64(rs, 8u * (unsigned)lane, s * (ATTN_SLAB * 4), AUX);
}
template <int MAXS>
__device__ __forceinline__ uint32_t attn_merge_finish(const AttnMergeLoads<MAXS>& ld, int n_splits, int lane) {
#pragma clang fp contract(off)  // (one arithmetic in every kernel this is inlined into: no compiler-chosen fma)
    const bool mine = lane < n_splits;
    const float m = mine ? __uint_as_float(ld.st.x) : -INFINITY;
    const float M = wave_max(m);
    const float w = (m == -INFINITY) ? 0.f : expf(m - M);  // -inf: split without a live key for this row
    const float L = wave_sum(mine ? __uint_as_float(ld.st.y) * w : 0.f);
    float O0 = 0.f, O1 = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const float ws = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), s));
        O0 = __builtin_fmaf(__uint_as_float(ld.os[s].x), ws, O0);
        O1 = __builtin_fmaf(__uint_as_float(ld.os[s].y), ws, O1);
    }
    return pack_bf(O0 / L, O1 / L);
}
__device__ __forceinline__ uint32_t attn_merge_pair(const float* __restrict__ src, int n_splits, int lane) {
    AttnMergeLoads<ATTN_MERGE_MAX_SPLITS> ld;
    attn_merge_issue<ATTN_MERGE_MAX_SPLITS>(src, n_splits, lane, ld);
    return attn_merge_finish<ATTN_MERGE_MAX_SPLITS>(ld, n_splits, lane);
}
// rows above which the 17..64-row machinery (gemm_mid.hip, split-K slabs + reducing RMSNorm) replaces the skinny kernel.  At 13..16 rows
// every 16-column workgroup of the skinny kernel still pulls all of A through L2 -- as many bytes as the weight stream -- and the
// residual + RMSNorm is a launch of its own (profiles/mid16_probe.py, 16 rows: q/k/v 15.6 -> 13.3 us, gate/up 43.7 -> 41.0,
// o_proj + norm 15.4 -> 12.8, down + norm 32.2 -> 25.5).  End to end, same box: 16 streams 56.2 -> 53.6 ms per chunk, 15: 57.8 -> 55.9,
// 14: 56.8 -> 55.3, 13: 55.6 -> 55.5; from 9 rows on it loses (12 streams 52.6 -> 53.3, 9: 47.6 -> 49.1), hence 12
#ifndef ISST_MID_MIN_ROWS
#define ISST_MID_MIN_ROWS 12
#endif
#define GEMM_FUSED_NORM_MAX_M 8  // rows for which the GEMM stages (and optionally RMS-normalises) A in LDS
int launch_gemm(const GemmArgs& g, hipStream_t stream);
void gemm_set_tuning(int waves_per_block, int ntiles_per_block);  // 0 = heuristic; profiling aid
bool gemm_tiled_supported(const GemmArgs& g);                      // gemm_tiled.hip: dense shapes (M > 64 or batched)
int launch_gemm_tiled(const GemmArgs& g, hipStream_t stream);
void gemm_tiled_set_raster(int on);                                // profiling aid: 0 = plain (column block, row block) grid
bool gemm_dense_supported(const GemmArgs& g);                     // gemm_dense.hip: 256 x 256 tiles, 8-wave ping-pong, LDS-DMA (many-row prefill / encoder)
bool gemm_dense_preferred(const GemmArgs& g);
bool gemm_dense_would_run(int M, int N, int K);                   // ... as a function of the shape alone (the engine picks its K slices by it)
int launch_gemm_dense(const GemmArgs& g, hipStream_t stream);
double gemm_dense_pick_mix(int M, int N, int ks, int* n_full, int* n_half);  // the launch's tile mix (row blocks of 256, then of 128) and its modelled length in 256-row tiles over all of K
void gemm_dense_set(int mode);                                     // profiling aid: 0 never, 1 heuristic, 2 wherever supported
bool gemm_wide_supported(const GemmArgs& g);                       // gemm_wide.hip: 65..256 rows, weights read once (8 waves: 4 consumers with the W ring in registers + 4 LDS-DMA loaders for the A ring)
bool gemm_wide_preferred(const GemmArgs& g);
int launch_gemm_wide(const GemmArgs& g, hipStream_t stream);
bool gemm_wide_enabled();
void gemm_wide_set(int mode, int variant);                         // profiling aid: mode 0 never, 1 heuristic, 2 wherever supported; variant = ring depths
bool gemm_mid_supported(const GemmArgs& g);                        // gemm_mid.hip: 17..64 rows, A staged through LDS
bool gemm_mid_preferred(const GemmArgs& g);                       // ... and long enough a weight stream to pay for the staging
int launch_gemm_mid(const GemmArgs& g, hipStream_t stream);
void gemm_mid_set_tuning(int wn);
void gemm_mid_set_min_rows(int rows);                              // rows above which gemm_mid replaces the skinny kernel (default 16)
int launch_pack_weight(const bf16_t* src, bf16_t* dst, int n_rows, int K, int row_offset_tiles, int tile_stride,
                       int tile_phase, int conv_k, hipStream_t stream);
// rows r of src -> rows (r % 8) + 8 * half of tile r / 8: gate_proj (half 0) and up_proj (half 1) packed this way give tiles that hold BOTH operands of 8 SwiGLU outputs
int launch_pack_weight_half(const bf16_t* src, bf16_t* dst, int n_rows, int K, int half, hipStream_t stream);
