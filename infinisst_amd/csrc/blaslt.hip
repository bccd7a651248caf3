// Plain library GEMM for the many-row prefill projections:  C[M][N] (bf16) = A[M][K] (bf16) @ W[N][K]^T (bf16, row-major), fp32 accumulation.
//
// Why a library call on this path: at 1408 rows (64 streams x 22 prompt rows) the four Llama projections are plain dense GEMMs, and hipBLASLt runs them at
// 0.95-1.23 PFLOP/s against 0.63-0.89 for gemm_tiled.hip's 128 x 128 kernel on the same shapes and data (profiles/dense_vs_library_probe.py); the epilogues
// the packed-weight kernel fuses (SwiGLU, residual + RMSNorm) become two bandwidth-bound passes (rowops.hip) and still leave ~4 % of the 64-stream step.
// Everything at <= 128 rows -- the weight-streaming regime this library exists for -- stays on the hand-written kernels, and so does the dense path when
// ISST_BLASLT=0 or the row-major weight copies were not made (engines whose row capacity never reaches this path).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <tuple>
#include "common.h"
#include "kernels.h"

namespace {
struct LtPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t lw = nullptr, la = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
};
hipblasLtHandle_t g_lt = nullptr;
void* g_ws = nullptr;
constexpr size_t LT_WS_BYTES = 64u << 20;
std::map<std::tuple<int, int, int, long, long, int>, LtPlan> g_plans;

bool lt_init() {
    if (g_lt) return true;
    if (hipblasLtCreate(&g_lt) != HIPBLAS_STATUS_SUCCESS) { g_lt = nullptr; return false; }
    if (hipMalloc(&g_ws, LT_WS_BYTES) != hipSuccess) { g_ws = nullptr; return false; }
    return true;
}
}  // namespace

bool gemm_lt_available() { return lt_init(); }

// bias != null: C = bf16(acc + bias[n]) (the library's bias epilogue: added in fp32 before the one rounding, as the packed kernels' EPI_BIAS)
int launch_gemm_lt(const bf16_t* A, long lda, const bf16_t* W, bf16_t* C, long ldc, int M, int N, int K, hipStream_t stream, const bf16_t* bias) {
    if (M <= 0) return ISST_OK;
    if (!lt_init()) return ISST_ERR_HIP;
    const auto key = std::make_tuple(M, N, K, lda, ldc, bias ? 1 : 0);
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        // column-major view of the row-major problem: C^T (N x M, ld = ldc) = W (K x N, ld = K)^T . A^T (K x M, ld = lda)
        LtPlan p;
        if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return ISST_ERR_HIP;
        const hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
        if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof opT) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof opN) != HIPBLAS_STATUS_SUCCESS)
            return ISST_ERR_HIP;
        if (bias) {
            const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
            const hipDataType bt = HIP_R_16BF;
            if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof epi) != HIPBLAS_STATUS_SUCCESS ||
                hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof bt) != HIPBLAS_STATUS_SUCCESS ||
                hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias) != HIPBLAS_STATUS_SUCCESS)
                return ISST_ERR_HIP;
        }
        if (hipblasLtMatrixLayoutCreate(&p.lw, HIP_R_16BF, K, N, K) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, K, M, lda) != HIPBLAS_STATUS_SUCCESS ||
            hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_16BF, N, M, ldc) != HIPBLAS_STATUS_SUCCESS)
            return ISST_ERR_HIP;
        hipblasLtMatmulPreference_t pref = nullptr;
        if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return ISST_ERR_HIP;
        const uint64_t ws = LT_WS_BYTES;
        (void)hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof ws);
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        const hipblasStatus_t hs = hipblasLtMatmulAlgoGetHeuristic(g_lt, p.desc, p.lw, p.la, p.lc, p.lc, pref, 1, res, &found);
        (void)hipblasLtMatmulPreferenceDestroy(pref);
        if (hs != HIPBLAS_STATUS_SUCCESS || found < 1) return ISST_ERR_HIP;
        p.algo = res[0].algo;
        p.ws = res[0].workspaceSize;
        it = g_plans.emplace(key, p).first;
    }
    const LtPlan& p = it->second;
    if (bias && hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof bias) != HIPBLAS_STATUS_SUCCESS) return ISST_ERR_HIP;  // (per call: plans are shared by layers)
    const float alpha = 1.f, beta = 0.f;
    const hipblasStatus_t st = hipblasLtMatmul(g_lt, p.desc, &alpha, W, p.lw, A, p.la, &beta, C, p.lc, C, p.lc, &p.algo, g_ws, p.ws <= LT_WS_BYTES ? p.ws : 0, stream);
    return st == HIPBLAS_STATUS_SUCCESS ? ISST_OK : ISST_ERR_HIP;
}
