// Probe (measurement tool, not product code): what does it cost just to READ what a narrow projection at 64 rows has to read -- every workgroup its slice of the
// weights (HBM, once) and its K slice of the 64 activation rows (the same 512 KB - 1.8 MB for everybody: L2 / Infinity Cache) -- at a given grid shape, with no
// MFMA, no LDS staging, no reduction and one store per wave?  VERDICT r04 next #2: "a bare A+W reader at the winning grid shape within 10 % of today's kernels
// ... and the item closed for good".  Shapes: q/k/v (N 6144, K 4096), o_proj (N 4096, K 4096), down_proj (N 4096, K 14336); grid = (N / cols) x slices
// workgroups of 256 threads; every thread keeps 4 16-byte loads of each stream in flight.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o aw_read_probe aw_read_probe.hip && ./aw_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// W: [N][K] bf16 as the packed layout stores it per 16-column tile: a workgroup's (cols x K-slice) block is `cols / 16` runs of (kslice * 32) bytes ... for a
// reader only the byte count and the contiguity of each run matter: run c of the block = W + ((col0 / 16 + c) * K + k0) * 32 bytes, length kslice * 32 bytes.
// A: [64][K] bf16 row-major; the block reads rows 0..63, bytes [k0 * 2, (k0 + kslice) * 2) of each (128-byte lines, as the staging loads do).
__global__ __launch_bounds__(256) void aw_read_kernel(const unsigned char* __restrict__ W, const unsigned char* __restrict__ A, unsigned* __restrict__ out, int K, int cols,
                                                      int kslice, int read_a, int read_w) {
    const int cb = blockIdx.x, sl = blockIdx.y, tid = threadIdx.x;
    const int k0 = sl * kslice;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    if (read_w) {
        const long run_bytes = (long)kslice * 32;  // 16 columns x kslice x 2 B
        for (int c = 0; c < cols / 16; ++c) {
            const unsigned char* base = W + ((long)(cb * (cols / 16) + c) * K + k0) * 32;
            for (long o = (long)tid * 16; o < run_bytes; o += 256 * 16 * 4) {
                u32x4_t q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long oo = o + (long)u * 256 * 16;
                    q[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(base + (oo < run_bytes ? oo : 0)));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc ^= q[u];
            }
        }
    }
    if (read_a) {
        const long row_bytes = (long)kslice * 2;
        const long total = 64 * row_bytes;
        for (long o = (long)tid * 16; o < total; o += 256 * 16 * 4) {
            u32x4_t q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                long oo = o + (long)u * 256 * 16;
                oo = oo < total ? oo : 0;
                const long row = oo / row_bytes, col = oo % row_bytes;
                q[u] = *reinterpret_cast<const u32x4_t*>(A + row * (long)K * 2 + (long)k0 * 2 + col);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= q[u];
        }
    }
    if ((tid & 63) == 0) out[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (tid >> 6)] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
    struct Shape { const char* name; int N, K; };
    const Shape shapes[] = {{"q/k/v", 6144, 4096}, {"o_proj", 4096, 4096}, {"down", 4096, 14336}};
    struct Grid { int cols, slices; };
    const Grid grids[] = {{32, 1}, {32, 2}, {64, 2}, {64, 4}, {128, 4}, {128, 8}, {128, 16}, {256, 8}, {256, 16}};
    hipStream_t st;
    CK(hipStreamCreate(&st));
    unsigned* out;
    CK(hipMalloc(&out, 1 << 20));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (const Shape& s : shapes) {
        const size_t wbytes = (size_t)s.N * s.K * 2;
        const int copies = (int)(600e6 / wbytes) + 2;  // rotate through more than the 256 MB Infinity Cache
        unsigned char* W;
        unsigned char* A;
        CK(hipMalloc(&W, wbytes * copies));
        CK(hipMalloc(&A, (size_t)64 * s.K * 2));
        CK(hipMemset(W, 1, wbytes * copies));
        CK(hipMemset(A, 2, (size_t)64 * s.K * 2));
        printf("%-7s N %5d K %5d  weights %6.1f MB (%.1f us at 6.6 TB/s), A %4.0f KB\n", s.name, s.N, s.K, wbytes / 1e6, wbytes / 6.6e6, 64.0 * s.K * 2 / 1024);
        for (const Grid& g : grids) {
            if (s.K % (g.slices * 64) || s.N % g.cols) continue;
            const int kslice = s.K / g.slices;
            dim3 grid(s.N / g.cols, g.slices);
            const int wgs = grid.x * grid.y;
            if (wgs < 96 || wgs > 1024) continue;
            float t[3];
            for (int mode = 0; mode < 3; ++mode) {  // 0: A + W, 1: W only, 2: A only
                const int ra = mode != 1, rw = mode != 2;
                for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(aw_read_kernel, grid, dim3(256), 0, st, W + (size_t)(i % copies) * wbytes, A, out, s.K, g.cols, kslice, ra, rw);
                CK(hipEventRecord(e0, st));
                const int n = 64;
                for (int i = 0; i < n; ++i) hipLaunchKernelGGL(aw_read_kernel, grid, dim3(256), 0, st, W + (size_t)(i % copies) * wbytes, A, out, s.K, g.cols, kslice, ra, rw);
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t[mode] = ms * 1000.f / n;
            }
            const double a_traffic = (double)wgs * 64 * kslice * 2;
            printf("    %3d columns x %2d K slices = %4d workgroups: A + W %6.2f us   W alone %6.2f   A alone %6.2f   (A through L2: %6.1f MB = %.1f x the weights)\n", g.cols, g.slices, wgs,
                   t[0], t[1], t[2], a_traffic / 1e6, a_traffic / wbytes);
        }
        CK(hipFree(W));
        CK(hipFree(A));
    }
    return 0;
}
