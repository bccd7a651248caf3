// Probe (measurement tool): is the transposed-V access pattern of the decoder attention (per 16-key tile: 8 x 8-byte loads per lane,
// each wave-load touching 16 rows x 32 bytes of a [128 dims][slots] plane) slower to stream than the same bytes read as the K tile is
// (4 x 16-byte loads per lane, 4 KiB contiguous)?   hipcc --offload-arch=gfx950 -O3 -o vt_load_probe vt_load_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// one workgroup (4 waves) per plane (stream, kv head); wave w takes tiles w, w+4, ...; PREFETCH tiles ahead
template <int MODE, int PF, int NW = 4>
__global__ __launch_bounds__(NW * 64) void stream_planes(const unsigned short* __restrict__ pool, int slots, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const unsigned short* plane = pool + (size_t)blockIdx.x * slots * 128;
    const int tiles = slots / 16;
    unsigned acc = 0;
    if (MODE == 0) {  // V^T: plane [128][slots]
        u32x2 ring[PF][8];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) ring[p][nt] = *reinterpret_cast<const u32x2*>(plane + (size_t)(16 * nt + fr) * slots + (wave + NW * p) * 16 + 4 * fq);
        for (int t = wave; t < tiles; t += NW * PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                u32x2 cur[8];
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) cur[nt] = ring[p][nt];
                int tn = t + NW * p + NW * PF;
                tn = tn < tiles ? tn : wave;
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) ring[p][nt] = *reinterpret_cast<const u32x2*>(plane + (size_t)(16 * nt + fr) * slots + tn * 16 + 4 * fq);
#pragma unroll
                for (int nt = 0; nt < 8; ++nt) acc ^= cur[nt].x ^ cur[nt].y;
            }
        }
    } else {  // K-like: plane [slots][128]
        u32x4 ring[PF][4];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < 4; ++s) ring[p][s] = *reinterpret_cast<const u32x4*>(plane + (size_t)((wave + NW * p) * 16 + fr) * 128 + 32 * s + 8 * fq);
        for (int t = wave; t < tiles; t += NW * PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                u32x4 cur[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) cur[s] = ring[p][s];
                int tn = t + NW * p + NW * PF;
                tn = tn < tiles ? tn : wave;
#pragma unroll
                for (int s = 0; s < 4; ++s) ring[p][s] = *reinterpret_cast<const u32x4*>(plane + (size_t)(tn * 16 + fr) * 128 + 32 * s + 8 * fq);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc ^= cur[s].x ^ cur[s].y ^ cur[s].z ^ cur[s].w;
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const int slots = 1088, planes = 512, reps = 30;
    const size_t bytes = (size_t)planes * slots * 128 * 2;
    unsigned short* pool[4];
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&pool[i], bytes)); CK(hipMemset(pool[i], i + 1, bytes)); }
    unsigned* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto kern) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(planes), dim3(256), 0, 0, pool[i % 4], slots, sink);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(planes), dim3(256), 0, 0, pool[i % 4], slots, sink);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %7.1f us per launch  %6.2f TB/s (%.0f MB)\n", name, ms * 1000 / reps, bytes / (ms / reps * 1e-3) / 1e12, bytes / 1e6);
    };
    run("V^T planes [128][slots], 1 ahead", stream_planes<0, 1>);
    run("V^T planes [128][slots], 2 ahead", stream_planes<0, 2>);
    run("row planes [slots][128], 1 ahead", stream_planes<1, 1>);
    run("row planes [slots][128], 2 ahead", stream_planes<1, 2>);
    // one workgroup of 16 waves per plane pair (K + V of one kv head = 2 planes): can 8 workgroups stream one stream's KV fast enough?
    {
        const int small = 16;  // 8 kv heads x (K, V)
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_planes<1, 1, 16>), dim3(small), dim3(1024), 0, 0, pool[i % 4], slots, sink);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_planes<1, 1, 16>), dim3(small), dim3(1024), 0, 0, pool[i % 4] + (size_t)(i % 16) * small * slots * 128, slots, sink);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("16 workgroups x 16 waves, one 278 KB plane each: %.2f us per launch\n", ms * 1000 / reps);
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_planes<1, 1, 4>), dim3(16 * 17), dim3(256), 0, 0, pool[i % 4], 64, sink);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_planes<1, 1, 4>), dim3(16 * 17), dim3(256), 0, 0, pool[i % 4] + (size_t)(i % 16) * small * slots * 128, 64, sink);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("272 workgroups x 4 waves, 64 slots each (today's split): %.2f us per launch\n", ms * 1000 / reps);
    }
    return 0;
}
