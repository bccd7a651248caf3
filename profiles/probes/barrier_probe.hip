// Probe (measurement tool, not product code): cost of a grid-wide barrier inside one persistent launch on MI355X versus
// the cost of a kernel boundary, to decide whether a persistent per-pass decode kernel can beat 6 launches per layer.
//   hipcc --offload-arch=gfx950 -O3 -o barrier_probe barrier_probe.hip && ./barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, int sleep) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (sleep) __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { ok = false; break; }  // never hang the box
        }
    }
    __syncthreads();
    return ok;
}

// relaxed polling, one acquire fence at the end
__device__ __forceinline__ bool grid_barrier_relaxed(unsigned* ctr, unsigned target) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > 20000000) { ok = false; break; }
        }
        __atomic_thread_fence(__ATOMIC_ACQUIRE);  // agent scope by default for hip
    }
    __syncthreads();
    return ok;
}

template <int MODE>
__global__ __launch_bounds__(256) void barrier_only(unsigned* ctr, int iters, int* err) {
    for (int it = 0; it < iters; ++it) {
        const unsigned target = (unsigned)(it + 1) * gridDim.x;
        bool ok = MODE == 2 ? grid_barrier_relaxed(ctr, target) : grid_barrier(ctr, target, MODE);
        if (!ok) { if (threadIdx.x == 0) *err = 1; return; }
    }
}

// every workgroup writes 4 KB, barrier, reads the 4 KB of another workgroup (on another XCD) and checks it
__global__ __launch_bounds__(256) void barrier_data(unsigned* ctr, int iters, uint4* buf, int* err, int* bad) {
    const int G = gridDim.x;
    int nbad = 0;
    for (int it = 0; it < iters; ++it) {
        uint4* cur = buf + (size_t)(it & 1) * G * 256;
        const unsigned v = (unsigned)it * 7919u + blockIdx.x;
        cur[(size_t)blockIdx.x * 256 + threadIdx.x] = make_uint4(v, v + 1, v + 2, threadIdx.x);
        if (!grid_barrier_relaxed(ctr, (unsigned)(it + 1) * G)) { if (threadIdx.x == 0) *err = 1; return; }
        const int other = (blockIdx.x + 3) % G;
        const uint4 r = cur[(size_t)other * 256 + threadIdx.x];
        const unsigned e = (unsigned)it * 7919u + other;
        if (r.x != e || r.y != e + 1 || r.z != e + 2 || r.w != threadIdx.x) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ __launch_bounds__(256) void tiny(uint4* buf) {
    buf[(size_t)blockIdx.x * 256 + threadIdx.x] = make_uint4(blockIdx.x, 1, 2, 3);
}

int main() {
    unsigned* ctr; int* err; int* bad; uint4* buf;
    CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&buf, sizeof(uint4) * 2 * 2048 * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int iters = 2000;
    for (int G : {256, 512, 1024, 2048}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            int herr = 0, hbad = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(ctr, 0, 4, st)); CK(hipMemsetAsync(err, 0, 4, st)); CK(hipMemsetAsync(bad, 0, 4, st));
                CK(hipEventRecord(e0, st));
                if (mode == 0) hipLaunchKernelGGL(barrier_only<0>, dim3(G), dim3(256), 0, st, ctr, iters, err);
                if (mode == 1) hipLaunchKernelGGL(barrier_only<1>, dim3(G), dim3(256), 0, st, ctr, iters, err);
                if (mode == 2) hipLaunchKernelGGL(barrier_only<2>, dim3(G), dim3(256), 0, st, ctr, iters, err);
                if (mode == 3) hipLaunchKernelGGL(barrier_data, dim3(G), dim3(256), 0, st, ctr, iters, buf, err, bad);
                CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
            }
            const char* names[] = {"acq-poll", "acq-poll+sleep", "relaxed-poll", "relaxed+4KB data/wg"};
            printf("G=%4d %-22s %.3f us/barrier  timeout=%d bad=%d\n", G, names[mode], best * 1000.f / iters, herr, hbad);
        }
    }
    for (int G : {256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny, dim3(G), dim3(256), 0, st, buf);
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("G=%4d back-to-back tiny launches: %.3f us/launch\n", G, ms * 1000.f / iters);
        }
    }
    // same, replayed from a graph
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(1024), dim3(256), 0, st, buf);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("G=1024 graph of 200 tiny launches: %.3f us/launch\n", ms * 1000.f / 2000);
        }
    }
    return 0;
}
