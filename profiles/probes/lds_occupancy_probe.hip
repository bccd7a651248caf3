// How many 512-thread workgroups does a CU of the MI355X host at once, by dynamic LDS per workgroup and registers per lane?
// (enc_attention_kernel<3>: 72.75 KB of LDS, 112 VGPRs, 512 threads -- the trace shows exactly ONE workgroup per CU although 2 x 72.75 < 160 KB.)
// Each workgroup spins ~20 us on the wall clock; 1024 workgroups; the number alive at once / 256 CUs is the occupancy.
// build: hipcc -O3 --offload-arch=gfx950 lds_occupancy_probe.hip -o lds_occupancy_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>


template <int REGS>
__global__ __launch_bounds__(512) void spin_kernel(unsigned long long* stamps, float* sink, int spin_ticks) {
    extern __shared__ float lds[];
    float keep[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) keep[i] = threadIdx.x * 0.5f + i;
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = keep[0];
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) keep[i] = keep[i] * 1.0001f + lds[(threadIdx.x + i) & 511];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += keep[i];
    if (s == 12345.678f) sink[0] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

template <int REGS>
static void run(int lds_bytes, unsigned long long* d_st, float* d_sink) {
    const int n = 1024;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) {
        printf("regs %3d lds %6d: set attribute failed\n", REGS, lds_bytes);
        return;
    }
    hipLaunchKernelGGL(spin_kernel<REGS>, dim3(n), dim3(512), lds_bytes, 0, d_st, d_sink, 2000);  // 20 us at 100 MHz
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return; }
    std::vector<unsigned long long> st(2 * n);
    hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < n; ++i) { ev.push_back({st[2 * i], 1}); ev.push_back({st[2 * i + 1], -1}); }
    std::sort(ev.begin(), ev.end());
    int alive = 0, peak = 0;
    for (auto& e : ev) { alive += e.second; peak = std::max(peak, alive); }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(spin_kernel<REGS>));
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin_kernel<REGS>, 512, lds_bytes);
    printf("regs/lane %3d (compiler %3d)  dynamic LDS %6d B: peak %4d workgroups alive = %.2f per CU   (runtime's occupancy answer: %d)\n", REGS, fa.numRegs, lds_bytes, peak,
           peak / 256.0, occ);
}

int main() {
    unsigned long long* d_st;
    float* d_sink;
    hipMalloc(&d_st, 2 * 1024 * 8);
    hipMalloc(&d_sink, 64);
    const int sizes[] = {16 << 10, 32 << 10, 48 << 10, 53 << 10, 60 << 10, 64 << 10, 65 << 10, 72 << 10, 74496, 80 << 10};
    for (int b : sizes) run<32>(b, d_st, d_sink);
    for (int b : sizes) run<96>(b, d_st, d_sink);
    return 0;
}
