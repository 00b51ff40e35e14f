// Microbenchmark: does the cost of a 64-bit LDS atomic wave-instruction scale with the number of active lanes?
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_lanes.hip -o tools/lds_atomic_lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k(int iters, int nbins, int rep_shift, int pct, int contiguous, unsigned long long* sink) {
    extern __shared__ unsigned long long h64[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) h64[i] = 0;
    __syncthreads();
    unsigned int x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    const int rmask = (1 << rep_shift) - 1;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        int bin = (int)((x >> 10) % (unsigned)nbins);
        int slot = (bin << rep_shift) + (lane & rmask);
        bool on = contiguous ? (lane * 100 < pct * 64) : ((int)((x >> 3) % 100u) < pct);
        if (on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(&h64[(slot + j * 1031) & 16383], (unsigned long long)(x & 0xFFFu) << 32 | 1ull);
        }
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s += h64[i];
    if (s == 0xdeadbeefull) sink[0] = s;
}

int main() {
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int blocks = 256, iters = 2048;
    int pcts[] = {100, 75, 50, 25, 12};
    for (int threads : {256, 1024}) for (int contiguous = 0; contiguous < 2; ++contiguous) for (int pct : pcts) {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 131072, 0, 32, 64, 2, pct, contiguous, sink);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 131072, 0, iters, 64, 2, pct, contiguous, sink);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        double instr = (double)blocks * (threads / 64) * iters * 8;           // wave-instructions issued
        double upd = (double)blocks * threads * iters * 8 * pct / 100.0;    // lane updates
        printf("threads=%4d %s active=%3d%% : %8.3f ms  %6.2f clk/wave-instr/CU  %6.2f updates/clk/CU\n", threads, contiguous ? "contig" : "random", pct, ms,
               ms * 1e-3 * 2.4e9 / (instr / 256.0), upd / (ms * 1e-3) / 256.0 / 2.4e9);
    }
    return 0;
}
