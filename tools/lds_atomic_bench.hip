// Microbenchmark: LDS atomic-add throughput on gfx950 (what bounds k_hist).
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o tools/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: ds_add_u32 x1, 1: ds_add_u32 x2 (g,h separate), 2: ds_add_u64 x1 (packed), 3: ds_add_u64 x2
template <int MODE>
__global__ __launch_bounds__(256) void k(int iters, int nbins, int rep_shift, unsigned long long* sink) {
    __shared__ unsigned long long h64[8192];
    unsigned int* h32 = reinterpret_cast<unsigned int*>(h64);
    for (int i = threadIdx.x; i < 8192; i += 256) h64[i] = 0;
    __syncthreads();
    unsigned int x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    const int lane = threadIdx.x & 63;
    const int rmask = (1 << rep_shift) - 1;
    for (int i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        int bin = nbins < 0 ? lane : (int)((x >> 10) % (unsigned)nbins);   // nbins<0: perfectly conflict free
        int slot = (bin << rep_shift) + (lane & rmask);
        if (MODE == 0) atomicAdd(&h32[slot], x & 0xFFFu);
        if (MODE == 1) { atomicAdd(&h32[slot * 2], x & 0xFFFu); atomicAdd(&h32[slot * 2 + 1], 1u); }
        if (MODE == 2) atomicAdd(&h64[slot], (unsigned long long)(x & 0xFFFu) << 32 | 1ull);
        if (MODE == 3) { atomicAdd(&h64[slot * 2], (unsigned long long)(x & 0xFFFu)); atomicAdd(&h64[slot * 2 + 1], 1ull); }
    }
    __syncthreads();
    unsigned long long s = 0;
    for (int i = threadIdx.x; i < 8192; i += 256) s += h64[i];
    if (s == 0xdeadbeefull) sink[0] = s;
}

template <int MODE>
int run(const char* name, int nbins, int rep_shift, unsigned long long* sink) {
    const int blocks = 256 * 8, iters = 4096;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, 64, nbins, rep_shift, sink);
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, iters, nbins, rep_shift, sink);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double upd = (double)blocks * 256 * iters;   // histogram updates (a (g,h) pair counts once)
    printf("%-22s bins=%4d rep=%2d : %8.3f ms  %7.2f G updates/s  (%.2f updates/clk/CU @2.4GHz)\n", name, nbins, 1 << rep_shift, ms,
           upd / ms * 1e-6, upd / (ms * 1e-3) / 256.0 / 2.4e9);
    return 0;
}

int main() {
    unsigned long long* sink; CK(hipMalloc(&sink, 8));
    int cfg[][2] = {{-1, 0}, {2, 0}, {2, 5}, {8, 0}, {8, 5}, {16, 4}, {64, 0}, {64, 2}, {255, 0}};
    for (auto& c : cfg) {
        run<0>("u32 x1", c[0], c[1], sink);
        run<1>("u32 x2 (g,h)", c[0], c[1], sink);
        run<2>("u64 x1 (packed g|h)", c[0], c[1], sink);
        run<3>("u64 x2 (g,h)", c[0], c[1], sink);
    }
    return 0;
}
