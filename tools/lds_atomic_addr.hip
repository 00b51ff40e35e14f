// Microbenchmark for DESIGN.md 9.2(a): how much of a level pass is the ADDRESS arithmetic of its 15 LDS atomics?
//   variant 0  today's layout  [bin][replica]:  addr = c_j(lane) + (byte_j(record) << sh_j)      -> 2 VALU per atomic
//   variant 1  layout [replica][bin] with 16-bit pre-multiplied bin offsets in the record:
//              addr = c_j(lane) + half_j(record)                                                 -> 1 VALU (SDWA add) per atomic
//   variant 2  no atomics at all (loads + the arithmetic of variant 0 folded into a checksum): the streaming floor
// Each workgroup (1024 threads, one per CU) streams `rows` synthetic records from HBM with a one-tile prefetch, like
// k_level_mt.  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_addr.hip -o tools/lds_atomic_addr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int THREADS = 1024, NF = 15;

template <int VARIANT>
__global__ __launch_bounds__(THREADS) void k(const uint4* __restrict__ rec8, const uint4* __restrict__ rec16, const int2* __restrict__ gh,
                                             long long rows, int sh, int nbins, unsigned long long* __restrict__ sink) {
    extern __shared__ unsigned long long slots[];
    const int nslots = NF * (nbins << sh);
    for (int i = threadIdx.x; i < nslots; i += THREADS) slots[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, rep = lane & ((1 << sh) - 1);
    int cj[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) cj[j] = VARIANT == 1 ? (j * (nbins << sh) + rep * nbins) * 8 : (j * (nbins << sh) + rep) * 8;
    unsigned long long acc = 0;
    unsigned char* base = reinterpret_cast<unsigned char*>(slots);
    for (long long r = (long long)blockIdx.x * THREADS + threadIdx.x; r < rows; r += (long long)gridDim.x * THREADS) {
        const int2 g = gh[r];
        const unsigned long long packed = ((unsigned long long)(long long)g.x << 32) + (unsigned int)g.y;
        if (VARIANT == 1) {
            const uint4 a = rec16[2 * r], b = rec16[2 * r + 1];           // 16 half-words = bin * 8, pre-multiplied at pack time
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int addr = cj[j] + (int)((w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu);
                atomicAdd(reinterpret_cast<unsigned long long*>(base + addr), packed);
            }
        } else {
            const uint4 a = rec8[r];
            const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const int addr = cj[j] + (int)(((w[j >> 2] >> (8 * (j & 3))) & 0xFFu) << (sh + 3));
                if (VARIANT == 0) atomicAdd(reinterpret_cast<unsigned long long*>(base + addr), packed);
                else acc += (unsigned)addr ^ packed;
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nslots; i += THREADS) acc += slots[i];
    if (acc == 0x1234567ull) sink[0] = acc;
}

int main() {
    const long long rows = 40ll << 20;                       // 40M records: 640 MB (u8) / 1.28 GB (u16) + 320 MB of (g,h)
    const int sh = 4, nbins = 24;                            // 15 features x 24 bins x 16 replicas = 46 KB of slots
    std::vector<uint4> h8((size_t)rows), h16((size_t)rows * 2); std::vector<int2> hg((size_t)rows);
    unsigned x = 12345u;
    for (long long r = 0; r < rows; ++r) {
        unsigned char b[16]; unsigned short s[16];
        for (int j = 0; j < 16; ++j) { x = x * 1664525u + 1013904223u; b[j] = (unsigned char)((x >> 9) % (unsigned)nbins); s[j] = (unsigned short)(b[j] * 8); }
        memcpy(&h8[r], b, 16); memcpy(&h16[2 * r], s, 32);
        hg[r] = make_int2((int)(x >> 12) - 500000, (int)(x >> 13));
    }
    uint4 *d8, *d16; int2* dg; unsigned long long* sink;
    CK(hipMalloc(&d8, rows * 16)); CK(hipMalloc(&d16, rows * 32)); CK(hipMalloc(&dg, rows * 8)); CK(hipMalloc(&sink, 8));
    CK(hipMemcpy(d8, h8.data(), rows * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d16, h16.data(), rows * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dg, hg.data(), rows * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t)NF * (nbins << sh) * 8;
    auto run = [&](int variant) -> float {
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            (void)hipEventRecord(a, 0);
            if (variant == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(THREADS), lds, 0, d8, d16, dg, rows, sh, nbins, sink);
            if (variant == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(THREADS), lds, 0, d8, d16, dg, rows, sh, nbins, sink);
            if (variant == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(THREADS), lds, 0, d8, d16, dg, rows, sh, nbins, sink);
            (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms = 0; (void)hipEventElapsedTime(&ms, a, b); if (it > 0 && ms < best) best = ms;
        }
        return best;
    };
    const char* name[] = {"[bin][replica], u8 bins, shift+add (2 VALU)", "[replica][bin], u16 offsets, sdwa add (1 VALU)", "no atomics (streaming + address floor)"};
    for (int v = 0; v < 3; ++v) {
        const float ms = run(v);
        printf("variant %d  %-48s %7.3f ms  %6.2f ns/row/CU-lane  %5.2f Grows/s\n", v, name[v], ms, ms * 1e6 / ((double)rows / (256.0 * THREADS)), rows / ms * 1e-6);
    }
    return 0;
}
