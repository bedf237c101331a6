// ubench_gather.hip -- random 32-byte gather ceiling of one MI355X (what bounds the trie walk).
// Every lane does `iters` DEPENDENT rounds of one 32-byte load (two dwordx4) at a pseudo-random slot of a table of
// `mb` MiB; many waves in flight hide latency.  Prints G loads/s and effective GB/s for several table sizes.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_gather ubench_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(64) void k_gather(const uint4* __restrict__ table, uint32_t mask, int iters, uint32_t* out) {
    uint32_t x = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
        x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
        const uint32_t s = x & mask;
        const uint4 a = table[2 * (size_t)s], b = table[2 * (size_t)s + 1];
        acc += a.x + b.w;
        x += acc; // dependent chain, like parent -> child
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    const size_t sizes_mb[] = {1, 4, 32, 128, 512, 2048};
    uint32_t* out;
    hipMalloc(&out, 4);
    for (size_t mb : sizes_mb) {
        const size_t bytes = mb << 20;
        uint4* t;
        if (hipMalloc(&t, bytes) != hipSuccess) break;
        hipMemset(t, 1, bytes);
        const uint32_t mask = (uint32_t)(bytes / 32 - 1);
        for (int lds_kb : {0, 14}) {
            const int blocks = 16384, iters = 64;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64), lds_kb * 1024, 0, t, mask, 8, out);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(64), lds_kb * 1024, 0, t, mask, iters, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double loads = (double)blocks * 64 * iters;
            printf("table %5zu MiB  lds/wave %2d KB : %.3f ms  %.2f G loads/s  %.1f GB/s (32 B each)\n", mb, lds_kb, ms,
                   loads / ms / 1e6, loads * 32 / ms / 1e6);
        }
        hipFree(t);
    }
    return 0;
}
