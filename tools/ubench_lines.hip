// ubench_lines.hip -- how should a wave fetch random 64-byte lines?  Same bytes, same instruction count:
//   A "lane":  every lane fetches its own 2 random lines with 8 dwordx4 loads (what a per-lane cuckoo probe does)
//   B "quad":  every 4 adjacent lanes fetch one line cooperatively (16 B each); 4 groups x 2 lines per iteration
// Reports G lines/s.  build: hipcc --offload-arch=gfx950 -O3 -o ubench_lines ubench_lines.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(64) void k_lane(const uint4* __restrict__ t, uint32_t mask, int iters, uint32_t* out) {
    uint32_t x = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = mix(x);
        const uint4* pa = t + 4 * (size_t)(x & mask);
        const uint4* pb = t + 4 * (size_t)(mix(x ^ 0x55555555u) & mask);
        const uint4 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
        acc += a0.x + a1.y + a2.z + a3.w + b0.x + b1.y + b2.z + b3.w;
        x += acc;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// 64 items per iteration as well: 4 groups of 16 items, each item fetched by a quad
__global__ __launch_bounds__(64) void k_quad(const uint4* __restrict__ t, uint32_t mask, int iters, uint32_t* out) {
    const uint32_t lane = threadIdx.x, part = lane & 3, q = lane >> 2;
    uint32_t x = (blockIdx.x * 64 + q) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        uint4 v[8];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const uint32_t y = mix(x + g * 0x9E3779B1u);
            v[2 * g] = t[4 * (size_t)(y & mask) + part];
            v[2 * g + 1] = t[4 * (size_t)(mix(y ^ 0x55555555u) & mask) + part];
        }
#pragma unroll
        for (int g = 0; g < 8; g++) acc += v[g].x + v[g].w;
        x = mix(x) + __shfl(acc, lane & ~3u);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// C "lane, key halves only": 4 loads of 16 B from 4 different 32-byte slots (2 lines), then one dependent 16 B
__global__ __launch_bounds__(64) void k_lane5(const uint4* __restrict__ t, uint32_t mask, int iters, uint32_t* out) {
    uint32_t x = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = mix(x);
        const uint4* pa = t + 4 * (size_t)(x & mask);
        const uint4* pb = t + 4 * (size_t)(mix(x ^ 0x55555555u) & mask);
        const uint4 a0 = pa[0], a2 = pa[2], b0 = pb[0], b2 = pb[2];
        const uint32_t s = a0.x + a2.z + b0.x + b2.z;
        const uint4 p = (s & 1) ? pa[1] : pb[3];
        acc += s + p.y;
        x += acc;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <class K> void run(const char* name, K kern, const uint4* t, uint32_t mask, uint32_t* out, int lds_kb) {
    const int blocks = 16384, iters = 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds_kb * 1024, 0, t, mask, 4, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds_kb * 1024, 0, t, mask, iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double lines = (double)blocks * 64 * iters * 2;
    printf("  %-6s lds %2d KB: %.3f ms  %.1f G lines/s  (%.1f G items/s)\n", name, lds_kb, ms, lines / ms / 1e6, lines / 2 / ms / 1e6);
}

int main() {
    uint32_t* out; hipMalloc(&out, 4);
    for (size_t mb : {2, 64, 2048}) {
        const size_t bytes = mb << 20;
        uint4* t; if (hipMalloc(&t, bytes) != hipSuccess) break;
        hipMemset(t, 1, bytes);
        const uint32_t mask = (uint32_t)(bytes / 64 - 1);
        printf("table %zu MiB\n", mb);
        for (int lds : {0, 14}) {
            run("lane8", k_lane, t, mask, out, lds);
            run("lane5", k_lane5, t, mask, out, lds);
            run("quad", k_quad, t, mask, out, lds);
        }
        hipFree(t);
    }
    return 0;
}
