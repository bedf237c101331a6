// ubench_mix.hip -- do L2 hits and L2 misses of random 64-byte lines cost a wave their SUM or their MAX?  (round 4
// added an all-hit and an all-miss measurement to explain k_walk's duration.)  Every lane fetches two random lines per iteration, as
// k_lane of tools/ubench_lines.hip does: both from a table L2 holds (2 MiB), both from one nothing holds (2 GiB), or one from each;
// 5 KB of LDS per one-wave workgroup = k_walk's 8 waves per SIMD.  If the mixed run takes about (hit + miss) / 2 the two resources serialise,
// if it takes about miss / 2 (= the max of the two halves) they overlap and the all-miss rate alone is the floor.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_mix tools/ubench_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(64) void k_two(const uint4* __restrict__ ta, uint32_t mask_a, const uint4* __restrict__ tb, uint32_t mask_b, int iters, uint32_t* out) {
    extern __shared__ uint32_t lds[];
    uint32_t x = (blockIdx.x * 64 + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (int i = 0; i < iters; i++) {
        x = mix(x);
        const uint4* pa = ta + 4 * (size_t)(x & mask_a);
        const uint4* pb = tb + 4 * (size_t)(mix(x ^ 0x55555555u) & mask_b);
        const uint4 a0 = pa[0], a1 = pa[1], a2 = pa[2], a3 = pa[3], b0 = pb[0], b1 = pb[1], b2 = pb[2], b3 = pb[3];
        acc += a0.x + a1.y + a2.z + a3.w + b0.x + b1.y + b2.z + b3.w;
        x += acc;
    }
    if (acc == 0x12345678u) out[0] = acc + lds[threadIdx.x];
}

static double run(const char* name, const uint4* ta, uint32_t ma, const uint4* tb, uint32_t mb, uint32_t* out) {
    const int blocks = 32768, iters = 32;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_two, dim3(blocks), dim3(64), 5 * 1024, 0, ta, ma, tb, mb, 4, out);
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_two, dim3(blocks), dim3(64), 5 * 1024, 0, ta, ma, tb, mb, iters, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double lines = (double)blocks * 64 * iters * 2;
    printf("  %-10s %.3f ms  %.1f G lines/s\n", name, best, lines / best / 1e6);
    return best;
}

int main() {
    uint32_t* out; hipMalloc(&out, 4);
    uint4 *small, *big;
    const size_t sb = 2ull << 20, bb = 2048ull << 20;
    if (hipMalloc(&small, sb) != hipSuccess || hipMalloc(&big, bb) != hipSuccess) return 1;
    hipMemset(small, 1, sb); hipMemset(big, 1, bb);
    const uint32_t ms_ = (uint32_t)(sb / 64 - 1), mb_ = (uint32_t)(bb / 64 - 1);
    const double hh = run("hit+hit", small, ms_, small, ms_, out);
    const double mm = run("miss+miss", big, mb_, big, mb_, out);
    const double hm = run("hit+miss", small, ms_, big, mb_, out);
    printf("  mixed %.3f ms: sum model (hh + mm) / 2 = %.3f, max model max(hh, mm) / 2 = %.3f\n", hm, (hh + mm) / 2, (hh > mm ? hh : mm) / 2);
    return 0;
}
