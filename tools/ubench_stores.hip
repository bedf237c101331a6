// ubench_stores.hip -- what does the WRITE_SIZE counter report for stores of KNOWN size?  (rocprofv3 --pmc WRITE_SIZE -- ./ubench_stores)
// Three kernels write exactly BYTES bytes each, the way the engine's kernels store:
//   k_store16  streaming 16-byte stores, a wave's 64 lanes = 1 KiB contiguous   (k_expand's long ranges)
//   k_store4   4-byte stores, a wave's 64 lanes = 256 bytes contiguous          (k_expand's short ranges, k_retain_expand_dyn)
//   k_store8s  8-byte stores to scattered 8-byte-aligned places                (the matched-range records of the walk kernels)
// tools/calibrate_writes.py divides the bytes written by what the counter says (KiB per dispatch) -> the factor profiles/<round>/write_calibration.json
// holds and tools/collect_profiles.py applies.  build: hipcc --offload-arch=gfx950 -O3 -o ubench_stores ubench_stores.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ __launch_bounds__(256) void k_store16(uint4* out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ __launch_bounds__(256) void k_store4(uint32_t* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void k_store8s(uint2* out, size_t n8, size_t mask) { // every 8-byte slot exactly once, in a scattered order
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) {
        const size_t j = (i * 0x9E3779B97F4A7C15ull) & mask; // odd multiplier: a permutation of [0, 2^k)
        out[j] = make_uint2((uint32_t)i, 7u);
    }
}

int main() {
    const size_t BYTES = 1ull << 30; // 1 GiB per kernel: far beyond L2 + MALL, every byte reaches HBM
    void* buf;
    if (hipMalloc(&buf, BYTES) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, BYTES);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_store16, dim3(8192), dim3(256), 0, 0, (uint4*)buf, BYTES / 16);
        hipLaunchKernelGGL(k_store4, dim3(8192), dim3(256), 0, 0, (uint32_t*)buf, BYTES / 4);
        hipLaunchKernelGGL(k_store8s, dim3(8192), dim3(256), 0, 0, (uint2*)buf, BYTES / 8, BYTES / 8 - 1);
    }
    (void)hipDeviceSynchronize();
    printf("bytes_per_dispatch %zu\n", BYTES);
    (void)hipFree(buf);
    return 0;
}
