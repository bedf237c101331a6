// ubench_write_bw.hip -- what is the ceiling of a kernel that only WRITES?  k_expand on C2 / C4 streams 4.2 / 2.2 GB of ids per launch at 4.3-4.4 TB/s
// (0.54-0.55 of the 8 TB/s the roofline prices against); this times pure stores of 4 GiB in the shapes such a kernel could use.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_write_bw tools/ubench_write_bw.hip && ./ubench_write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store(uint4* p, const uint4& v) {
    v4u x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<v4u*>(p));
}

template <bool NT> __global__ __launch_bounds__(256) void k_grid16(uint4* out, size_t n16) { // grid-stride, a wave's 64 lanes = 1 KiB contiguous
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = make_uint4((uint32_t)i, 1u, 2u, 3u);
        if (NT) nt_store(out + i, v);
        else out[i] = v;
    }
}
// one-wave workgroups, every wave owns ONE contiguous piece of `per` bytes (k_expand: a wave's rows are one piece of the output)
template <bool NT, int UNROLL> __global__ __launch_bounds__(64) void k_piece16(uint4* out, size_t per16) {
    uint4* p = out + (size_t)blockIdx.x * per16;
    for (size_t i = threadIdx.x; i < per16; i += 64 * UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint4 v = make_uint4((uint32_t)i, 1u, 2u, 3u);
            if (i + 64 * u < per16) {
                if (NT) nt_store(p + i + 64 * u, v);
                else p[i + 64 * u] = v;
            }
        }
    }
}
// ... where a piece is a run of RANGES of `len` ids each, every range streamed the way k_expand streams a long range: ALIGNED = a head of 4-byte stores up to
// the next 16-byte boundary, 16-byte stores, a tail; otherwise 16-byte stores from the range's first id on, wherever it lies, and a tail (pieces start 4 bytes off)
template <bool ALIGNED> __global__ __launch_bounds__(64) void k_ranges(uint32_t* out, size_t per4, uint32_t len) {
    uint32_t* p = out + (size_t)blockIdx.x * per4 + 1;
    const uint32_t lane = threadIdx.x;
    for (size_t r0 = 0; r0 + len < per4; r0 += len) {
        uint32_t* const dst = p + r0;
        const uint32_t b = (uint32_t)r0, c = len;
        const uint32_t head = ALIGNED ? min((uint32_t)(((16u - ((uintptr_t)dst & 15u)) & 15u) >> 2), c) : 0u;
        if (lane < head) dst[lane] = b + lane;
        const uint32_t n4 = (c - head) >> 2;
        for (uint32_t q = lane; q < n4; q += 64) {
            const uint32_t v = b + head + 4 * q;
            uint32_t* d = dst + head + 4 * q;
            const v4u x = {v, v + 1, v + 2, v + 3};
            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d), "v"(x) : "memory");
        }
        for (uint32_t o = head + 4 * n4 + lane; o < c; o += 64) dst[o] = b + o;
    }
}
__global__ __launch_bounds__(64) void k_piece4(uint32_t* out, size_t per4) { // ... with 4-byte stores (a wave's 64 lanes = 256 bytes)
    uint32_t* p = out + (size_t)blockIdx.x * per4;
    for (size_t i = threadIdx.x; i < per4; i += 64) p[i] = (uint32_t)i;
}

int main() {
    const size_t BYTES = 4ull << 30;
    void* buf;
    if (hipMalloc(&buf, BYTES) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    auto time_it = [&](const char* name, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        float best = 1e9f, sum = 0;
        for (int r = 0; r < 5; r++) {
            (void)hipEventRecord(e0, 0);
            launch();
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best, sum += ms;
        }
        printf("%-58s %.3f ms best, %.3f mean  = %.2f TB/s (best)\n", name, best, sum / 5, BYTES / (best * 1e-3) / 1e12);
    };
    time_it("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 1, BYTES, 0); });
    time_it("grid-stride 16-byte stores, 8192 x 256", [&] { hipLaunchKernelGGL(k_grid16<false>, dim3(8192), dim3(256), 0, 0, (uint4*)buf, BYTES / 16); });
    time_it("grid-stride 16-byte stores, 2048 x 256", [&] { hipLaunchKernelGGL(k_grid16<false>, dim3(2048), dim3(256), 0, 0, (uint4*)buf, BYTES / 16); });
    time_it("grid-stride 16-byte nontemporal stores, 8192 x 256", [&] { hipLaunchKernelGGL(k_grid16<true>, dim3(8192), dim3(256), 0, 0, (uint4*)buf, BYTES / 16); });
    for (size_t per : {64ull << 10, 256ull << 10, 1ull << 20}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one wave per %zu KiB piece, 16-byte stores", per >> 10);
        time_it(nm, [&] { hipLaunchKernelGGL((k_piece16<false, 1>), dim3(BYTES / per), dim3(64), 0, 0, (uint4*)buf, per / 16); });
        snprintf(nm, sizeof nm, "one wave per %zu KiB piece, 16-byte stores x 4 unrolled", per >> 10);
        time_it(nm, [&] { hipLaunchKernelGGL((k_piece16<false, 4>), dim3(BYTES / per), dim3(64), 0, 0, (uint4*)buf, per / 16); });
        snprintf(nm, sizeof nm, "one wave per %zu KiB piece, nontemporal 16-byte stores", per >> 10);
        time_it(nm, [&] { hipLaunchKernelGGL((k_piece16<true, 1>), dim3(BYTES / per), dim3(64), 0, 0, (uint4*)buf, per / 16); });
    }
    for (uint32_t len : {117u, 701u, 2003u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "one wave per 256 KiB piece, ranges of %u ids, aligned head + 16 B + tail", len);
        time_it(nm, [&] { hipLaunchKernelGGL(k_ranges<true>, dim3(BYTES / (256 << 10) - 1), dim3(64), 0, 0, (uint32_t*)buf, (size_t)(256 << 10) / 4, len); });
        snprintf(nm, sizeof nm, "one wave per 256 KiB piece, ranges of %u ids, unaligned 16 B + tail", len);
        time_it(nm, [&] { hipLaunchKernelGGL(k_ranges<false>, dim3(BYTES / (256 << 10) - 1), dim3(64), 0, 0, (uint32_t*)buf, (size_t)(256 << 10) / 4, len); });
    }
    time_it("one wave per 256 KiB piece, 4-byte stores", [&] { hipLaunchKernelGGL(k_piece4, dim3(BYTES / (256 << 10)), dim3(64), 0, 0, (uint32_t*)buf, (256 << 10) / 4); });
    (void)hipFree(buf);
    return 0;
}
