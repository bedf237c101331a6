// ubench_occ.hip -- how many one-wave workgroups does a CU of the MI355X really hold?  A kernel that does nothing but wait ~DUR shader
// ticks, launched as 15625 workgroups of 64 threads with a given static LDS size / VGPR count / scratch use; every wave reports when
// and where (HW_ID, XCC_ID) it ran; the host computes the peak number of waves resident on one SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_occ ubench_occ.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int LDS, int SCRATCH, int VG>
__global__ __launch_bounds__(64) void k_wait(uint4* out, uint32_t dur, uint32_t salt) {
    __shared__ uint32_t lds[LDS / 4];
    if (SCRATCH & 2) asm volatile("s_mov_b32 s77, 0" ::: "s77");   // .sgpr_count 78
    if (SCRATCH & 4) asm volatile("v_mov_b32 v63, 0" ::: "v63");   // .vgpr_count 64
    if (SCRATCH & 8) asm volatile("s_mov_b32 s93, 0" ::: "s93");   // .sgpr_count 94
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = salt;
    uint32_t priv[(SCRATCH & 1) ? 64 : 1];
    if (SCRATCH & 1) { // dynamically indexed private array: lives in scratch
        for (int i = 0; i < 64; i++) priv[i] = salt * i;
    }
    uint32_t regs[VG];
#pragma unroll
    for (int i = 0; i < VG; i++) regs[i] = salt + i * threadIdx.x;
    while (__builtin_amdgcn_s_memtime() - t0 < dur) {
#pragma unroll
        for (int i = 0; i < VG; i++) regs[i] = regs[i] * 1664525u + 1013904223u;
        __builtin_amdgcn_s_sleep(8);
    }
    uint32_t acc = lds[(threadIdx.x + 1) & 63];
#pragma unroll
    for (int i = 0; i < VG; i++) acc += regs[i];
    if (SCRATCH & 1) acc += priv[acc & 63];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x] = make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)(t1 - t0) | (acc == 0x1234567u), (hw & 0xFFFFu) | (xcc << 16));
    }
}

struct BigArgs { unsigned long long w[40]; };
template <int LDS, int VTOP = 63>
__global__ __launch_bounds__(64, 8) void k_wait_big(BigArgs b, uint4* out, uint32_t dur, uint32_t salt) {
    __shared__ uint32_t lds[LDS / 4];
    asm volatile("s_mov_b32 s71, 0" ::: "s71");
    if (VTOP == 63) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (VTOP == 57) asm volatile("v_mov_b32 v57, 0" ::: "v57");
    if (VTOP == 59) asm volatile("v_mov_b32 v59, 0" ::: "v59");
    if (VTOP == 55) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    lds[threadIdx.x] = salt + (uint32_t)b.w[salt & 31];
    uint32_t priv[16];
    for (int i = 0; i < 16; i++) priv[i] = salt * i;
    while (__builtin_amdgcn_s_memtime() - t0 < dur) __builtin_amdgcn_s_sleep(8);
    uint32_t acc = lds[(threadIdx.x + 1) & 63] + priv[salt & 15];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11));
        out[blockIdx.x] = make_uint4((uint32_t)t0, (uint32_t)(t0 >> 32), (uint32_t)(t1 - t0) | (acc == 0x1234567u), (hw & 0xFFFFu) | (xcc << 16));
    }
}
static void report(const char* what, uint4* d_out, int n, float ms) {
    std::vector<uint4> h(n);
    hipMemcpy(h.data(), d_out, sizeof(uint4) * n, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<std::pair<unsigned long long, int>>> per;
    int slots[16] = {0};
    for (int i = 0; i < n; i++) {
        const unsigned long long st = ((unsigned long long)h[i].y << 32) | h[i].x, en = st + h[i].z;
        const uint32_t key = (h[i].w >> 4) & 0xFFFFFu & ~0xCu;
        per[key].push_back({st, 1}), per[key].push_back({en, -1});
        slots[h[i].w & 15]++;
    }
    int pk_max = 0;
    double pk_sum = 0;
    for (auto& kv : per) {
        std::sort(kv.second.begin(), kv.second.end());
        int c = 0, pk = 0;
        for (auto& x : kv.second) c += x.second, pk = std::max(pk, c);
        pk_max = std::max(pk_max, pk), pk_sum += pk;
    }
    int pipes[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; i++) pipes[(h[i].w >> 6) & 3]++;
    printf("%s: %.3f ms, peak per SIMD %d (mean of peaks %.2f); pipes %d %d %d %d; wave slots used:", what, ms, pk_max, pk_sum / per.size(), pipes[0], pipes[1],
           pipes[2], pipes[3]);
    for (int i = 0; i < 16; i++) printf(" %d", slots[i]);
    printf("\n");
}
template <int LDS, int VTOP = 63> void run_big(uint4* d_out, int n, bool nonblocking, hipStream_t use = nullptr) {
    hipStream_t st = use;
    if (nonblocking && !use) hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    BigArgs b{};
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k_wait_big<LDS, VTOP>), dim3(n), dim3(64), 0, st, b, d_out, 1000u, 1u);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL((k_wait_big<LDS, VTOP>), dim3(n), dim3(64), 0, st, b, d_out, 100000u, 7u);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k_wait_big<LDS, VTOP>);
    char what[128];
    snprintf(what, sizeof what, "big kernarg, LDS %d, numRegs %d, local %zu, stream %s", LDS, fa.numRegs, fa.localSizeBytes, nonblocking ? "nonblocking" : "null");
    report(what, d_out, n, ms);
}
template <int LDS, int SCRATCH, int VG> void run(uint4* d_out, int n) {
    std::vector<uint4> h(n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL((k_wait<LDS, SCRATCH, VG>), dim3(n), dim3(64), 0, 0, d_out, 1000u, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_wait<LDS, SCRATCH, VG>), dim3(n), dim3(64), 0, 0, d_out, 100000u, 7u);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), d_out, sizeof(uint4) * n, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<std::pair<unsigned long long, int>>> per;
    unsigned long long t0 = ~0ull, t1 = 0;
    double busy = 0;
    for (int i = 0; i < n; i++) {
        const unsigned long long st = ((unsigned long long)h[i].y << 32) | h[i].x, en = st + h[i].z;
        t0 = std::min(t0, st), t1 = std::max(t1, en), busy += h[i].z;
        const uint32_t key = (h[i].w >> 4) & 0xFFFFFu & ~0xCu;
        per[key].push_back({st, 1}), per[key].push_back({en, -1});
    }
    int pk_max = 0;
    double pk_sum = 0;
    for (auto& kv : per) {
        std::sort(kv.second.begin(), kv.second.end());
        int c = 0, pk = 0;
        for (auto& x : kv.second) c += x.second, pk = std::max(pk, c);
        pk_max = std::max(pk_max, pk), pk_sum += pk;
    }
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, (const void*)k_wait<LDS, SCRATCH, VG>);
    printf("LDS %5d B scratch %d regs %3d (numRegs %3d, local %3zu B): %.3f ms, span %llu ticks (%.2f GHz), mean resident %.0f waves = %.2f per SIMD, peak per SIMD %d (mean of peaks %.2f), %zu SIMDs\n",
           LDS, SCRATCH, VG, fa.numRegs, fa.localSizeBytes, ms, t1 - t0, (double)(t1 - t0) / (ms * 1e6), busy / (double)(t1 - t0), busy / (double)(t1 - t0) / 1024.0, pk_max,
           pk_sum / per.size(), per.size());
}

int main() {
    const int n = 15625;
    uint4* d_out;
    hipMalloc(&d_out, sizeof(uint4) * n);
    run_big<5024, 63>(d_out, n, false);
    run_big<5024, 59>(d_out, n, false);
    run_big<5024, 57>(d_out, n, false);
    run_big<5024, 55>(d_out, n, false);
    return 0;
}
