// ubench_write_skew.hip -- how much of k_expand's time on C4 is the SHAPE of the work?  One wave per 64-row block as k_expand has them, every wave streams
// its block's bytes with 16-byte stores and nothing else (sizes: one number per line = ids of a block, written by tools/c4_order_probe.py); the same bytes
// again in equal pieces, and in slices of at most 32 k ids (128 KiB) handed to one wave each.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_write_skew tools/ubench_write_skew.hip && ./ubench_write_skew gpurun_out/c4_block_ids.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(64) void k_pieces(uint4* out, const unsigned long long* off16) { // wave b writes quads off16[b] .. off16[b + 1]
    const unsigned long long b = off16[blockIdx.x], e = off16[blockIdx.x + 1];
    for (unsigned long long i = b + threadIdx.x; i < e; i += 64) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main(int argc, char** argv) {
    std::vector<unsigned long long> ids;
    FILE* f = fopen(argc > 1 ? argv[1] : "gpurun_out/c4_block_ids.txt", "r");
    if (!f) return 1;
    unsigned long long v;
    while (fscanf(f, "%llu", &v) == 1) ids.push_back(v);
    fclose(f);
    unsigned long long total16 = 0;
    for (auto x : ids) total16 += (x + 3) / 4;
    void* buf;
    if (hipMalloc(&buf, total16 * 16 + 64) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    auto run = [&](const char* name, const std::vector<unsigned long long>& off) {
        unsigned long long* d;
        (void)hipMalloc(&d, off.size() * 8);
        (void)hipMemcpy(d, off.data(), off.size() * 8, hipMemcpyHostToDevice);
        float best = 1e9f;
        for (int r = 0; r < 6; r++) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_pieces, dim3((uint32_t)off.size() - 1), dim3(64), 0, 0, (uint4*)buf, d);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (r) best = ms < best ? ms : best;
        }
        printf("%-64s %6zu waves  %.3f ms = %.2f TB/s\n", name, off.size() - 1, best, total16 * 16 / (best * 1e-3) / 1e12);
        (void)hipFree(d);
    };
    std::vector<unsigned long long> off(1, 0);
    for (auto x : ids) off.push_back(off.back() + (x + 3) / 4);
    printf("%zu blocks, %.3f GB, largest block %.2f MB, mean %.2f MB\n", ids.size(), total16 * 16 / 1e9, *std::max_element(ids.begin(), ids.end()) * 4 / 1e6, total16 * 16 / 1e6 / ids.size());
    run("one wave per block (the blocks' own sizes)", off);
    {
        std::vector<unsigned long long> eq(1, 0);
        for (size_t i = 1; i <= ids.size(); i++) eq.push_back(total16 * i / ids.size());
        run("the same number of waves, equal pieces", eq);
    }
    for (unsigned long long slice16 : {8192ull, 2048ull}) { // 32 k ids, 8 k ids
        std::vector<unsigned long long> sl(1, 0);
        for (size_t i = 0; i + 1 < off.size(); i++)
            for (unsigned long long p = off[i]; p < off[i + 1];) {
                p = std::min(off[i + 1], p + slice16);
                sl.push_back(p);
            }
        char nm[96];
        snprintf(nm, sizeof nm, "blocks cut into slices of <= %llu KiB, one wave per slice", slice16 * 16 / 1024);
        run(nm, sl);
    }
    { // the four-way split of heavy blocks (> 2 x the mean) k_expand has for the dist direction
        std::vector<unsigned long long> sp(1, 0);
        const unsigned long long mean16 = total16 / ids.size();
        for (size_t i = 0; i + 1 < off.size(); i++) {
            const unsigned long long n = off[i + 1] - off[i];
            const int parts = n > 2 * mean16 ? 4 : 1;
            for (int p = 1; p <= parts; p++) sp.push_back(off[i] + n * p / parts);
        }
        run("blocks of more than 2 x the mean cut into four equal parts", sp);
    }
    return 0;
}
