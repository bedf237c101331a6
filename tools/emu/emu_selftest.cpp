// emu_selftest.cpp -- the emulator's own cross-lane operations against their definitions (tests/test_expand_emu.py runs it):
// ballot with lanes that have returned, readlane / readfirstlane, the DPP row shifts and broadcasts as the inclusive scan of
// bmq_expand_kernel.h uses them (against a serial prefix sum, for random inputs and with a tail of lanes gone), LDS hand-over through
// wave_sync, and -- in a forked child -- that a divergent cross-lane operation aborts instead of pairing up lanes that are not at the same place.
#define BMQ_WAVE_EMU 1
#include "wave_emu.h"

#include <sys/wait.h>
#include <unistd.h>

#include <random>

namespace bmq {
// the scan under test, as in bifromq_amd/csrc/bmq_expand_kernel.h
inline uint32_t wave_incl_scan(uint32_t v) {
    v += dpp_take<0x111, 0xF>(v);
    v += dpp_take<0x112, 0xF>(v);
    v += dpp_take<0x114, 0xF>(v);
    v += dpp_take<0x118, 0xF>(v);
    v += dpp_take<0x142, 0xA>(v);
    v += dpp_take<0x143, 0xC>(v);
    return v;
}
} // namespace bmq
using namespace bmq;

static int g_fail = 0;
#define CHECK(c)                                                  \
    do {                                                          \
        if (!(c)) {                                               \
            fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            g_fail++;                                             \
        }                                                         \
    } while (0)

int main() {
    std::mt19937_64 rng(9);
    for (int round = 0; round < 50; round++) {
        uint32_t in[64], out[64], ball[64], first[64], rl[64];
        const uint32_t alive = round % 5 == 4 ? 1 + (uint32_t)(rng() % 64) : 64; // lanes >= alive return at once
        for (auto& x : in) x = (uint32_t)(rng() % 1000);
        static uint32_t lds[64];
        uint32_t handed[64];
        wemu::run_wave(0, [&] {
            const uint32_t lane = threadIdx.x;
            if (lane >= alive) return;
            out[lane] = wave_incl_scan(in[lane]);
            ball[lane] = (uint32_t)__builtin_popcountll(ballot64((in[lane] & 1u) != 0));
            first[lane] = sgpr(in[lane]);
            rl[lane] = read_lane(in[lane], alive - 1);
            lds[lane] = in[lane] * 3u;
            wave_sync();
            handed[lane] = lds[(lane + 1) % alive]; // written by another lane, which may not have run yet without the sync
        });
        uint32_t acc = 0, odd = 0;
        for (uint32_t l = 0; l < alive; l++) odd += in[l] & 1u;
        for (uint32_t l = 0; l < alive; l++) {
            acc += in[l];
            if (alive == 64) CHECK(out[l] == acc); // (with lanes gone the scan's sources are partly missing: only the full wave is defined)
            CHECK(ball[l] == odd);
            CHECK(first[l] == in[0]);
            CHECK(rl[l] == in[alive - 1]);
            CHECK(handed[l] == in[(l + 1) % alive] * 3u);
        }
    }
    // divergence: half of the lanes ask for a ballot, the other half for a wave_sync -> abort()
    const pid_t pid = fork();
    if (pid == 0) {
        fclose(stderr);
        wemu::run_wave(0, [&] {
            if (threadIdx.x & 1u) (void)ballot64(true);
            else wave_sync();
        });
        _exit(0); // not reached
    }
    int status = 0;
    waitpid(pid, &status, 0);
    CHECK(WIFSIGNALED(status) && WTERMSIG(status) == SIGABRT);
    if (g_fail) return 1;
    printf("emu selftest ok\n");
    return 0;
}
