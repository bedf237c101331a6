// wave_emu.h -- a wave64 emulator for HOST-side logic tests of single-wave HIP kernels (test tooling, not product: nothing under
// bifromq_amd/ includes it, and it never stands in for the GPU path).
//
// A kernel body is compiled by g++ unchanged; its 64 lanes run as 64 fibers (ucontext) of one thread.  Every cross-lane operation
// (ballot, readlane, readfirstlane, DPP moves, shuffles, wave_sync) is a rendezvous: a fiber parks there until all lanes that have not
// returned are parked at the SAME operation (same kind, same parameters, same running number) -- anything else is reported as
// divergence and aborts.  Between two rendezvous a lane runs ahead of the others on its own, so code that relies on lockstep execution
// without a wave_sync() between an LDS write and another lane's read of it fails here (on purpose).  `__shared__` variables are function
// statics: one workgroup at a time, contents survive from one workgroup to the next like stale LDS does.
#pragma once
#include <ucontext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>

struct uint4 {
    uint32_t x, y, z, w;
};
struct uint2 {
    uint32_t x, y;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }

#define __global__
#define __device__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)

namespace wemu {

enum Kind : int { K_SYNC = 1, K_BALLOT, K_FIRST, K_READLANE, K_DPP, K_SHFL, K_SHFL_UP, K_SHFL_XOR };

struct Dim3 {
    uint32_t x, y, z;
};

struct State {
    ucontext_t sched;
    ucontext_t ctx[64];
    char* stacks = nullptr;
    bool done[64], parked[64];
    uint32_t seq[64];
    int kind[64];
    uint64_t param[64];
    uint64_t val[64], res[64];
    int cur = -1;
    uint32_t block = 0;
    std::function<void()> body;
    unsigned long long rendezvous = 0;
};
inline State& st() {
    static State s;
    return s;
}
inline Dim3 tid() { return Dim3{(uint32_t)st().cur, 0, 0}; }
inline Dim3 bid() { return Dim3{st().block, 0, 0}; }
inline uint32_t& grid_size() { // what gridDim.x reads (persistent kernels stride by it); set by the harness
    static uint32_t g = 1;
    return g;
}
inline Dim3 gdim() { return Dim3{grid_size(), 1, 1}; }

inline uint64_t& trace_lanes() { // WEMU_TRACE=<lane mask>: every cross-lane operation of these lanes goes to stderr (debugging a divergence)
    static uint64_t m = getenv("WEMU_TRACE") ? strtoull(getenv("WEMU_TRACE"), nullptr, 0) : 0;
    return m;
}
inline uint64_t collective(int kind, uint64_t param, uint64_t value) {
    State& s = st();
    const int me = s.cur;
    s.kind[me] = kind, s.param[me] = param, s.val[me] = value, s.parked[me] = true, s.seq[me]++;
    if (trace_lanes() >> me & 1ull) fprintf(stderr, "wave_emu: lane %d #%u kind %d param %llu value %llu\n", me, s.seq[me], kind, (unsigned long long)param, (unsigned long long)value);
    swapcontext(&s.ctx[me], &s.sched);
    return s.res[me];
}

inline void fiber_main() {
    State& s = st();
    s.body();
    s.done[s.cur] = true;
    swapcontext(&s.ctx[s.cur], &s.sched);
}

inline void resolve() {
    State& s = st();
    int first = -1;
    for (int l = 0; l < 64; l++)
        if (!s.done[l]) {
            if (first < 0) first = l;
            if (s.kind[l] != s.kind[first] || s.seq[l] != s.seq[first] || (s.kind[l] != K_SHFL && s.param[l] != s.param[first])) {
                fprintf(stderr, "wave_emu: divergent cross-lane operation: lane %d kind %d #%u param %llu, lane %d kind %d #%u param %llu\n", first,
                        s.kind[first], s.seq[first], (unsigned long long)s.param[first], l, s.kind[l], s.seq[l], (unsigned long long)s.param[l]);
                abort();
            }
        }
    const int kind = s.kind[first];
    const uint64_t param = s.param[first];
    s.rendezvous++;
    auto active = [&](int l) { return l >= 0 && l < 64 && !s.done[l]; };
    switch (kind) {
    case K_SYNC:
        break;
    case K_BALLOT: {
        uint64_t m = 0;
        for (int l = 0; l < 64; l++)
            if (active(l) && s.val[l]) m |= 1ull << l;
        for (int l = 0; l < 64; l++) s.res[l] = m;
        break;
    }
    case K_FIRST:
        for (int l = 0; l < 64; l++) s.res[l] = s.val[first];
        break;
    case K_READLANE:
        if (!active((int)param)) {
            fprintf(stderr, "wave_emu: readlane of inactive lane %d\n", (int)param);
            abort();
        }
        for (int l = 0; l < 64; l++) s.res[l] = s.val[param];
        break;
    case K_DPP: { // param = ctrl | row_mask << 16; bank mask 0xF, bound_ctrl off, old = 0
        const uint32_t ctrl = (uint32_t)param & 0xFFFFu, rows = (uint32_t)(param >> 16);
        for (int l = 0; l < 64; l++) {
            int src = -1;
            if (ctrl >= 0x111 && ctrl <= 0x11F) { // row_shr:n
                const int n = (int)ctrl - 0x110;
                if ((l & 15) >= n) src = l - n;
            } else if (ctrl == 0x142) { // row_bcast:15 -- lane 15 of every row to the next row
                if (l >= 16) src = (l & ~15) - 1;
            } else if (ctrl == 0x143) { // row_bcast:31 -- lane 31 to rows 2 and 3
                if (l >= 32) src = 31;
            } else {
                fprintf(stderr, "wave_emu: DPP control %#x not modelled\n", ctrl);
                abort();
            }
            const bool row_on = (rows >> (l >> 4)) & 1u;
            s.res[l] = (row_on && active(src)) ? s.val[src] : 0;
        }
        break;
    }
    case K_SHFL:
        for (int l = 0; l < 64; l++) {
            const int src = (int)(s.param[l] & 63u);
            s.res[l] = active(src) ? s.val[src] : s.val[l];
        }
        break;
    case K_SHFL_UP:
        for (int l = 0; l < 64; l++) s.res[l] = (l >= (int)param && active(l - (int)param)) ? s.val[l - (int)param] : s.val[l];
        break;
    case K_SHFL_XOR:
        for (int l = 0; l < 64; l++) s.res[l] = active(l ^ (int)param) ? s.val[l ^ (int)param] : s.val[l];
        break;
    default:
        abort();
    }
    for (int l = 0; l < 64; l++) s.parked[l] = false;
}

// runs `body` as workgroup `block` (one wave of 64 lanes)
inline void run_wave(uint32_t block, std::function<void()> body) {
    State& s = st();
    const size_t STK = 256 * 1024;
    if (!s.stacks) s.stacks = (char*)malloc(64 * STK);
    s.block = block;
    s.body = std::move(body);
    for (int l = 0; l < 64; l++) {
        s.done[l] = s.parked[l] = false;
        s.seq[l] = 0;
        getcontext(&s.ctx[l]);
        s.ctx[l].uc_stack.ss_sp = s.stacks + (size_t)l * STK;
        s.ctx[l].uc_stack.ss_size = STK;
        s.ctx[l].uc_link = &s.sched;
        makecontext(&s.ctx[l], (void (*)())fiber_main, 0);
    }
    for (;;) { // a lane runs until it parks at a cross-lane operation or returns: after one round every live lane is parked
        bool all_done = true;
        for (int l = 0; l < 64; l++)
            if (!s.done[l] && !s.parked[l]) {
                s.cur = l;
                swapcontext(&s.sched, &s.ctx[l]);
            }
        for (int l = 0; l < 64; l++) all_done = all_done && s.done[l];
        if (all_done) break;
        resolve();
    }
    s.cur = -1;
}

} // namespace wemu

#define threadIdx (wemu::tid())
#define blockIdx (wemu::bid())
#define gridDim (wemu::gdim())

// ---- the cross-lane vocabulary of the kernels ----
namespace bmq {
inline void wave_sync() { wemu::collective(wemu::K_SYNC, 0, 0); }
inline unsigned long long ballot64(bool p) { return wemu::collective(wemu::K_BALLOT, 0, p ? 1 : 0); }
inline uint32_t sgpr(uint32_t v) { return (uint32_t)wemu::collective(wemu::K_FIRST, 0, v); }
inline uint32_t read_lane(uint32_t v, uint32_t l) { return (uint32_t)wemu::collective(wemu::K_READLANE, l, v); }
template <int CTRL, int ROWS> inline uint32_t dpp_take(uint32_t v) { return (uint32_t)wemu::collective(wemu::K_DPP, (uint64_t)CTRL | ((uint64_t)ROWS << 16), v); }
inline uint32_t rank_below(unsigned long long mask) { return (uint32_t)__builtin_popcountll(mask & ((1ull << threadIdx.x) - 1ull)); }
inline uint32_t lane_bit(unsigned long long m) { return (uint32_t)(m >> threadIdx.x) & 1u; }
inline uint32_t first_bit(unsigned long long m) { return (uint32_t)__builtin_ffsll((long long)m) - 1u; }
inline uint32_t count_bits(unsigned long long m) { return (uint32_t)__builtin_popcountll(m); }
inline uint32_t uniform_word(const uint32_t* p) { return *p; }
inline uint32_t copy_here(uint32_t v) { return v; }
inline uint32_t lane_here() { return threadIdx.x; }
} // namespace bmq

// HIP spellings the older kernels use
inline unsigned long long __ballot(bool p) { return wemu::collective(wemu::K_BALLOT, 0, p ? 1 : 0); }
inline bool __all(bool p) { return wemu::collective(wemu::K_BALLOT, 1, p ? 0 : 1) == 0; }
inline uint32_t __shfl(uint32_t v, uint32_t l) { return (uint32_t)wemu::collective(wemu::K_SHFL, l, v); }
inline uint32_t __shfl(uint32_t v, int l) { return (uint32_t)wemu::collective(wemu::K_SHFL, (uint32_t)l, v); }
inline uint32_t __shfl_up(uint32_t v, int d) { return (uint32_t)wemu::collective(wemu::K_SHFL_UP, (uint64_t)d, v); }
inline unsigned long long __shfl_xor(unsigned long long v, int d) { return wemu::collective(wemu::K_SHFL_XOR, (uint64_t)d, v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
#define WEMU_ATOMIC(name, op, T)        \
    inline T name(T* p, T v) {          \
        const T o = *p;                 \
        *p = o op v;                    \
        return o;                       \
    }
WEMU_ATOMIC(atomicOr, |, uint32_t)
WEMU_ATOMIC(atomicOr, |, unsigned long long)
WEMU_ATOMIC(atomicAdd, +, uint32_t)
WEMU_ATOMIC(atomicAdd, +, unsigned long long)
#undef WEMU_ATOMIC
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
    const unsigned long long o = *p;
    *p = o > v ? o : v;
    return o;
}
using std::max;
using std::min;
