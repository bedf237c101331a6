// bmq_exec_dev.h -- DevExec: the builder functions of bmq_build_core.h as gfx950 kernels on the engine's HIP stream, over
// HBM-resident arrays (see bmq_dist_index.h for the Exec concept).  One lane per route key / directory slot / trie slot:
// the work is pointer chasing with lock-free CAS claims, bound by memory latency -- occupancy comes from the number of keys
// in a batch (100 k ops = 1563 waves; a 10 M-key bulk load = 156 k waves), not from anything clever inside a lane.
// The grouping sort and the tenant-run scan are rocPRIM/hipCUB device primitives (radix sort of 42-bit targets, prefix sum).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "bmq_build_core.h"
#include "bmq_fanout_core.h"
#include "bmq_fanout_kernels.h"
#include "bmq_retain_core.h"

namespace bmq {

constexpr int BK = 64; // lanes per builder workgroup: one wave, independent lanes

__global__ __launch_bounds__(256) void k_b_fill_slots(TrieSlot* p, unsigned long long n) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        uint4* q = reinterpret_cast<uint4*>(p + i);
        q[0] = make_uint4(NONE, 0u, 0u, 0u);
        q[1] = make_uint4(0u, 0u, NONE, 0u);
    }
}
__global__ __launch_bounds__(256) void k_b_iota(uint32_t* p, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = i;
}
__global__ __launch_bounds__(BK) void k_b_prepare(DistIndexMut ix, OpBatch ob) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) prepare_one(ix, ob, i);
}
__global__ __launch_bounds__(BK) void k_b_prepare_check(DistIndexMut ix, OpBatch ob, uint32_t n_dir) {
    if (ix.bc->gate) return; // (an earlier stage of the batch handed over to the host: BuildCounters.gate)
    const uint32_t d = blockIdx.x * BK + threadIdx.x;
    if (d < n_dir) prepare_check_one(ix, ob, d);
}
// closes the gate behind a stage that left work for the host
__global__ void k_b_gate(BuildCounters* bc, uint32_t stage) {
    if (threadIdx.x || blockIdx.x || bc->gate) return;
    if (bc->err || bc->n_unknown || bc->n_grow || bc->n_deferred) bc->gate = stage;
}
__global__ __launch_bounds__(BK) void k_b_bulk_prepare(DistIndexMut ix, OpBatch ob) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) bulk_prepare_one(ix, ob, i);
}
__global__ __launch_bounds__(BK) void k_b_bulk_tenants(DistIndexMut ix, OpBatch ob, const uint32_t* scan) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) bulk_tenants_one(ix, ob, i, scan);
}
__global__ __launch_bounds__(BK) void k_b_bulk_counts(OpBatch ob, uint32_t n_ten, const uint32_t* nn_incl) {
    const uint32_t t = blockIdx.x * BK + threadIdx.x;
    if (t < n_ten) bulk_counts_one(ob, t, n_ten, nn_incl);
}
__global__ __launch_bounds__(BK) void k_b_locate(DistIndexMut ix, OpBatch ob, uint32_t phase) {
    if (ix.bc->gate) return;
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) locate_one(ix, ob, i, phase);
}
__global__ __launch_bounds__(BK) void k_b_group(DistIndexMut ix, OpBatch ob) {
    if (ix.bc->gate) return;
    const uint32_t p = blockIdx.x * BK + threadIdx.x;
    if (p < ob.n) group_one(ix, ob, p);
}
__global__ __launch_bounds__(BK) void k_b_rehash(DistIndexMut ix, uint32_t old_base, uint32_t old_slots, uint32_t new_base, uint32_t new_buckets, uint32_t pass, uint32_t d) {
    const uint32_t s = blockIdx.x * BK + threadIdx.x;
    if (s < old_slots) rehash_one(ix, old_base, new_base, new_buckets, s, pass, d);
}
__global__ __launch_bounds__(BK) void k_b_dict_rehash(const DictSlot* old, uint32_t old_slots, DistIndexMut ix) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < old_slots) dict_rehash_one(old, i, ix);
}
__global__ __launch_bounds__(64) void k_b_find(DistIndexMut ix, const uint8_t* q, uint32_t tenant_len, uint32_t filter_len, uint32_t* out, uint32_t cap) {
    if (threadIdx.x == 0 && blockIdx.x == 0) find_copy(ix, q, tenant_len, filter_len, out, cap);
}
__global__ __launch_bounds__(256) void k_b_gather_refs(DistIndexMut ix, const uint32_t* ids, uint32_t n, uint32_t id_end, unsigned long long* out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) gather_ref_one(ix, ids, i, id_end, out);
}
__global__ __launch_bounds__(BK) void k_b_gather_bytes(DistIndexMut ix, const unsigned long long* refs, const uint64_t* offs, uint32_t n, uint8_t* out) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < n) gather_bytes_one(ix, refs, offs, i, out);
}

// fan-out grouping (bmq_fanout_core.h): one lane per (topic, route) pair
__global__ __launch_bounds__(256) void k_fo_fill(DistIndexMut ix, FanoutState st, FanoutBatch b) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.total) fo_fill_one(ix, st, b, i);
}
__global__ __launch_bounds__(256) void k_fo_verify(DistIndexMut ix, FanoutState st, FanoutBatch b) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.total) fo_verify_one(ix, st, b, i);
}
__global__ __launch_bounds__(256) void k_fo_keys(DistIndexMut ix, FanoutState st, FanoutBatch b) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < b.total) fo_key_one(ix, st, b, i);
}
__global__ __launch_bounds__(256) void k_fo_emit(FanoutBatch b) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < b.total) fo_emit_one(b, j);
}
__global__ __launch_bounds__(256) void k_fo_groups(FanoutState st, FanoutBatch b) {
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j < b.total) fo_group_one(st, b, j);
}

// retain direction: mutation of the retained-topic index (bmq_retain_core.h), one lane per op / per id
__global__ __launch_bounds__(BK) void k_r_locate(RetainMut m, RetainOps ob, uint32_t phase) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) rlocate_one(m, ob, i, phase);
}
__global__ __launch_bounds__(BK) void k_r_commit(RetainMut m, RetainOps ob) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < ob.n) rcommit_one(m, ob, i);
}
// dead ids in front of every 64-id word: ONE workgroup (a million ids are 16 k words; 100 M ids 1.6 M words = 1.5 k per thread)
__global__ __launch_bounds__(1024) void k_r_rank(const unsigned long long* bits, uint32_t* rank, uint32_t n_words) {
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_words + 1023) / 1024, lo = min(n_words, tid * per), hi = min(n_words, lo + per);
    uint32_t s = 0;
    for (uint32_t w = lo; w < hi; w++) s += (uint32_t)__popcll(bits[w]);
    part[tid] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = tid >= d ? part[tid - d] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = part[tid] - s;
    for (uint32_t w = lo; w < hi; w++) {
        rank[w] = run;
        run += (uint32_t)__popcll(bits[w]);
    }
    if (tid == 1023) rank[n_words] = part[1023];
}
__global__ __launch_bounds__(256) void k_r_rehash(RetainMut m, uint32_t n_nodes) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_nodes) ov_rehash_one(m, i);
}
__global__ __launch_bounds__(BK) void k_r_topic_len(RetainMut m, const uint32_t* ids, uint32_t n, uint32_t* lens) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < n) ov_topic_len_one(m, ids, i, lens);
}
__global__ __launch_bounds__(BK) void k_r_topic_write(RetainMut m, const uint32_t* ids, uint32_t n, const unsigned long long* offs, uint8_t* out) {
    const uint32_t i = blockIdx.x * BK + threadIdx.x;
    if (i < n) ov_topic_write_one(m, ids, i, offs, out);
}
__global__ __launch_bounds__(256) void k_r_gc_flags(RetainMut m, GcQuery q, uint8_t* flags) {
    const uint32_t id = blockIdx.x * 256 + threadIdx.x;
    if (id < q.n_ids) flags[id] = (uint8_t)gc_flag_one(m, q, id);
}
__global__ __launch_bounds__(64) void k_r_find_tenant(RetainMut m, const uint8_t* name, uint32_t len, uint32_t* out) {
    if (threadIdx.x || blockIdx.x) return;
    LevelScan lv;
    unsigned long long pos = 0;
    lv.start = 0;
    scan_level_bytes<0u>(name, pos, len, lv.h, lv.inl, lv.len);
    out[0] = ov_find(m.onodes, m.oedges, m.oedge_mask, m.opool, 0u, lv.h.h1, lv.h.h2, lv.len, name, 0);
}

struct DevExec {
    std::string err;
    int device = -1;
    hipStream_t stream = nullptr;
    void* tmp = nullptr; // hipCUB temporary storage
    size_t tmp_cap = 0;

    bool ck(hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        err = std::string(what) + ": " + hipGetErrorString(e);
        return false;
    }
#define BMQ_X(expr) ck((expr), #expr)
    bool launched() { return BMQ_X(hipGetLastError()); }

    void* alloc(size_t bytes) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes + 64) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return p;
    }
    void release(void* p) {
        if (p) (void)hipFree(p);
    }
    ~DevExec() {
        release(tmp);
        if (pinned) (void)hipHostFree(pinned);
        if (ev_up) (void)hipEventDestroy(ev_up);
        if (ev_back) (void)hipEventDestroy(ev_back);
    }
    bool sync() { return BMQ_X(hipStreamSynchronize(stream)); }
    bool copy_in_async(void* d, const void* s, size_t n) { return n == 0 || BMQ_X(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, stream)); }
    // an apply batch's ops: uploaded on `upload_stream` (the engine's copy stream, if it has one) -- beside whatever `stream` still runs --,
    // `stream` waits for them (uploads_done); its counters come back into page-locked memory behind an event (read_back_async / _wait)
    hipStream_t upload_stream = nullptr;
    hipEvent_t ev_up = nullptr, ev_back = nullptr;
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    bool upload_async(void* d, const void* s, size_t n) {
        return n == 0 || BMQ_X(hipMemcpyAsync(d, s, n, hipMemcpyDefault, upload_stream ? upload_stream : stream)); // (s: host or device memory)
    }
    bool uploads_done() {
        if (!upload_stream) return true;
        if (!ev_up && !BMQ_X(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming))) return false;
        return BMQ_X(hipEventRecord(ev_up, upload_stream)) && BMQ_X(hipStreamWaitEvent(stream, ev_up, 0));
    }
    bool read_back_async(void* /*host*/, const void* dev, size_t n) {
        if (pinned_cap < n) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr, pinned_cap = 0;
            if (!BMQ_X(hipHostMalloc(&pinned, n, hipHostMallocDefault))) return false;
            pinned_cap = n;
        }
        if (!ev_back && !BMQ_X(hipEventCreateWithFlags(&ev_back, hipEventDisableTiming))) return false;
        return BMQ_X(hipMemcpyAsync(pinned, dev, n, hipMemcpyDeviceToHost, stream)) && BMQ_X(hipEventRecord(ev_back, stream));
    }
    bool read_back_wait(void* host, size_t n) {
        if (!BMQ_X(hipEventSynchronize(ev_back))) return false;
        memcpy(host, pinned, n);
        return true;
    }
    bool copy_in(void* d, const void* s, size_t n) { return copy_in_async(d, s, n) && sync(); }
    bool copy_out(void* d, const void* s, size_t n) { return (n == 0 || BMQ_X(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, stream))) && sync(); }
    bool copy(void* d, const void* s, size_t n) { return n == 0 || BMQ_X(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, stream)); }
    bool zero(void* p, size_t n) { return n == 0 || BMQ_X(hipMemsetAsync(p, 0, n, stream)); }
    static dim3 grid(unsigned long long n, unsigned b) { return dim3((unsigned)((n + b - 1) / b)); }

    bool fill_slots(TrieSlot* p, uint64_t n) {
        // at most 2^32 slots in the table, 256 per block
        for (uint64_t o = 0; o < n; o += (1ull << 31)) {
            const uint64_t m = std::min<uint64_t>(n - o, 1ull << 31);
            hipLaunchKernelGGL(k_b_fill_slots, grid(m, 256), dim3(256), 0, stream, p + o, (unsigned long long)m);
        }
        return launched();
    }
    bool prepare(const DistIndexMut& ix, const OpBatch& ob) {
        hipLaunchKernelGGL(k_b_prepare, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob);
        return launched();
    }
    static constexpr bool gated = true; // the stages of an apply batch can run back to back behind BuildCounters.gate
    bool gate(BuildCounters* bc, uint32_t stage) {
        hipLaunchKernelGGL(k_b_gate, dim3(1), dim3(1), 0, stream, bc, stage);
        return launched();
    }
    bool prepare_check(const DistIndexMut& ix, const OpBatch& ob, uint32_t n_dir) {
        hipLaunchKernelGGL(k_b_prepare_check, grid(n_dir, BK), dim3(BK), 0, stream, ix, ob, n_dir);
        return launched();
    }
    bool bulk_prepare(const DistIndexMut& ix, const OpBatch& ob) {
        hipLaunchKernelGGL(k_b_bulk_prepare, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob);
        return launched();
    }
    bool ensure_tmp(size_t bytes) {
        if (bytes <= tmp_cap) return true;
        if (!sync()) return false;
        release(tmp);
        tmp = alloc(bytes + bytes / 4);
        tmp_cap = tmp ? bytes + bytes / 4 : 0;
        if (!tmp) err = "out of device memory (sort scratch)";
        return tmp != nullptr;
    }
    bool scan_flags(const uint32_t* in, uint32_t* out, uint32_t n) { // inclusive
        size_t bytes = 0;
        if (!BMQ_X(hipcub::DeviceScan::InclusiveSum(nullptr, bytes, in, out, (int)n, stream))) return false;
        if (!ensure_tmp(bytes)) return false;
        return BMQ_X(hipcub::DeviceScan::InclusiveSum(tmp, bytes, in, out, (int)n, stream));
    }
    bool bulk_tenants(const DistIndexMut& ix, const OpBatch& ob, const uint32_t* scan) {
        hipLaunchKernelGGL(k_b_bulk_tenants, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob, scan);
        return launched();
    }
    bool bulk_counts(const OpBatch& ob, uint32_t n_ten, const uint32_t* nn_incl) {
        hipLaunchKernelGGL(k_b_bulk_counts, grid(n_ten, BK), dim3(BK), 0, stream, ob, n_ten, nn_incl);
        return launched();
    }
    bool locate(const DistIndexMut& ix, const OpBatch& ob) {
        if (!ob.op) {
            hipLaunchKernelGGL(k_b_locate, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob, 2u);
            return launched();
        }
        hipLaunchKernelGGL(k_b_locate, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob, 3u); // every op; a delete that finds no filter is marked
        hipLaunchKernelGGL(k_b_locate, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob, 4u); // the marked deletes, behind every put of the batch
        return launched();
    }
    bool sort_targets(const OpBatch& ob) { // stable: ops on one target stay in op order.  Values = op indices.
        uint32_t* iota = ob.unknown_list; // free again after prepare
        hipLaunchKernelGGL(k_b_iota, grid(ob.n, 256), dim3(256), 0, stream, iota, ob.n);
        if (!launched()) return false;
        size_t bytes = 0;
        const int end_bit = (int)TARGET_KIND_SHIFT + 2;
        if (!BMQ_X(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, ob.target, ob.sorted_target, iota, ob.order, (int)ob.n, 0, end_bit, stream)))
            return false;
        if (!ensure_tmp(bytes)) return false;
        return BMQ_X(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, ob.target, ob.sorted_target, iota, ob.order, (int)ob.n, 0, end_bit, stream));
    }
    bool group(const DistIndexMut& ix, const OpBatch& ob) {
        hipLaunchKernelGGL(k_b_group, grid(ob.n, BK), dim3(BK), 0, stream, ix, ob);
        return launched();
    }
    bool rehash(const DistIndexMut& ix, uint32_t old_base, uint32_t old_slots, uint32_t new_base, uint32_t new_buckets, uint32_t d) {
        hipLaunchKernelGGL(k_b_rehash, grid(old_slots, BK), dim3(BK), 0, stream, ix, old_base, old_slots, new_base, new_buckets, 0u, d);
        hipLaunchKernelGGL(k_b_rehash, grid(old_slots, BK), dim3(BK), 0, stream, ix, old_base, old_slots, new_base, new_buckets, 1u, d); // ('+' children beside their parents)
        return launched();
    }
    bool dict_rehash(const DictSlot* old, uint32_t old_slots, const DistIndexMut& ix) {
        hipLaunchKernelGGL(k_b_dict_rehash, grid(old_slots, BK), dim3(BK), 0, stream, old, old_slots, ix);
        return launched();
    }
    bool find(const DistIndexMut& ix, const uint8_t* q, uint32_t tenant_len, uint32_t filter_len, uint32_t* out, uint32_t cap) {
        hipLaunchKernelGGL(k_b_find, dim3(1), dim3(64), 0, stream, ix, q, tenant_len, filter_len, out, cap);
        return launched();
    }
    // ---- fan-out grouping (bmq_fanout.h) ----
    bool fill_bytes(void* p, int byte, size_t n) { return n == 0 || BMQ_X(hipMemsetAsync(p, byte, n, stream)); }
    bool fo_fill(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        hipLaunchKernelGGL(k_fo_fill, grid(b.total, 256), dim3(256), 0, stream, ix, st, b);
        return launched();
    }
    bool fo_verify(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        hipLaunchKernelGGL(k_fo_verify, grid(b.total, 256), dim3(256), 0, stream, ix, st, b);
        return launched();
    }
    bool fo_keys(const DistIndexMut& ix, const FanoutState& st, const FanoutBatch& b) {
        hipLaunchKernelGGL(k_fo_keys, grid(b.total, 256), dim3(256), 0, stream, ix, st, b);
        return launched();
    }
    bool sort_pairs32(const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in, uint32_t* vals_out, uint32_t n, int end_bit) {
        size_t bytes = 0; // LSD radix sort: stable
        if (!BMQ_X(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, stream))) return false;
        if (!ensure_tmp(bytes)) return false;
        return BMQ_X(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, stream));
    }
    bool fo_emit(const FanoutBatch& b) {
        hipLaunchKernelGGL(k_fo_emit, grid(b.total, 256), dim3(256), 0, stream, b);
        return launched();
    }
    bool fo_groups(const FanoutState& st, const FanoutBatch& b) {
        hipLaunchKernelGGL(k_fo_groups, grid(b.total, 256), dim3(256), 0, stream, st, b);
        return launched();
    }
    bool gather_refs(const DistIndexMut& ix, const uint32_t* ids, uint32_t n, uint32_t id_end, unsigned long long* out) {
        hipLaunchKernelGGL(k_b_gather_refs, grid(n, 256), dim3(256), 0, stream, ix, ids, n, id_end, out);
        return launched();
    }
    bool gather_bytes(const DistIndexMut& ix, const unsigned long long* refs, const uint64_t* offs, uint32_t n, uint8_t* out) {
        hipLaunchKernelGGL(k_b_gather_bytes, grid(n, BK), dim3(BK), 0, stream, ix, refs, offs, n, out);
        return launched();
    }
    // ---- fan-out grouping, fast path (bmq_fanout_kernels.h): counting sort that carries its payload ----
    static constexpr bool has_fanout_fast = true;
    bool fo_dense(const FanoutState& st, uint16_t* dense, uint32_t* n_used) {
        hipLaunchKernelGGL(k_fo_dense, dim3(1), dim3(1024), 0, stream, st.gt_hash, st.gt_cap, dense, n_used);
        return launched();
    }
    bool fo_fast(const DistIndexMut& ix, const FanoutState& st, const FanoutFast& f) {
        const dim3 grid((f.n_tiles + FO_WAVES - 1) / FO_WAVES), block(FO_WAVES * 64);
        static const bool timing = bmq_env("BMQ_TIMING") != nullptr; // profiling experiments only: per-kernel HIP-event times on stderr
        hipEvent_t ev[5] = {};
        if (timing)
            for (auto& e : ev) (void)hipEventCreate(&e);
        if (timing) (void)hipEventRecord(ev[0], stream);
        hipLaunchKernelGGL(k_fo_hist, grid, block, 0, stream, ix, st, f);
        if (!launched()) return false;
        if (timing) (void)hipEventRecord(ev[1], stream);
        const int n = (int)((size_t)f.n_bins * f.n_tiles) + 1; // + the end mark
        size_t bytes = 0;
        if (!BMQ_X(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, f.hist, f.hist, n, stream))) return false;
        if (!ensure_tmp(bytes)) return false;
        if (!BMQ_X(hipcub::DeviceScan::ExclusiveSum(tmp, bytes, f.hist, f.hist, n, stream))) return false;
        if (timing) (void)hipEventRecord(ev[2], stream);
        hipLaunchKernelGGL(k_fo_scatter, dim3((f.n_tiles + FO_SC_WAVES - 1) / FO_SC_WAVES), dim3(FO_SC_WAVES * 64), FO_SC_WAVES * fo_scatter_lds(f.n_bins, f.tile),
                           stream, f);
        if (timing) (void)hipEventRecord(ev[3], stream);
        hipLaunchKernelGGL(k_fo_groups2, dim3(1), dim3(1024), 0, stream, st, f);
        if (timing) {
            (void)hipEventRecord(ev[4], stream);
            (void)hipEventSynchronize(ev[4]);
            float t[4] = {0, 0, 0, 0};
            for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&t[i], ev[i], ev[i + 1]);
            fprintf(stderr, "[bmq] fan-out grouping (%u pairs, %u keys, %u tiles): hist %.3f ms, scan %.3f, scatter %.3f, groups %.3f\n", f.total, f.n_bins,
                    f.n_tiles, t[0], t[1], t[2], t[3]);
            for (auto& e : ev) (void)hipEventDestroy(e);
        }
        return launched();
    }
    // ---- retain direction (bmq_retain_core.h) ----
    bool r_locate(const RetainMut& m, const RetainOps& ob) {
        hipLaunchKernelGGL(k_r_locate, grid(ob.n, BK), dim3(BK), 0, stream, m, ob, 0u); // the adds: overlay nodes come into being
        hipLaunchKernelGGL(k_r_locate, grid(ob.n, BK), dim3(BK), 0, stream, m, ob, 1u); // the removes: they find them
        return launched();
    }
    bool r_commit(const RetainMut& m, const RetainOps& ob) {
        hipLaunchKernelGGL(k_r_commit, grid(ob.n, BK), dim3(BK), 0, stream, m, ob);
        return launched();
    }
    bool r_rank(const RetainMut& m, uint32_t n_words) {
        hipLaunchKernelGGL(k_r_rank, dim3(1), dim3(1024), 0, stream, m.dead_bits, m.dead_rank, n_words);
        return launched();
    }
    bool r_rehash(const RetainMut& m, uint32_t n_nodes) {
        hipLaunchKernelGGL(k_r_rehash, grid(n_nodes, 256), dim3(256), 0, stream, m, n_nodes);
        return launched();
    }
    bool r_topic_lens(const RetainMut& m, const uint32_t* ids, uint32_t n, uint32_t* lens) {
        hipLaunchKernelGGL(k_r_topic_len, grid(n, BK), dim3(BK), 0, stream, m, ids, n, lens);
        return launched();
    }
    bool r_topic_write(const RetainMut& m, const uint32_t* ids, uint32_t n, const unsigned long long* offs, uint8_t* out) {
        hipLaunchKernelGGL(k_r_topic_write, grid(n, BK), dim3(BK), 0, stream, m, ids, n, offs, out);
        return launched();
    }
    // ids (ascending) gc_flag_one accepts -> out_ids[0 .. *out_count); flags: scratch of q.n_ids bytes
    bool r_gc_select(const RetainMut& m, const GcQuery& q, uint8_t* flags, uint32_t* out_ids, uint32_t* out_count) {
        if (q.n_ids == 0) return zero(out_count, sizeof(uint32_t));
        hipLaunchKernelGGL(k_r_gc_flags, grid(q.n_ids, 256), dim3(256), 0, stream, m, q, flags);
        if (!launched()) return false;
        hipcub::CountingInputIterator<uint32_t> iota(0u);
        size_t bytes = 0;
        if (!BMQ_X(hipcub::DeviceSelect::Flagged(nullptr, bytes, iota, flags, out_ids, out_count, (int)q.n_ids, stream))) return false;
        if (!ensure_tmp(bytes)) return false;
        return BMQ_X(hipcub::DeviceSelect::Flagged(tmp, bytes, iota, flags, out_ids, out_count, (int)q.n_ids, stream));
    }
    bool r_find_tenant(const RetainMut& m, const uint8_t* name, uint32_t len, uint32_t* out) {
        hipLaunchKernelGGL(k_r_find_tenant, dim3(1), dim3(64), 0, stream, m, name, len, out);
        return launched();
    }
#undef BMQ_X
};

} // namespace bmq
