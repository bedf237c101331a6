"""ctypes loader of libbmq.so (the C-ABI engine, include/bmq.h) and libbmq_gen.so (workload generator).

The libraries are built IN-TREE by `make -C bifromq_amd/csrc all` (hipcc --offload-arch=gfx950); see
__graft_entry__.build().  Loading fails loudly if the HIP library is missing -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BMQ_LIB") or os.path.join(_HERE, "libbmq.so")  # BMQ_LIB: build-variant experiments only
GEN_PATH = os.path.join(_HERE, "libbmq_gen.so")
CSRC = os.path.join(_HERE, "csrc")

# every symbol include/bmq.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    "bmq_engine_create", "bmq_engine_destroy", "bmq_last_error", "bmq_version", "bmq_rebuild", "bmq_compact", "bmq_compact_begin", "bmq_compact_poll", "bmq_compact_swap", "bmq_compact_abort", "bmq_routes_apply", "bmq_routes_apply_async", "bmq_routes_apply_wait",
    "bmq_index_info_get", "bmq_route_key", "bmq_route_keys", "bmq_index_find", "bmq_match_batch", "bmq_match_batch_dev",
    "bmq_match_finish", "bmq_set_kernel_timing", "bmq_match_submit", "bmq_match_wait", "bmq_match_submit_fmt", "bmq_match_submit_dev", "bmq_match_wait_dev", "bmq_match_wait_counts", "bmq_match_wait_ranges", "bmq_match_wait_grouped", "bmq_host_alloc", "bmq_host_free", "bmq_sync", "bmq_stats_get", "bmq_stream", "bmq_match_all", "bmq_route_key_encode",
    "bmq_route_key_decode", "bmq_java_string_hash", "bmq_range_lookup", "bmq_comm_unique_id", "bmq_comm_init", "bmq_comm_destroy", "bmq_exchange_fanout",
    "bmq_exchange_csr", "bmq_exchange_wait", "bmq_partition_batch_dev", "bmq_retain_message_key", "bmq_retain_filter_route", "bmq_retain_rebuild", "bmq_retain_rebuild_ex", "bmq_retain_apply", "bmq_retain_apply_ex", "bmq_retain_topic",
    "bmq_retain_topic_info", "bmq_retain_find_all", "bmq_retain_expired", "bmq_retain_apply_batch", "bmq_retain_compact", "bmq_retain_compact_begin", "bmq_retain_compact_build", "bmq_retain_compact_swap", "bmq_retain_compact_abort", "bmq_retain_info_get",
    "bmq_retain_live_ids", "bmq_retain_topics",
    "bmq_retain_match_batch", "bmq_retain_match_batch_dev", "bmq_retain_match_limited", "bmq_batcher_create", "bmq_batcher_destroy",
    "bmq_batcher_match_all", "bmq_batcher_submit", "bmq_batcher_stats_get", "bmq_poller_stats_get", "bmq_poller_control",
    "bmq_route_cache_create", "bmq_route_cache_destroy", "bmq_route_cache_get", "bmq_route_cache_get_async", "bmq_route_cache_get_batch", "bmq_batcher_match_batch", "bmq_route_cache_is_cached", "bmq_route_cache_apply",
    "bmq_route_cache_rebuild", "bmq_route_cache_reset", "bmq_route_cache_expire", "bmq_route_cache_stats_get", "bmq_route_cache_tenant_stats_get",
    "bmq_route_cache_set_caps", "bmq_route_cache_set_event_sink", "bmq_routes_cap", "bmq_fanout_group", "bmq_fanout_group_dev", "bmq_router_find_by_key", "bmq_router_find_by_boundary", "bmq_retain_range_lookup", "bmq_router_create", "bmq_router_destroy", "bmq_router_lookup_key", "bmq_router_lookup_boundary", "bmq_router_retain_lookup",
]


def build(force: bool = False) -> None:
    """(Re)build the in-tree shared libraries when sources are newer than the binaries."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h", ".inc"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "bmq.h"))
    newest = max(os.path.getmtime(s) for s in srcs)
    stale = force or any(not os.path.exists(p) or os.path.getmtime(p) < newest for p in (LIB_PATH, GEN_PATH))
    if stale:
        cmd = ["make", "-C", CSRC, "all"] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL)


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("wave_queue_cap", C.c_uint32),
                ("wave_pair_cap", C.c_uint32), ("slow_scratch_mb", C.c_uint32), ("kernel_timing", C.c_uint32),
                ("dedup_min_topics", C.c_uint32), ("dedup_sorted", C.c_uint32), ("region_slack", C.c_uint32), ("reserved", C.c_uint32 * 4)]


class Stats(C.Structure):
    _fields_ = [("n_topics", C.c_uint64), ("n_visit", C.c_uint64), ("n_match", C.c_uint64), ("n_ranges", C.c_uint64),
                ("n_slow_topics", C.c_uint64), ("n_sorted_rows", C.c_uint64), ("topic_bytes", C.c_uint64),
                ("ms_total", C.c_float), ("ms_walk", C.c_float), ("ms_expand", C.c_float), ("n_walked", C.c_uint32),
                ("n_split_blocks", C.c_uint32), ("reserved0", C.c_uint32)]


class IndexInfo(C.Structure):
    _fields_ = [("n_routes", C.c_uint64), ("n_tenants", C.c_uint64), ("n_nodes", C.c_uint64), ("n_tokens", C.c_uint64),
                ("trie_slots", C.c_uint64), ("dict_slots", C.c_uint64), ("device_bytes", C.c_uint64),
                ("epoch", C.c_uint64), ("generation", C.c_uint64), ("next_route_id", C.c_uint64),
                ("garbage_bytes", C.c_uint64)]


class BatcherConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_batch_topics", C.c_uint32), ("reserved", C.c_uint32 * 6)]


class BatcherStats(C.Structure):
    _fields_ = [("n_requests", C.c_uint64), ("n_topics", C.c_uint64), ("n_batches", C.c_uint64),
                ("max_batch_topics", C.c_uint64), ("n_deduped", C.c_uint64)]


class PollerStats(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("running", C.c_uint32), ("n_starts", C.c_uint64), ("n_served", C.c_uint64), ("n_fallback", C.c_uint64),
                ("n_unserved", C.c_uint64), ("n_timeouts", C.c_uint64), ("n_bad_input", C.c_uint64)]


class RouteCacheConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("mutation_log_entries", C.c_uint32), ("max_routes_per_tenant", C.c_uint64), ("expiry_ms", C.c_uint64),
                ("shards_per_tenant", C.c_uint64), ("direct_batch_topics", C.c_uint64), ("default_max_persistent_fanout", C.c_int32),
                ("default_max_group_fanout", C.c_int32), ("tenant_idle_ms", C.c_uint64)]


class RouteCacheStats(C.Structure):
    _fields_ = [("hits", C.c_uint64), ("misses", C.c_uint64), ("evictions", C.c_uint64), ("invalidations", C.c_uint64), ("expired", C.c_uint64),
                ("stale_loads", C.c_uint64), ("entries", C.c_uint64), ("cached_routes", C.c_uint64), ("tenants", C.c_uint64),
                ("tenants_expired", C.c_uint64)]


class RetainInfo(C.Structure):
    _fields_ = [("n_topics", C.c_uint64), ("n_tenants", C.c_uint64), ("id_bound", C.c_uint64), ("loaded_topics", C.c_uint64),
                ("loaded_removed", C.c_uint64), ("added_ids", C.c_uint64), ("overlay_nodes", C.c_uint64), ("epoch", C.c_uint64),
                ("generation", C.c_uint64)]


class RangesInfo(C.Structure):
    _fields_ = [("n_ranges", C.c_uint64), ("n_side_ids", C.c_uint64), ("n_ids", C.c_uint64), ("n_overlapping_rows", C.c_uint64)]


class RouteCacheTenantStats(C.Structure):
    _fields_ = [("hits", C.c_uint64), ("misses", C.c_uint64), ("evictions", C.c_uint64), ("entries", C.c_uint64), ("cached_routes", C.c_uint64),
                ("last_get_ms", C.c_uint64), ("max_persistent_fanout", C.c_int32), ("max_group_fanout", C.c_int32)]


_lib = None
_gen = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
        P = C.POINTER
        sig = {
            "bmq_engine_create": (C.c_int, [P(Config), P(vp)]),
            "bmq_engine_destroy": (None, [vp]),
            "bmq_last_error": (C.c_char_p, [vp]),
            "bmq_version": (C.c_char_p, []),
            "bmq_rebuild": (C.c_int, [vp, vp, vp, u32]),
            "bmq_retain_compact_begin": (C.c_int, [vp]),
            "bmq_retain_compact_build": (C.c_int, [vp]),
            "bmq_retain_compact_swap": (C.c_int, [vp, P(u64), P(u64)]),
            "bmq_retain_compact_abort": (C.c_int, [vp]),
            "bmq_compact_begin": (C.c_int, [vp]),
            "bmq_compact_poll": (C.c_int, [vp, u32, P(u32)]),
            "bmq_compact_swap": (C.c_int, [vp, P(u64), P(u64)]),
            "bmq_compact_abort": (C.c_int, [vp]),
            "bmq_routes_apply": (C.c_int, [vp, vp, vp, vp, u32]),
            "bmq_routes_apply_async": (C.c_int, [vp, vp, vp, vp, u32]),
            "bmq_routes_apply_wait": (C.c_int, [vp]),
            "bmq_compact": (C.c_int, [vp]),
            "bmq_index_info_get": (C.c_int, [vp, P(IndexInfo)]),
            "bmq_route_key": (C.c_int, [vp, u32, C.c_char_p, u32, P(u32)]),
            "bmq_route_keys": (C.c_int, [vp, vp, u32, vp, u64, vp]),
            "bmq_index_find": (C.c_int, [vp, C.c_char_p, u32, C.c_char_p, u32, vp, u32, P(u32)]),
            "bmq_match_batch": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, P(u64)]),
            "bmq_match_batch_dev": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, vp]),
            "bmq_match_finish": (C.c_int, [vp, P(u64)]),
            "bmq_set_kernel_timing": (C.c_int, [vp, C.c_int]),
            "bmq_match_submit": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, P(C.c_int)]),
            "bmq_match_wait": (C.c_int, [vp, C.c_int, vp, vp, u64, P(u64)]),
            "bmq_match_submit_fmt": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, C.c_int, P(C.c_int)]),
            "bmq_match_wait_counts": (C.c_int, [vp, C.c_int, vp, P(u64)]),
            "bmq_match_submit_dev": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, vp, P(C.c_int)]),
            "bmq_match_wait_dev": (C.c_int, [vp, C.c_int, P(u64)]),
            "bmq_match_wait_ranges": (C.c_int, [vp, C.c_int, vp, vp, vp, u64, vp, u64, P(RangesInfo)]),
            "bmq_match_wait_grouped": (C.c_int, [vp, C.c_int, vp, vp, u64, vp, vp, u32, P(u32), P(u32), P(u64)]),
            "bmq_host_alloc": (vp, [C.c_size_t]),
            "bmq_host_free": (None, [vp]),
            "bmq_sync": (C.c_int, [vp]),
            "bmq_stats_get": (C.c_int, [vp, P(Stats)]),
            "bmq_stream": (vp, [vp]),
            "bmq_match_all": (C.c_int, [vp, C.c_char_p, u32, vp, vp, u32, i32, i32, vp, vp, u64, P(u64), vp, u32, P(u32)]),
            "bmq_route_key_encode": (u32, [C.c_char_p, u32, C.c_char_p, u32, C.c_uint8, C.c_char_p, u32, C.c_char_p, u32]),
            "bmq_route_key_decode": (C.c_int, [C.c_char_p, u32, P(u32)]),
            "bmq_java_string_hash": (i32, [C.c_char_p, u32]),
            "bmq_comm_unique_id": (C.c_int, [C.c_char_p]),
            "bmq_comm_init": (C.c_int, [vp, C.c_int, C.c_int, C.c_char_p]),
            "bmq_comm_destroy": (None, [vp]),
            "bmq_exchange_fanout": (C.c_int, [vp, vp, u32, vp]),
            "bmq_exchange_csr": (C.c_int, [vp, vp, vp, u32, u64, vp, vp, u64, vp]),
            "bmq_exchange_wait": (C.c_int, [vp]),
            "bmq_partition_batch_dev": (C.c_int, [vp, vp, u32, i32, vp, vp, vp, u32, vp, vp, vp, vp, P(u32), P(u64)]),
            "bmq_retain_message_key": (u32, [C.c_char_p, u32, C.c_char_p, u32, C.c_char_p, u32]),
            "bmq_retain_filter_route": (C.c_int, [C.c_char_p, u32, C.c_char_p, u32, C.c_char_p, u32, P(u32), C.c_char_p, u32, P(u32), P(u32)]),
            "bmq_range_lookup": (C.c_int, [vp, C.c_char_p, u32, vp, vp, u32, vp, vp, vp, vp, vp, u32, vp]),
            "bmq_retain_rebuild": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32]),
            "bmq_retain_apply": (C.c_int, [vp, C.c_char_p, u32, vp, vp, vp, u32]),
            "bmq_retain_rebuild_ex": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp]),
            "bmq_retain_apply_ex": (C.c_int, [vp, C.c_char_p, u32, vp, vp, vp, vp, vp, u32]),
            "bmq_retain_topic_info": (C.c_int, [vp, u32, P(u64), P(u32), P(u64)]),
            "bmq_retain_find_all": (C.c_int, [vp, P(u64), P(u64)]),
            "bmq_retain_expired": (C.c_int, [vp, C.c_char_p, u32, u64, C.c_int64, vp, u32, P(u32)]),
            "bmq_retain_topic": (C.c_int, [vp, u32, C.c_char_p, u32, P(u32), P(u32)]),
            "bmq_retain_topics": (C.c_int, [vp, vp, u32, vp, u64, vp, vp]),
            "bmq_retain_apply_batch": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, vp, vp, vp, u32, vp]),
            "bmq_retain_compact": (C.c_int, [vp]),
            "bmq_retain_info_get": (C.c_int, [vp, P(RetainInfo)]),
            "bmq_retain_live_ids": (C.c_int, [vp, C.c_char_p, u32, vp, u32, P(u32)]),
            "bmq_retain_match_batch": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, P(u64)]),
            "bmq_retain_match_batch_dev": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, vp]),
            "bmq_retain_match_limited": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, u64, vp, vp, u64, P(u64), vp]),
            "bmq_batcher_create": (C.c_int, [vp, P(BatcherConfig), P(vp)]),
            "bmq_batcher_destroy": (None, [vp]),
            "bmq_batcher_match_all": (C.c_int, [vp, C.c_char_p, u32, vp, vp, u32, vp, vp, u64, P(u64), P(u64)]),
            "bmq_batcher_submit": (C.c_int, [vp, C.c_char_p, u32, C.c_char_p, u32, vp, vp]),
            "bmq_batcher_stats_get": (C.c_int, [vp, P(BatcherStats)]),
            "bmq_poller_stats_get": (C.c_int, [vp, P(PollerStats)]),
            "bmq_poller_control": (C.c_int, [vp, C.c_int]),
            "bmq_route_cache_create": (C.c_int, [vp, vp, P(RouteCacheConfig), P(vp)]),
            "bmq_route_cache_destroy": (None, [vp]),
            "bmq_route_cache_get": (C.c_int, [vp, C.c_char_p, u32, C.c_char_p, u32, u64, vp, u32, P(u32), P(u64)]),
            "bmq_batcher_match_batch": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, vp, vp, u64, P(u64), P(u64)]),
            "bmq_route_cache_get_batch": (C.c_int, [vp, vp, vp, u32, vp, vp, vp, u32, u64, vp, vp, u64, P(u64), vp]),
            "bmq_route_cache_get_async": (C.c_int, [vp, C.c_char_p, u32, C.c_char_p, u32, u64, vp, vp]),
            "bmq_route_cache_is_cached": (C.c_int, [vp, C.c_char_p, u32, C.c_char_p, u32]),
            "bmq_route_cache_apply": (C.c_int, [vp, vp, vp, vp, u32]),
            "bmq_route_cache_rebuild": (C.c_int, [vp, vp, vp, u32]),
            "bmq_route_cache_reset": (C.c_int, [vp]),
            "bmq_route_cache_expire": (C.c_int, [vp, u64, P(u64)]),
            "bmq_route_cache_stats_get": (C.c_int, [vp, P(RouteCacheStats)]),
            "bmq_route_cache_tenant_stats_get": (C.c_int, [vp, C.c_char_p, u32, P(RouteCacheTenantStats)]),
            "bmq_route_cache_set_caps": (C.c_int, [vp, C.c_char_p, u32, i32, i32]),
            "bmq_route_cache_set_event_sink": (C.c_int, [vp, vp, vp]),
            "bmq_routes_cap": (C.c_int, [vp, vp, vp, u32, i32, i32, vp, vp, vp, vp, u32, P(u32)]),
            "bmq_fanout_group": (C.c_int, [vp, vp, vp, u32, vp, vp, u64, vp, vp, u32, P(u32), P(u32)]),
            "bmq_fanout_group_dev": (C.c_int, [vp, vp, vp, u32, u64, vp, vp, vp, vp, u32, P(u32), P(u32)]),
            "bmq_router_find_by_key": (C.c_int, [vp, vp, vp, vp, vp, u32, C.c_char_p, u32, P(i32)]),
            "bmq_router_find_by_boundary": (C.c_int, [vp, vp, vp, vp, vp, u32, C.c_uint8, C.c_char_p, u32, C.c_char_p, u32, P(u32), P(u32)]),
            "bmq_retain_range_lookup": (C.c_int, [C.c_char_p, u32, vp, vp, u32, vp, vp, vp, vp, vp, u32, u32, vp]),
            "bmq_router_create": (C.c_int, [vp, vp, vp, vp, vp, u32, P(vp)]),
            "bmq_router_destroy": (None, [vp]),
            "bmq_router_lookup_key": (C.c_int, [vp, C.c_char_p, u32, P(i32)]),
            "bmq_router_lookup_boundary": (C.c_int, [vp, C.c_uint8, C.c_char_p, u32, C.c_char_p, u32, P(u32), P(u32)]),
            "bmq_router_retain_lookup": (C.c_int, [vp, C.c_char_p, u32, vp, vp, u32, u32, vp]),
        }
        assert sorted(sig) == sorted(ABI_SYMBOLS)
        for name, (res, args) in sig.items():
            fn = getattr(L, name)  # AttributeError here == symbol missing from the .so
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def gen() -> C.CDLL:
    global _gen
    if _gen is None:
        if not os.path.exists(GEN_PATH):
            raise RuntimeError(f"{GEN_PATH} is missing: run __graft_entry__.build()")
        G = C.CDLL(GEN_PATH)
        vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
        sig = {
            "bmqgen_create": (vp, [u64, u32, u32, u32, C.c_int]), "bmqgen_destroy": (None, [vp]),
            "bmqgen_create_list": (vp, [u64, vp, u32, u32, C.c_int]),
            "bmqgen_n_keys": (u32, [vp]), "bmqgen_key_bytes": (vp, [vp]), "bmqgen_key_off": (vp, [vp]),
            "bmqgen_n_tenants": (u32, [vp]), "bmqgen_tenant_bytes": (vp, [vp]), "bmqgen_tenant_off": (vp, [vp]),
            "bmqgen_tenant_first": (vp, [vp]),
            "bmqgen_topics": (u32, [vp, u64, u32, u32, u32, u32, C.c_int]),
            "bmqgen_topic_bytes": (vp, [vp]), "bmqgen_topic_off": (vp, [vp]), "bmqgen_topic_tenant": (vp, [vp]),
            "bmqgen_retain": (u32, [vp, u64, u32, C.c_int]),
            "bmqgen_drive_singletons": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, vp, u32, u32, vp, vp, C.POINTER(C.c_double)]),
            "bmqgen_row_hash": (u64, [vp, u64]),
            "bmqgen_drive_cache": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, vp, u32, u32, u32, vp, vp, C.POINTER(C.c_double)]),
            "bmqgen_drive_cache_async": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, vp, u32, u32, u32, vp, vp, C.POINTER(C.c_double)]),
            "bmqgen_drive_async": (C.c_int, [vp, vp, vp, vp, u32, vp, vp, vp, u32, u32, vp, vp, C.POINTER(C.c_double)]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(G, name)
            fn.restype = res
            fn.argtypes = args
        _gen = G
    return _gen
