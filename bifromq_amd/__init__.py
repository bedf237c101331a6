"""bifromq_amd -- MI355X-native MQTT topic-filter match engine (drop-in for apache/bifromq's dist/retain match path).

The product is the C-ABI library bifromq_amd/libbmq.so (include/bmq.h) built from bifromq_amd/csrc/ for gfx950.
This package is the Python face used by tests and bench.py.
"""
from . import _lib
from .engine import Batcher, BmqError, Engine, RangeRouter, RouteCache, INT_MAX, decode_route_key, java_string_hash, pack, route_key, route_key_from_mqtt
from .generations import GenerationalEngine
from .matcher import MatchedRoutes, TenantRouteMatcher
from .workload import Workload

__all__ = ["Engine", "GenerationalEngine", "Batcher", "RouteCache", "RangeRouter", "BmqError", "TenantRouteMatcher", "MatchedRoutes", "Workload", "pack", "route_key",
           "route_key_from_mqtt", "decode_route_key", "java_string_hash", "INT_MAX"]
