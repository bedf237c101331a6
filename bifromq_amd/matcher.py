"""Host-side mirror of the reference's operator interface for the dist match path.

`TenantRouteMatcher.match_all(topics, max_persistent_fanout, max_group_fanout)` has the argument meaning and
result shape of ITenantRouteMatcher.matchAll (DW/cache/ITenantRouteMatcher.java:37): one entry per input topic
(also for topics without routes), each a `MatchedRoutes` with the accepted routes and the fan-out counters of
DW/cache/MatchedRoutes.java:60-83.  The work is done by bmq_match_all() in libbmq.so (HIP kernels + the C++
cap logic); this file only shapes the result.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Tuple

from .engine import INT_MAX, Engine, decode_route_key


@dataclass
class MatchedRoutes:
    tenant_id: str
    topic: str
    max_persistent_fanout: int
    max_group_fanout: int
    route_ids: List[int] = field(default_factory=list)     # accepted routes, ascending id (the caps were applied in KV key order)
    throttled: List[Tuple[int, int]] = field(default_factory=list)  # (event type, rejected route id)
    persistent_fanout: int = 0
    group_fanout: int = 0

    def routes(self, engine: Engine):
        """-> [(flag, tenant, mqttTopicFilter, receiver)] like Set<Matching> (SCHEMA/cache/Matching.java:29-56)"""
        return [decode_route_key(engine.route_key(i)) for i in self.route_ids]


class TenantRouteMatcher:
    """One per (range, tenant), like TenantRouteCacheFactory.create (DW/cache/TenantRouteCacheFactory.java:67-71)."""

    def __init__(self, engine: Engine, tenant_id: str):
        self.engine = engine
        self.tenant_id = tenant_id

    def match_all(self, topics: Iterable[str], max_persistent_fanout: int = INT_MAX,
                  max_group_fanout: int = INT_MAX) -> Dict[str, MatchedRoutes]:
        uniq = list(dict.fromkeys(topics))  # Set<String> semantics
        rows, events = self.engine.match_all(self.tenant_id, uniq, max_persistent_fanout, max_group_fanout)
        out: Dict[str, MatchedRoutes] = {}
        for i, t in enumerate(uniq):
            mr = MatchedRoutes(self.tenant_id, t, max_persistent_fanout, max_group_fanout, rows[i])
            for rid in rows[i]:
                flag, _, _, recv = decode_route_key(self.engine.route_key(rid))
                if flag == 1:
                    mr.persistent_fanout += recv.split("\0", 1)[0] == "1"
                else:
                    mr.group_fanout += 1
            out[t] = mr
        for typ, ti, rid, _ in events:
            out[uniq[ti]].throttled.append((typ, rid))
        return out
