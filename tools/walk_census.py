#!/usr/bin/env python3
"""How many trie nodes a publish topic discovers on the C3 workload, and how many of them a different index layout would not have to
fetch (planning tool, CPU only: a dictionary-of-dictionaries filter trie over a few tenants of bench.py's C3 workload, walked with the
engine's matching rule).  The engine reports N_visit = 13.65 per topic for this workload (`visits_per_topic` of the bench line): the
model below has to reproduce that before its other columns mean anything.
    python tools/walk_census.py [n_tenants=16] [n_topics=100000]"""
import sys
from collections import Counter

import numpy as np

sys.path.insert(0, ".")
import bifromq_amd as B  # noqa: E402
from bifromq_amd.workload import MODE_MIXED, unpack  # noqa: E402


class Node:
    __slots__ = ("kids", "own", "hash", "unary_tail")

    def __init__(self):
        self.kids, self.own, self.hash, self.unary_tail = {}, 0, 0, False


def main():
    n_tenants = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    n_topics = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    w = B.Workload(0xB1F20003, 1000, 10000, MODE_MIXED)   # bench.py's C3 population
    tn = w.tenants()
    first = w.tenant_first()
    keys = w.keys()
    roots = {}
    n_nodes = 0
    for t in range(n_tenants):
        root = roots[tn[t]] = Node()
        for k in keys[first[t]:first[t + 1]]:
            flag, tenant, filt, _ = B.decode_route_key(k)
            levels = filt.split("/")
            node = root
            ends_hash = levels[-1] == "#"
            for lv in (levels[:-1] if ends_hash else levels):
                nxt = node.kids.get(lv)
                if nxt is None:
                    nxt = node.kids[lv] = Node()
                    n_nodes += 1
                node = nxt
            if ends_hash:
                node.hash += 1
            else:
                node.own += 1

    # a node is the head of a UNARY TAIL when it and everything below it has exactly one child and routes only at the very end: a compressed
    # layout stores such a tail in one record (suffix hash) -- a topic then pays one fetch for the whole tail instead of one per level
    def mark(node):
        ok = True
        for c in node.kids.values():
            ok = mark(c) and ok
        node.unary_tail = (len(node.kids) == 0) or (len(node.kids) == 1 and ok and node.own == 0 and node.hash == 0 and next(iter(node.kids.values())).unary_tail)
        return node.unary_tail

    for r in roots.values():
        mark(r)
    data, off, tt = w.topics(11, n_topics, 0, n_tenants, 900, True)
    topics = unpack(data, off)
    visits = Counter()
    tot = Counter()
    for i, tp in enumerate(topics):
        levels = tp.decode().split("/")
        sys_topic = levels[0].startswith("$")
        root = roots[tn[tt[i]]]
        # (node, level index, on the all-literal path, inside a unary tail whose head was already fetched)
        stack = [(root, 0, True, False)]
        v = lit = comp = 0
        while stack:
            node, li, literal, in_tail = stack.pop()
            if li == len(levels):
                continue
            for lab, lit_edge in ((levels[li], True), ("+", False)):
                if not lit_edge and sys_topic and li == 0:
                    continue  # '+' and '#' do not match a $-topic's first level
                c = node.kids.get(lab)
                if c is not None:
                    v += 1
                    if literal and lit_edge:
                        lit += 1
                    if in_tail:
                        comp += 1  # fetched today, part of its parent's record in a compressed layout
                    stack.append((c, li + 1, literal and lit_edge, in_tail or c.unary_tail))
        tot["visits"] += v
        tot["literal_path"] += lit
        tot["inside_unary_tail"] += comp
        tot["levels"] += len(levels)
        visits[v] += 1
    n = len(topics)
    print("tenants %d (of 1000), trie nodes %d, topics %d" % (n_tenants, n_nodes, n))
    print("nodes discovered per topic      %.2f   (engine, whole C3 population: 13.65)" % (tot["visits"] / n))
    print("  on the all-literal path       %.2f" % (tot["literal_path"] / n))
    print("  inside a unary tail           %.2f   (a path-compressed layout fetches the tail's head only)" % (tot["inside_unary_tail"] / n))
    print("levels per topic                %.2f   (= dictionary look-ups per topic today; hashed edge keys need none)" % (tot["levels"] / n))
    print("line requests per topic today ~ %.1f (visits + levels + topic bytes); compressed tails + hashed edges ~ %.1f" % (
        (tot["visits"] + tot["levels"]) / n, (tot["visits"] - tot["inside_unary_tail"]) / n))
    qs = np.percentile(np.repeat(list(visits.keys()), list(visits.values())), [50, 90, 99, 100])
    print("visits per topic: p50 %d  p90 %d  p99 %d  max %d" % tuple(qs))


if __name__ == "__main__":
    main()
