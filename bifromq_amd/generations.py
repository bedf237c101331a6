"""Compaction that does not stop the world -- on the CALLER's side of the C-ABI, with two engine handles.

(Round 5: the engine does this itself, inside one handle and without shipping the keys through the host -- bmq_compact_begin / _poll / _swap,
Engine.compact_begin / compact_poll / compact_swap.  This module stays as the variant that needs nothing below the boundary, and for a caller
that wants the old generation to keep answering for its own ids until the last reader has left.)

What it answers: TopicLevelTrie contracts tombed nodes as it goes (bifromq-util/.../index/TopicLevelTrie.java:257-384); here dead ids,
abandoned regions and id lists only grow until `bmq_compact`, which rebuilds the index inside the engine and holds every entry point for the
length of a bulk load.  The same rebuild can run BESIDE the serving index instead:

    generation A serves (match batches, bmq_routes_apply) ............................................... retired, closed when unpinned
         |  1. start logging the ops applied to A            3. replay the log into B (A keeps serving;      ^
         |  2. export A's live route keys (bmq_route_keys),     rounds until the log is empty, the last      | 4. swap: matches and
         |     build generation B from them (bmq_rebuild on     one under the lock: a stall of ONE small     |    mutations go to B
         |     B's own stream, its own buffers)                 apply, not of a bulk load)                   |

Nothing new is needed below the boundary: `bmq_engine_create`, `bmq_route_keys`, `bmq_rebuild`, `bmq_routes_apply` (include/bmq.h).  Route ids
are renumbered by the swap exactly as `bmq_compact` renumbers them (they are dense ranks again; `generation` counts the swaps), so whatever
a caller keeps per route id it re-derives per generation -- `pin()` tells it which generation a result came from.  Objects created ON an
engine handle (the batching front, the route cache) belong to one generation and are re-created on the new one by their owner.  The wrapper
carries the ROUTE index (the dist direction); a handle that also holds retained topics is refused (the reference keeps the two in separate
coprocs, and `bmq_retain_compact` rebuilds 1 M retained topics in a few milliseconds).
The price: two copies of the index in HBM while B is built (the C3 index is 4 GB of 288), and B's builder kernels share the GPU with A's
batches for the length of the bulk load.  The JVM adapter does the same with two `NativeMatcher` handles."""
from __future__ import annotations

import threading
from contextlib import contextmanager
from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Engine


class _Gen:
    def __init__(self, eng: Engine, number: int):
        self.eng, self.number, self.pins, self.retired = eng, number, 0, False


class GenerationalEngine:
    """An Engine that can be compacted while it serves.  Matching goes through `pin()`:

        with g.pin() as (eng, generation):
            row, ids = eng.match_batch(...)        # ids are route ids OF THIS generation: resolve them through `eng`

    Mutations go through `apply()` (they reach the serving generation and, while a compaction runs, its log)."""

    EXPORT_CHUNK = 1 << 20  # route ids per bmq_route_keys call
    MAX_REPLAY_ROUNDS = 8   # replay rounds outside the lock before the last one is taken under it

    def __init__(self, **engine_args):
        self._args = engine_args
        self._lock = threading.Lock()            # guards _cur, _log, pins
        self._compacting = threading.Lock()      # one compaction at a time
        self._cur = _Gen(Engine(**engine_args), 0)
        self._log: Optional[List[Tuple[int, bytes]]] = None
        self._closed = False                     # close() was called: a compaction still running must not install its generation
        self._log_broken = False                 # an apply() failed after it was logged: the running compaction is abandoned
        self.hooks = {}                          # test hooks: name -> callable, run at the named point of compact_online()

    # ---- serving ----
    @contextmanager
    def pin(self):
        with self._lock:
            g = self._cur
            g.pins += 1
        try:
            yield g.eng, g.number
        finally:
            with self._lock:
                g.pins -= 1
                close = g.retired and g.pins == 0
            if close:
                g.eng.close()

    @property
    def generation(self) -> int:
        return self._cur.number

    def rebuild(self, keys: Iterable[bytes] = (), packed=None):
        with self._lock:
            if self._log is not None:
                raise RuntimeError("rebuild while a compaction is running")
            self._cur.eng.rebuild(keys, packed)
        return self

    def apply(self, ops: Sequence[Tuple[int, bytes]]):
        """(0 = put | 1 = delete, route key)*: to the serving generation, in order, and to the log of a running compaction"""
        ops = list(ops)
        with self._lock:
            if self._log is not None:
                self._log.extend(ops)  # logged FIRST: a batch the serving generation applied in part must not be missing from the next one
            try:
                self._cur.eng.apply(ops)
            except Exception:
                if self._log is not None:
                    self._log_broken = True  # what exactly reached the serving generation is unknown: the compaction gives up
                raise
        return self

    def close(self):
        with self._lock:
            self._closed = True
            g = self._cur
            g.retired = True
            close = g.pins == 0
        if close:
            g.eng.close()

    # ---- the compaction ----
    def _hook(self, name: str):
        f = self.hooks.get(name)
        if f:
            f()

    def compact_online(self) -> dict:
        """Builds the next generation beside the serving one and swaps.  -> what happened: routes carried over, ops replayed (and how many of
        them under the lock), replay rounds."""
        if not self._compacting.acquire(blocking=False):
            raise RuntimeError("a compaction is already running")
        try:
            with self._lock:
                a = self._cur
                if self._closed:
                    raise RuntimeError("closed")
                # (everything that can fail comes BEFORE the pin and the log exist: ADVICE r4 -- a raising info() used to leak both)
                if a.eng.retain_find_all()[0]:
                    # (dist worker and retain store are separate coprocs with an index each: DW/DistWorkerCoProc.java, RetainStoreCoProc.java)
                    raise NotImplementedError("this handle also holds retained topics: only the route index is carried into the next generation")
                n_ids = int(a.eng.info().next_route_id)
                a.pins += 1          # A must outlive the export whatever happens
                self._log = []
                self._log_broken = False
            nxt = None
            try:
                self._hook("after_snapshot")
                # 2. A's live keys.  A key deleted meanwhile comes back empty (its delete is in the log: a no-op on B), a key added meanwhile
                #    has an id >= n_ids (its put is in the log)
                live: List[bytes] = []
                for lo in range(0, n_ids, self.EXPORT_CHUNK):
                    live += [k for k in a.eng.route_keys(np.arange(lo, min(lo + self.EXPORT_CHUNK, n_ids), dtype=np.uint32)) if k]
                self._hook("after_export")
                nxt = Engine(**self._args)
                nxt.rebuild(sorted(live))
                self._hook("after_build")
                # 3. replay: outside the lock while the log keeps filling, the last round under it
                replayed = under_lock = rounds = 0
                while True:
                    with self._lock:
                        chunk, self._log = self._log, []
                        if self._log_broken:
                            raise RuntimeError("an apply() failed while the compaction ran: the next generation is abandoned")
                        if self._closed:
                            raise RuntimeError("closed while the compaction ran: the next generation is dropped")
                        last = not chunk or rounds >= self.MAX_REPLAY_ROUNDS
                        if last:
                            if chunk:
                                nxt.apply(chunk)
                                under_lock = len(chunk)
                                replayed += len(chunk)
                            # 4. swap
                            self._log = None
                            self._cur = _Gen(nxt, a.number + 1)
                            a.retired = True
                            break
                    nxt.apply(chunk)
                    replayed += len(chunk)
                    rounds += 1
                    self._hook("after_replay_round")
                nxt = None
                return {"generation": a.number + 1, "routes_carried": len(live), "ops_replayed": replayed, "ops_replayed_under_lock": under_lock,
                        "replay_rounds": rounds}
            finally:
                with self._lock:
                    if self._cur is a:  # failed before the swap: A goes on serving, the half-built generation is dropped
                        self._log = None
                    a.pins -= 1
                    close = a.retired and a.pins == 0
                if nxt is not None:
                    nxt.close()
                if close:
                    a.eng.close()
        finally:
            self._compacting.release()
