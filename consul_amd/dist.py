"""Population sharding: every replica's node ids are block-partitioned over `n_shards` simulators
and the cross-shard gossip records of each tick are exchanged between swim_tick_begin and
swim_tick_end.

Two exchanges implement the same contract — "after run(), every shard has been handed the records
every other shard addressed to it":

  TorchExchange   one process per GPU; an all-to-all over torch.distributed.  With the "nccl"
                  backend that is RCCL over xGMI on the device buffers; with "gloo" the same code
                  moves host buffers (how the CPU tests cover the N>1 path).
  LocalExchange   several shards inside one process (e.g. all on one GPU): pointer hand-over.
                  Used to test the sharded kernels where only one device is available.

The records are swim_edge (16 bytes).  Counts are exchanged first (one small all-to-all), then the
payload with an uneven all_to_all_single; uniform random peer choice spreads remote records evenly
over the destination shards, which suits xGMI's point-to-point mesh.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from .sim import Sim


class _DevMem:
    """Expose a raw device pointer through __cuda_array_interface__ so torch can alias it."""

    def __init__(self, ptr: int, n_words: int):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i4", "data": (ptr, False),
                                         "version": 2, "strides": None}


class TorchExchange:
    """All-to-all of the per-destination record segments over torch.distributed.

    GPU (backend nccl = RCCL over xGMI), device driven: the collectives are issued on the simulator's
    own HIP stream (torch.cuda.ExternalStream), so kernels and exchange are ordered without host
    synchronisation.  Per tick: all-gather the device-resident record counters (no D2H/H2D bounce),
    read the gathered matrix once (the only host sync), and — only if some rank has remote records —
    a list-form all_to_all (grouped ncclSend/ncclRecv) from slices of the aliased outbound buffers
    straight into a persistent receive buffer.
    CPU (backend gloo, used by the tests): host counts + uneven all_to_all_single of the concatenated
    segments (gloo has no list form).
    """

    MAX_SHARDS = 16

    def __init__(self, group, device_index: int | None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.on_gpu = device_index is not None
        self.device = torch.device("cuda", device_index) if self.on_gpu else torch.device("cpu")
        self._bound = None          # simulator the GPU aliases belong to
        self._recv = None
        self._empty = torch.empty((0, 4), dtype=torch.int32, device=self.device)
        self.skipped = 0            # ticks in which no rank had anything to send

    # ---- GPU path -------------------------------------------------------------------------------
    def _bind(self, sim: Sim):
        torch = self.torch
        self._stream = torch.cuda.ExternalStream(sim.stream_ptr(), device=self.device)
        self._segs = []
        cnt_ptr = 0
        for sh in range(self.world):
            seg, cnt_ptr = sim.outbound_raw(sh)
            cap = sim.outbound_capacity(sh)
            self._segs.append(torch.as_tensor(_DevMem(seg, cap * 4), device=self.device).view(cap, 4))
        # the n_shards counters, then the shard's activity word (swim_peer_activity)
        self._counts = torch.as_tensor(_DevMem(cnt_ptr, self.world + 1), device=self.device)
        self._gathered = torch.empty((self.world, self.world + 1), dtype=torch.int32, device=self.device)
        self._caps = [sim.outbound_capacity(sh) for sh in range(self.world)]
        self._bound = sim
        # everything this exchange issues goes to the simulator's stream: make it torch's current stream once
        # (entering a stream context per tick costs ~9 us of host time that sits on the tick's critical path)
        torch.cuda.set_stream(self._stream)

    def _run_gpu(self, sim: Sim):
        torch, dist = self.torch, self.dist
        if self._bound is not sim:
            self._bind(sim)
        if True:
            dist.all_gather_into_tensor(self._gathered, self._counts, group=self.group)
            m = self._gathered.cpu().tolist()              # the tick's only host synchronisation; plain lists from here
            # nobody emitted anything and nobody can hold a queued broadcast: next tick's probes need not file
            # piggy-back orders for nodes of other shards, and the quiescent tick stays off the wire
            W = self.world
            sim.peer_activity(any(any(row) for row in m))
            remote_total = sum(m[i][j] for i in range(W) for j in range(W) if i != j)
            if remote_total == 0:
                self.skipped += 1
                return
            # every rank derives the same (clamped) sizes from the same matrix; a clamped segment has
            # already raised the sender's sticky overflow flag
            rcap = min(c for sh, c in enumerate(self._caps) if sh != self.rank)     # (outbound_capacity is the same number on every rank: one configuration)
            send = [0 if sh == self.rank else min(m[self.rank][sh], rcap) for sh in range(W)]
            recv_n = [0 if src == self.rank else min(m[src][self.rank], rcap) for src in range(W)]
            total = sum(recv_n)
            if self._recv is None or self._recv.shape[0] < total:
                self._recv = torch.empty((max(total, 1) * 2, 4), dtype=torch.int32, device=self.device)
            outs, off = [], 0
            for n in recv_n:
                outs.append(self._recv[off:off + n]); off += n
            ins = [self._segs[sh][: send[sh]] for sh in range(self.world)]
            dist.all_to_all(outs, ins, group=self.group)
            if total:
                sim.inbound(self._recv.data_ptr(), total)  # asynchronous copy on the same stream

    # ---- CPU path -------------------------------------------------------------------------------
    def _run_cpu(self, sim: Sim):
        torch, dist = self.torch, self.dist
        segs, counts = [], []
        for sh in range(self.world):
            ptr, n = sim.outbound(sh)
            if sh == self.rank:
                n = 0                            # the local segment never crosses the wire
            if n:
                buf = (C.c_int32 * (n * 4)).from_address(ptr)
                segs.append(torch.from_numpy(np.frombuffer(buf, dtype=np.int32).reshape(n, 4)))
            else:
                segs.append(self._empty)
            counts.append(n)
        send_counts = torch.tensor(counts, dtype=torch.int64)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        rc = [int(x) for x in recv_counts.tolist()]
        total = sum(rc)
        recv = torch.empty((total, 4), dtype=torch.int32)
        send = torch.cat(segs, dim=0) if sum(counts) else self._empty
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=counts, group=self.group)
        if total:
            sim.inbound(recv.data_ptr(), total)  # the oracle copies synchronously

    def run(self, sims: Sequence[Sim]):
        (sim,) = sims
        if self.on_gpu:
            self._run_gpu(sim)
        else:
            self._run_cpu(sim)

    def close(self):
        """Give torch its default stream back (the simulator's stream is about to be destroyed)."""
        if self.on_gpu and self._bound is not None:
            self.torch.cuda.current_stream(self.device).synchronize()
            self.torch.cuda.set_stream(self.torch.cuda.default_stream(self.device))
        self._bound = None


class LocalExchange:
    """All shards live in this process; outbound segment j of shard i is handed to shard j."""

    def run(self, sims: Sequence[Sim]):
        n = len(sims)
        active = any([s.activity() for s in sims])
        for s in sims:
            s.peer_activity(active)
        segs = [[s.outbound(j) for j in range(n)] for s in sims]
        for i in range(n):
            for j in range(n):
                if i != j and segs[i][j][1]:
                    sims[j].inbound(*segs[i][j])
        for s in sims:                       # the copies read other shards' buffers: finish them before anyone moves on
            s.sync()


class LibraryExchange:
    """The library's own device-driven exchange (swim_xchg_*: peer-mapped mailboxes, no host round trip, no collective).
    `gather` turns this process's handle(s) into the list of all shards' handles, indexed by rank — identity when every
    shard lives here, an all-gather (torch.distributed.all_gather_object, a pipe, files) when there is one per process."""

    def __init__(self, gather=None):
        self.gather = gather
        self.connected = False

    def connect(self, sims: Sequence[Sim]):
        mine = {s.cfg.shard_rank: s.xchg_export() for s in sims}
        handles = self.gather(mine) if self.gather else [mine[r] for r in range(len(sims))]
        for s in sims:
            s.xchg_connect(handles)
        self.connected = True


class ShardedSim:
    """Drive one or more shards of the same population in lock step."""

    def __init__(self, sims, exchange):
        self.sims: List[Sim] = list(sims) if isinstance(sims, (list, tuple)) else [sims]
        self.sim = self.sims[0]
        self.exchange = exchange

    def step(self, n_ticks: int = 1):
        if n_ticks <= 0:
            return
        if isinstance(self.exchange, LibraryExchange):
            if not self.exchange.connected:
                self.exchange.connect(self.sims)
            if len(self.sims) == 1:            # one shard per process: the whole run goes down in one call
                self.sims[0].xchg_step(n_ticks)
            else:                              # several shards in this process (tests): keep their streams within a tick of
                for _ in range(n_ticks):       # each other — a shard's wait kernel spins until its sources have signalled,
                    for s in self.sims:        # and the runtime multiplexes streams onto a handful of hardware queues
                        s.xchg_step(1)
            return
        for s in self.sims:
            s.tick_begin()
        for k in range(n_ticks):
            self.exchange.run(self.sims)
            for s in self.sims:            # the end of tick t and the begin of t+1 go down as one call
                s.tick_end() if k == n_ticks - 1 else s.tick_end_begin()

    def step_ms(self, ms: int):
        self.step(ms // self.sim.derived.quantum_ms)

    def __getattr__(self, name):
        # stimulus is replicated on every shard (each applies what it owns)
        if name in ("kill", "revive", "leave", "update", "partition", "set_loss", "set_tcp_class", "sync", "watch", "join"):
            def fan(*a, **k):
                out = None
                for s in self.sims:
                    out = getattr(s, name)(*a, **k)
                return out
            return fan
        raise AttributeError(name)

    def user_event(self, replica: int, origin: int, event_id: int) -> int:
        lt = [s.user_event(replica, origin, event_id) for s in self.sims]
        return min(lt)                      # non-owners answer NONE (0xFFFFFFFF)

    def force_leave(self, replica: int, origin: int, node: int, prune: bool = False) -> int:
        return min(s.force_leave(replica, origin, node, prune) for s in self.sims)

    def digest(self) -> int:
        return sum(s.digest() for s in self.sims) & 0xFFFFFFFFFFFFFFFF

    def stats(self) -> dict:
        tot = None
        for s in self.sims:
            st = s.stats()
            if tot is None:
                tot = st
                continue
            for k, v in st.items():
                if k in ("ticks", "gossip_rounds"):
                    continue
                if k == "inbox_peak":
                    tot[k] = max(tot[k], v)
                    continue
                tot[k] = [a + b for a, b in zip(tot[k], v)] if isinstance(v, list) else tot[k] + v
        return tot

    def detection(self, replica: int = 0):
        """swim_detection_get summed over the shards (each counts its own observers)."""
        pairs, by = 0, [0, 0, 0, 0]
        for s in self.sims:
            p, b = s.detection(replica)
            pairs += p
            by = [x + y for x, y in zip(by, b)]
        return pairs, by

    # ---- queries: what one shard holds is routed to its owner, what every shard holds a part of is merged -------------------
    def _owner(self, node: int) -> Sim:
        per = self.sim.cfg.n_nodes // self.sim.cfg.n_shards
        for s in self.sims:
            if s.cfg.shard_rank == node // per:
                return s
        raise LookupError(f"node {node} lives on shard {node // per}, which is not in this process")

    def view(self, replica: int, observer: int, subject: int):
        return self._owner(observer).view(replica, observer, subject)

    def members(self, replica: int, observer: int):
        return self._owner(observer).members(replica, observer)

    def node_info(self, replica: int, node: int):
        return self._owner(node).node_info(replica, node)

    def census(self, replica: int, subject: int):
        """swim_census_get over the shards of this process: every shard counts its own observers (and stamps the first / all
        times for them), so counts add up, a FIRST time is the earliest shard's, an ALL time the latest one's — and unset while any
        shard that has observers is not there yet.  (n_current counts the observers that hold the highest incarnation THEIR SHARD
        has seen of the subject: while a new incarnation has not reached every shard the sum runs ahead of an unsharded run's.)"""
        from . import abi
        parts = [s.census(replica, subject) for s in self.sims]
        out = abi.Census()
        out.n_observers = sum(p.n_observers for p in parts)
        for j in range(4):
            out.by_state[j] = sum(p.by_state[j] for p in parts)
        out.n_current = sum(p.n_current for p in parts)
        out.first_suspect_ms = min(p.first_suspect_ms for p in parts)
        out.first_dead_ms = min(p.first_dead_ms for p in parts)
        seeing = [p for p in parts if p.n_observers]
        out.all_dead_ms = max((p.all_dead_ms for p in seeing), default=abi.NONE)
        out.all_current_ms = max((p.all_current_ms for p in seeing), default=abi.NONE)
        return out

    def close(self):
        if hasattr(self.exchange, "close"):
            self.exchange.close()
        for s in self.sims:
            s.close()
