"""Population sharding: every replica's node ids are block-partitioned over `n_shards` simulators
and the cross-shard gossip records of each tick are exchanged between swim_tick_begin and
swim_tick_end.

Two exchanges implement the same contract — "after run(), every shard has been handed the records
every other shard addressed to it":

  TorchExchange   one process per GPU; ONE equal-split all_to_all_single over torch.distributed per
                  tick, of frames whose headers carry the device-side counts; the frames follow the load (one 64-byte-per-peer
                  host look per tick; a tick that does not fit is exchanged again before it is delivered).
                  With the "nccl" backend that is RCCL over xGMI on device buffers; with "gloo" the
                  same code moves host buffers (how the CPU tests cover the N>1 path).
  LocalExchange   several shards inside one process (e.g. all on one GPU): pointer hand-over.
                  Used to test the sharded kernels where only one device is available.

The records are swim_edge (16 bytes); uniform random peer choice spreads remote records evenly over
the destination shards, which suits xGMI's point-to-point mesh.
"""
from __future__ import annotations

from typing import List, Sequence

from .sim import Sim


class TorchExchange:
    """The per-tick exchange as ONE equal-split all_to_all_single over torch.distributed, with frames sized from the load.

    An all-to-all wants its sizes on the host; the record counts of a tick live on the device.  Every rank sends every other
    rank a FRAME of F 16-byte records whose first record is a header the device writes (swimsim.h: swim_frame_pack_fill —
    {records the segment has, activity | need << 1, tick + 1, magic}, `need` = the sender's largest segment of the tick).
    Per tick: pack (one kernel) -> all_to_all_single(recv, send) -> read the W headers back (64 bytes per peer: the one host
    look of the tick) -> deliver (one kernel).  F follows the load: twice the largest `need` any rank reported over the last
    WINDOW ticks, a power of two, at least MIN_FRAME records — a quiet tick moves 1 KB per peer.  When a tick's largest segment
    does not fit (every rank sees every sender's `need`, so all decide alike) the tick is packed and exchanged AGAIN with frames
    that hold it, before anything is delivered: nothing is lost, nothing overflows.  (Round 4 shipped the library's bound —
    1 + the largest outbound capacity, tens of millions of records at bench scale — whatever the fill: ADVICE r4.)
    On the GPU (backend nccl = RCCL over xGMI) everything is issued on the simulator's own HIP stream
    (torch.cuda.ExternalStream made torch's current stream); with gloo the same code moves host buffers of the checker,
    which is how the CPU tests cover the N > 1 path.

    frame_records: None = adaptive (above).  A number = fixed frames of that size through swim_frame_pack: no host look at
    all, and a segment that does not fit raises the sticky edge-list overflow (SWIM_EOVERFLOW at the next sync).
    """

    MAX_SHARDS = 16
    MIN_FRAME = 64
    WINDOW = 8

    def __init__(self, group, device_index: int | None, frame_records: int | None = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.on_gpu = device_index is not None
        self.device = torch.device("cuda", device_index) if self.on_gpu else torch.device("cpu")
        self.frame_records = frame_records
        self._bound = None          # simulator the frames were sized for
        self._send = self._recv = None
        self._F = 0
        self._cap = 0               # records per frame the buffers can hold
        self._recent = []           # the population's largest segment in each of the last WINDOW ticks
        self.ticks = self.retries = 0
        self.records_on_wire = 0    # frame records this rank put on the wire (x 16 bytes)

    def _alloc(self, F: int):
        torch = self.torch
        # (zero-filled once: a frame's tail beyond its count is never read, but it does cross the wire)
        self._send = torch.zeros((self.world * F, 4), dtype=torch.int32, device=self.device)
        self._recv = torch.zeros((self.world * F, 4), dtype=torch.int32, device=self.device)
        self._cap = F

    def _bind(self, sim: Sim):
        torch = self.torch
        F = self.frame_records or self.MIN_FRAME
        if F < 2:
            raise ValueError("a frame holds a header and at least one record")
        self._F = F
        if self.on_gpu:
            # everything this exchange issues goes to the simulator's stream: make it torch's current stream once
            # (entering a stream context per tick costs ~9 us of host time that sits on the tick's critical path)
            self._stream = torch.cuda.ExternalStream(sim.stream_ptr(), device=self.device)
            torch.cuda.set_stream(self._stream)
        self._alloc(max(F, 1024))
        if self.on_gpu:
            self._stream.synchronize()
        self._bound = sim

    @property
    def frame_bytes_per_tick(self) -> int:
        """What one rank put on the wire per tick on average so far (its own frame stays home)."""
        return 16 * self.records_on_wire // max(self.ticks, 1) if self.ticks else (self.world - 1) * self._F * 16

    def _exchange(self, sim: Sim, F: int, fill: bool):
        W = self.world
        if F > self._cap:                                        # (what is in the buffers is dead: every tick packs afresh)
            if self.on_gpu:
                self._stream.synchronize()
            self._alloc(1 << (F - 1).bit_length())
        send, recv = self._send[:W * F], self._recv[:W * F]
        (sim.frame_pack_fill if fill else sim.frame_pack)(send.data_ptr(), F)
        self.dist.all_to_all_single(recv, send, group=self.group)
        self.records_on_wire += (W - 1) * F
        return send, recv

    def run(self, sims: Sequence[Sim]):
        (sim,) = sims
        if self._bound is not sim:
            self._bind(sim)
        W, F = self.world, self._F
        self.ticks += 1
        if self.frame_records:                                   # fixed frames: no host look, overflow is loud
            _, recv = self._exchange(sim, F, fill=False)
            sim.frame_deliver(recv.data_ptr(), F)
            return
        send, recv = self._exchange(sim, F, fill=True)
        # the one host look of the tick: `need` of every sender (the own frame's header carries this rank's)
        # (ONE read-back — the maximum is taken on the device; rounds 4-5 read the two words back separately: two device-to-host syncs on the tick's critical path)
        own = send[self.rank * F, 1]
        need = int((self.torch.maximum(recv.view(W, F, 4)[:, 0, 1].max(), own) if W > 1 else own).item()) >> 1
        if need > F - 1:                                         # (every rank computes the same maximum: all repeat together)
            self.retries += 1
            F = 1 << need.bit_length()                           # > need
            send, recv = self._exchange(sim, F, fill=True)
        sim.frame_deliver(recv.data_ptr(), F)
        self._recent = (self._recent + [need])[-self.WINDOW:]
        self._F = max(self.MIN_FRAME, 1 << (2 * max(self._recent)).bit_length())

    def close(self):
        """Give torch its default stream back (the simulator's stream is about to be destroyed)."""
        if self.on_gpu and self._bound is not None:
            self.torch.cuda.current_stream(self.device).synchronize()
            self.torch.cuda.set_stream(self.torch.cuda.default_stream(self.device))
        self._bound = None


class LocalFramedExchange:
    """The framed exchange between shards that live in ONE process (tests; one device): every shard packs its frames, the
    frames are transposed with plain copies (what the collective does between processes), every shard delivers.  `alloc(n)`
    returns a buffer of n records and `ptr(buf)` its address — numpy on the checker, device memory on the product
    library; `copy(dst, d0, src, s0, n)` moves n records; `sync()` orders the phases (the shards run on different streams).
    `read(buf, i)` (optional) returns record i of a buffer as four words: with it and no fixed frame_records the frames are
    sized from the load exactly like TorchExchange's — swim_frame_pack_fill, the headers read back, a tick whose largest
    segment does not fit packed and moved again before anything is delivered."""

    ORACLE_FRAME = 1 << 16          # (the checker's lists are unbounded: swim_frame_records answers 0 there)

    def __init__(self, alloc, ptr, copy, sync=lambda: None, frame_records: int | None = None, read=None):
        self.alloc, self.ptr, self.copy, self.sync, self.read = alloc, ptr, copy, sync, read
        self.frame_records = frame_records
        self.adaptive = read is not None and not frame_records
        self._F = self._cap = 0
        self._recent = []
        self.retries = 0

    def _move(self, sims, F, fill):
        W = len(sims)
        if F > self._cap:
            self._cap = F
            self._send = [self.alloc(W * F) for _ in sims]
            self._recv = [self.alloc(W * F) for _ in sims]
        for s, b in zip(sims, self._send):
            (s.frame_pack_fill if fill else s.frame_pack)(self.ptr(b), F)
        for s in sims:
            s.sync()
        for i in range(W):
            for j in range(W):
                if i != j:
                    self.copy(self._recv[j], i * F, self._send[i], j * F, F)
        self.sync()

    def run(self, sims: Sequence[Sim]):
        W = len(sims)
        if not self._F:
            self._F = TorchExchange.MIN_FRAME if self.adaptive else self.frame_records or sims[0].frame_records() or self.ORACLE_FRAME
        F = self._F
        self._move(sims, F, self.adaptive)
        if self.adaptive:
            need = max(self.read(b, i * F)[1] >> 1 for i, b in enumerate(self._send))     # every shard's own header carries its need
            if need > F - 1:
                self.retries += 1
                F = 1 << need.bit_length()
                self._move(sims, F, True)
            self._recent = (self._recent + [need])[-TorchExchange.WINDOW:]
            self._F = max(TorchExchange.MIN_FRAME, 1 << (2 * max(self._recent)).bit_length())
        for s, b in zip(sims, self._recv):
            s.frame_deliver(self.ptr(b), F)
        for s in sims:
            s.sync()


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
