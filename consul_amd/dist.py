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
    def __init__(self, group, device_index: int | None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.on_gpu = device_index is not None
        self.device = torch.device("cuda", device_index) if self.on_gpu else torch.device("cpu")
        self._keep = None

    def _segment(self, ptr: int, count: int):
        torch = self.torch
        if count == 0:
            return torch.empty((0, 4), dtype=torch.int32, device=self.device)
        if self.on_gpu:
            return torch.as_tensor(_DevMem(ptr, count * 4), device=self.device).view(count, 4)
        buf = (C.c_int32 * (count * 4)).from_address(ptr)
        return torch.from_numpy(np.frombuffer(buf, dtype=np.int32).reshape(count, 4))

    def run(self, sims: Sequence[Sim]):
        (sim,) = sims
        torch, dist = self.torch, self.dist
        segs, counts = [], []
        for sh in range(self.world):
            ptr, n = sim.outbound(sh)            # syncs the simulator's stream: the records are complete
            if sh == self.rank:
                n = 0                            # the local segment never crosses the wire
            segs.append(self._segment(ptr, n)); counts.append(n)
        send_counts = torch.tensor(counts, dtype=torch.int64, device=self.device)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        rc = [int(x) for x in recv_counts.tolist()]
        send = torch.cat(segs, dim=0) if sum(counts) else torch.empty((0, 4), dtype=torch.int32, device=self.device)
        recv = torch.empty((sum(rc), 4), dtype=torch.int32, device=self.device)
        dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=counts, group=self.group)
        if self.on_gpu:
            torch.cuda.current_stream(self.device).synchronize()   # hand-over to the simulator's stream
        if recv.shape[0]:
            sim.inbound(recv.data_ptr(), recv.shape[0])
            if self.on_gpu:
                sim.sync()                                         # recv may be recycled after this
        self._keep = recv


class LocalExchange:
    """All shards live in this process; outbound segment j of shard i is handed to shard j."""

    def run(self, sims: Sequence[Sim]):
        n = len(sims)
        segs = [[s.outbound(j) for j in range(n)] for s in sims]
        for i in range(n):
            for j in range(n):
                if i != j and segs[i][j][1]:
                    sims[j].inbound(*segs[i][j])


class ShardedSim:
    """Drive one or more shards of the same population in lock step."""

    def __init__(self, sims, exchange):
        self.sims: List[Sim] = list(sims) if isinstance(sims, (list, tuple)) else [sims]
        self.sim = self.sims[0]
        self.exchange = exchange

    def step(self, n_ticks: int = 1):
        for _ in range(n_ticks):
            for s in self.sims:
                s.tick_begin()
            self.exchange.run(self.sims)
            for s in self.sims:
                s.tick_end()

    def step_ms(self, ms: int):
        self.step(ms // self.sim.derived.quantum_ms)

    def __getattr__(self, name):
        # stimulus is replicated on every shard (each applies what it owns)
        if name in ("kill", "revive", "leave", "update", "partition", "set_loss", "sync"):
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
                tot[k] = [a + b for a, b in zip(tot[k], v)] if isinstance(v, list) else tot[k] + v
        return tot

    def close(self):
        for s in self.sims:
            s.close()
