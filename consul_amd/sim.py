"""Thin object wrapper over the swimsim C-ABI (include/swimsim.h).

`Sim` is library-agnostic: it is handed a ctypes CDLL already bound with `abi.bind`.  The
product entry point is `consul_amd.open_sim()` / `consul_amd.lib.load()`, which binds the HIP
library and nothing else; tests construct `Sim(oracle_cdll, cfg)` to drive the checker.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, Sequence

import numpy as np

from . import abi

EDGE_DTYPE = np.dtype([("dst", "<u4"), ("subject", "<u4"), ("incarnation", "<u4"), ("meta", "<u4")])


class SwimError(RuntimeError):
    def __init__(self, fn: str, rc: int, detail: str = ""):
        self.rc = rc
        super().__init__(f"{fn} -> {abi.ERRNAMES.get(rc, rc)}{': ' + detail if detail else ''}")


def preset(cdll, which: int = abi.PRESET_LAN, **overrides) -> abi.Config:
    """memberlist.DefaultLANConfig()/DefaultWANConfig()/DefaultLocalConfig() + field overrides."""
    cfg = abi.Config()
    rc = cdll.swim_config_preset(C.byref(cfg), which)
    if rc:
        raise SwimError("swim_config_preset", rc)
    for k, v in overrides.items():
        if k in ("msg_len", "ctl_len"):
            for i, x in enumerate(v):
                getattr(cfg, k)[i] = x
        else:
            if not hasattr(cfg, k):
                raise AttributeError(f"swim_config has no field {k!r}")
            setattr(cfg, k, v)
    return cfg


def derive(cdll, cfg: abi.Config) -> abi.Derived:
    d = abi.Derived()
    rc = cdll.swim_config_derive(C.byref(cfg), C.byref(d))
    if rc:
        raise SwimError("swim_config_derive", rc)
    return d


def _ids(ids: Iterable[int]):
    arr = np.ascontiguousarray(np.fromiter(ids, dtype=np.uint32))
    return arr, arr.ctypes.data_as(C.POINTER(abi.u32)), arr.size


class Sim:
    def __init__(self, cdll, cfg: abi.Config):
        self._l = cdll
        self.cfg = cfg
        self.derived = derive(cdll, cfg)
        h = abi.SimP()
        rc = cdll.swim_create(C.byref(cfg), C.byref(h))
        if rc:
            raise SwimError("swim_create", rc)
        self._h = h

    # -- plumbing -----------------------------------------------------------------------------
    def _ck(self, fn: str, rc: int):
        if rc:
            err = self._l.swim_last_error(self._h)
            raise SwimError(fn, rc, err.decode() if err else "")

    def close(self):
        if getattr(self, "_h", None):
            self._l.swim_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def backend(self) -> str:
        return self._l.swim_backend().decode()

    # -- time ---------------------------------------------------------------------------------
    def step(self, n_ticks: int = 1):
        self._ck("swim_step", self._l.swim_step(self._h, n_ticks))

    def sync(self):
        self._ck("swim_sync", self._l.swim_sync(self._h))

    def now(self):
        t, ms = abi.u32(), abi.u32()
        self._ck("swim_now", self._l.swim_now(self._h, C.byref(t), C.byref(ms)))
        return t.value, ms.value

    def step_ms(self, ms: int):
        q = self.derived.quantum_ms
        if ms % q:
            raise ValueError(f"{ms} ms is not a multiple of the {q} ms quantum")
        self.step(ms // q)

    # -- split tick (sharded population) --------------------------------------------------------
    def tick_begin(self):
        self._ck("swim_tick_begin", self._l.swim_tick_begin(self._h))

    def outbound(self, shard: int):
        """(raw pointer, count) of the segment addressed to `shard` (device ptr on the HIP lib)."""
        p, n = C.c_void_p(), abi.u32()
        self._ck("swim_outbound", self._l.swim_outbound(self._h, shard, C.byref(p), C.byref(n)))
        return p.value or 0, n.value

    def outbound_capacity(self, shard: int) -> int:
        """Upper bound on records ever returned for `shard` (size of the buffer behind outbound())."""
        return int(self._l.swim_outbound_capacity(self._h, shard))

    def stream_ptr(self) -> int:
        p = C.c_void_p()
        self._ck("swim_stream", self._l.swim_stream(self._h, C.byref(p)))
        return p.value or 0

    def outbound_raw(self, shard: int):
        """(segment pointer, counters pointer) on the device, no synchronisation."""
        seg, cnt = C.c_void_p(), C.c_void_p()
        self._ck("swim_outbound_raw", self._l.swim_outbound_raw(self._h, shard, C.byref(seg), C.byref(cnt)))
        return seg.value or 0, cnt.value or 0

    def inbound(self, ptr: int, count: int):
        self._ck("swim_inbound", self._l.swim_inbound(self._h, C.c_void_p(ptr), count))

    def peer_activity(self, active: bool):
        self._ck("swim_peer_activity", self._l.swim_peer_activity(self._h, 1 if active else 0))

    def activity(self) -> bool:
        a = C.c_int(1)
        self._ck("swim_activity", self._l.swim_activity(self._h, C.byref(a)))
        return bool(a.value)

    # -- framed exchange (one equal-split collective per tick, the counts stay on the device) --------
    def frame_records(self) -> int:
        """The smallest frame that can never overflow (0 on the checker: its lists are unbounded)."""
        return int(self._l.swim_frame_records(self._h))

    def frame_pack(self, send_ptr: int, frame_records: int):
        self._ck("swim_frame_pack", self._l.swim_frame_pack(self._h, C.c_void_p(send_ptr), frame_records))

    def frame_pack_fill(self, send_ptr: int, frame_records: int):
        """Frames sized from the load: what does not fit is not an error — the headers say what there is (swimsim.h)."""
        self._ck("swim_frame_pack_fill", self._l.swim_frame_pack_fill(self._h, C.c_void_p(send_ptr), frame_records))

    def frame_deliver(self, recv_ptr: int, frame_records: int):
        self._ck("swim_frame_deliver", self._l.swim_frame_deliver(self._h, C.c_void_p(recv_ptr), frame_records))

    def tick_end(self):
        self._ck("swim_tick_end", self._l.swim_tick_end(self._h))

    def tick_end_begin(self):
        self._ck("swim_tick_end_begin", self._l.swim_tick_end_begin(self._h))

    # -- device-driven exchange (swim_xchg_*) ----------------------------------------------------
    def xchg_export(self) -> bytes:
        h = abi.XchgHandle()
        self._ck("swim_xchg_export", self._l.swim_xchg_export(self._h, C.byref(h)))
        return bytes(h.bytes)

    def xchg_connect(self, handles: Sequence[bytes]):
        """`handles[rank]` = xchg_export() of every shard of the population (the own entry is ignored)."""
        arr = (abi.XchgHandle * len(handles))()
        for i, b in enumerate(handles):
            C.memmove(arr[i].bytes, b, len(b))
        self._ck("swim_xchg_connect", self._l.swim_xchg_connect(self._h, arr))

    def xchg_step(self, n_ticks: int = 1):
        self._ck("swim_xchg_step", self._l.swim_xchg_step(self._h, n_ticks))

    # -- stimulus -----------------------------------------------------------------------------
    def kill(self, replica: int, ids: Iterable[int]):
        a, p, n = _ids(ids)
        self._ck("swim_inject_kill", self._l.swim_inject_kill(self._h, replica, p, n))

    def revive(self, replica: int, ids: Iterable[int]):
        a, p, n = _ids(ids)
        self._ck("swim_inject_revive", self._l.swim_inject_revive(self._h, replica, p, n))

    def leave(self, replica: int, ids: Iterable[int]):
        a, p, n = _ids(ids)
        self._ck("swim_inject_leave", self._l.swim_inject_leave(self._h, replica, p, n))

    def update(self, replica: int, ids: Iterable[int]):
        a, p, n = _ids(ids)
        self._ck("swim_inject_update", self._l.swim_inject_update(self._h, replica, p, n))

    def join(self, replica: int, ids: Iterable[int], via: int):
        """serf.Create + serf.Join([via]) for nodes that are not running."""
        a, p, n = _ids(ids)
        self._ck("swim_inject_join", self._l.swim_inject_join(self._h, replica, p, n, via))

    def force_leave(self, replica: int, origin: int, node: int, prune: bool = False) -> int:
        """serf.RemoveFailedNode[Prune] called on `origin`; returns the intent's Lamport time."""
        lt = C.c_uint64()
        self._ck("swim_force_leave", self._l.swim_force_leave(self._h, replica, origin, node, int(prune), C.byref(lt)))
        return lt.value

    def event_queued(self, replica: int, node: int, event_id: int, ltime: int) -> bool:
        """serf's notifyCh of a broadcast, as a question: is {event_id, ltime} still in `node`'s serf queue?"""
        q = C.c_int()
        self._ck("swim_event_queued", self._l.swim_event_queued(self._h, replica, node, event_id, ltime, C.byref(q)))
        return bool(q.value)

    def partition(self, replica: int, group_of_node: Sequence[int]):
        g = np.ascontiguousarray(group_of_node, dtype=np.uint8)
        if g.size != self.cfg.n_nodes:
            raise ValueError("partition mask must have n_nodes entries")
        self._ck("swim_inject_partition",
                 self._l.swim_inject_partition(self._h, replica, g.ctypes.data_as(C.POINTER(abi.u8))))

    def watch(self, replica: int, subject: int):
        """Keep census / first-* stamps / trace of `subject` from now on (swim_watch)."""
        self._ck("swim_watch", self._l.swim_watch(self._h, replica, subject))

    def set_loss(self, prob: float):
        self._ck("swim_set_loss", self._l.swim_set_loss(self._h, min(int(prob * 2**32), 2**32 - 1)))

    def set_tcp_class(self, replica: int, ids: Iterable[int], tcp_class: int):
        """memberlist's DisableTcpPingsForNode as Consul's WAN pool sets it: no TCP fallback ping between nodes of different classes."""
        a, p, n = _ids(ids)
        self._ck("swim_set_tcp_class", self._l.swim_set_tcp_class(self._h, replica, p, n, tcp_class))

    def user_event(self, replica: int, origin: int, event_id: int) -> int:
        lt = C.c_uint64()
        self._ck("swim_user_event",
                 self._l.swim_user_event(self._h, replica, origin, event_id, C.byref(lt)))
        return lt.value

    # -- observation --------------------------------------------------------------------------
    def view(self, replica: int, observer: int, subject: int) -> abi.Member:
        m = abi.Member()
        self._ck("swim_view", self._l.swim_view(self._h, replica, observer, subject, C.byref(m)))
        return m

    def members(self, replica: int, observer: int):
        n = self.cfg.n_nodes
        buf = (abi.Member * n)()
        cnt = C.c_size_t()
        self._ck("swim_members", self._l.swim_members(self._h, replica, observer, buf, n, C.byref(cnt)))
        return np.frombuffer(buf, dtype=np.dtype([("id", "<u4"), ("incarnation", "<u4"),
                                                  ("state_change_ms", "<u4"), ("state", "u1"),
                                                  ("status", "u1"), ("n_confirm", "u1"),
                                                  ("_pad", "u1")]))[: cnt.value].copy()

    def poll_events(self, cap: int = 4096):
        buf = (abi.Event * cap)()
        cnt = C.c_size_t()
        self._ck("swim_poll_events", self._l.swim_poll_events(self._h, buf, cap, C.byref(cnt)))
        return [(e.time_ms, e.replica, e.type, e.node, e.ltime, e.incarnation, e.observer) for e in buf[: cnt.value]]

    def watch_events(self, replica: int, node: int):
        """swim_watch_events: record this node's serf events too (an EventCh per agent)."""
        self._ck("swim_watch_events", self._l.swim_watch_events(self._h, replica, node))

    def node_info(self, replica: int, node: int) -> abi.NodeInfo:
        o = abi.NodeInfo()
        self._ck("swim_node_info_get", self._l.swim_node_info_get(self._h, replica, node, C.byref(o)))
        return o

    def census(self, replica: int, subject: int) -> abi.Census:
        o = abi.Census()
        self._ck("swim_census_get", self._l.swim_census_get(self._h, replica, subject, C.byref(o)))
        return o

    def detection(self, replica: int = 0):
        """(pairs, [alive, suspect, dead, left]) over (acting observer, unreachable subject) pairs: swim_detection_get."""
        o = abi.Detection()
        self._ck("swim_detection_get", self._l.swim_detection_get(self._h, replica, C.byref(o)))
        return int(o.pairs), [int(x) for x in o.by_state]

    def trace(self, replica: int, subject: int, first_tick: int, n: int) -> np.ndarray:
        out = np.zeros((n, 5), dtype=np.uint32)
        self._ck("swim_trace_read", self._l.swim_trace_read(
            self._h, replica, subject, first_tick, n, out.ctypes.data_as(C.POINTER(abi.u32))))
        return out

    def coordinate(self, replica: int, node: int) -> abi.Coordinate:
        """serf.GetCoordinate() of a virtual node (SWIM_F_COORDINATES)"""
        o = abi.Coordinate()
        self._ck("swim_coordinate_get", self._l.swim_coordinate_get(self._h, replica, node, C.byref(o)))
        return o

    def distance(self, a, b) -> float:
        """librtt.ComputeDistance: seconds; +inf when either coordinate is None"""
        return float(self._l.swim_coordinate_distance(C.byref(a) if a is not None else None, C.byref(b) if b is not None else None))

    def rtt_truth(self, replica: int, a: int, b: int) -> int:
        """the latency model's round-trip time between two nodes, microseconds, without jitter"""
        o = abi.u32()
        self._ck("swim_rtt_truth", self._l.swim_rtt_truth(self._h, replica, a, b, C.byref(o)))
        return int(o.value)

    def info(self, what: int) -> int:
        """swim_info: how the handle is laid out (abi.INFO_*)."""
        o = C.c_uint64()
        self._ck("swim_info", self._l.swim_info(self._h, what, C.byref(o)))
        return int(o.value)

    def stats(self) -> dict:
        o = abi.Stats()
        self._ck("swim_stats", self._l.swim_stats(self._h, C.byref(o)))
        d = {}
        for name, typ in abi.Stats._fields_:
            v = getattr(o, name)
            d[name] = list(v) if hasattr(v, "__len__") else int(v)
        return d

    def edges(self, cap: int = 1 << 22) -> np.ndarray:
        """Last tick's rumour deliveries, canonically sorted (order of emission is not defined)."""
        cnt = C.c_size_t()
        self._ck("swim_debug_edges", self._l.swim_debug_edges(self._h, None, 0, C.byref(cnt)))
        n = cnt.value
        if n > cap:
            raise SwimError("swim_debug_edges", abi.ERANGE, f"{n} edges > cap {cap}")
        arr = np.zeros(n, dtype=EDGE_DTYPE)
        if n:
            self._ck("swim_debug_edges", self._l.swim_debug_edges(
                self._h, arr.ctypes.data_as(C.POINTER(abi.Edge)), n, C.byref(cnt)))
        return np.sort(arr, order=["dst", "subject", "meta", "incarnation"])

    # -- memberlist.Transport bridge (rumour granularity) ------------------------------------------
    def transport_write_to(self, replica: int, attached: int, dst: int, msgs):
        """Transport.WriteToAddress: `msgs` = iterable of (subject, incarnation, type, from)."""
        msgs = list(msgs)
        buf = (abi.Edge * max(len(msgs), 1))()
        for i, (subject, inc, typ, frm) in enumerate(msgs):
            buf[i] = abi.Edge(0, subject, inc, (typ << 30) | (frm & 0x3FFFFFFF))
        self._ck("swim_transport_write_to",
                 self._l.swim_transport_write_to(self._h, replica, attached, dst, buf, len(msgs)))

    def transport_poll(self, replica: int, attached: int, cap: int = 4096):
        """Transport.PacketCh: sorted list of (sender, subject, incarnation, type, from)."""
        buf = (abi.Edge * cap)()
        n = C.c_size_t()
        self._ck("swim_transport_poll", self._l.swim_transport_poll(self._h, replica, attached, buf, cap, C.byref(n)))
        return sorted((e.dst, e.subject, e.incarnation, e.meta >> 30, e.meta & 0x3FFFFFFF) for e in buf[: n.value])

    def profile(self, enable: bool = True):
        self._ck("swim_profile", self._l.swim_profile(self._h, int(enable)))

    def profile_read(self) -> dict:
        """{kernel name: (launches, total_ms)} since the last read (HIP events on the sim's stream)."""
        buf = (abi.KernelTime * 32)()
        cnt = C.c_size_t()
        self._ck("swim_profile_read", self._l.swim_profile_read(self._h, buf, 32, C.byref(cnt)))
        return {k.name.decode(): (int(k.launches), float(k.total_ms)) for k in buf[: cnt.value]}

    def save(self, path: str) -> None:
        """swim_checkpoint_save: the whole population, between two ticks, into a file"""
        self._ck("swim_checkpoint_save", self._l.swim_checkpoint_save(self._h, os.fsencode(path)))

    def load(self, path: str) -> None:
        """swim_checkpoint_load: back into a handle created from the same configuration"""
        self._ck("swim_checkpoint_load", self._l.swim_checkpoint_load(self._h, os.fsencode(path)))

    def digest(self) -> int:
        o = abi.u64()
        self._ck("swim_state_digest", self._l.swim_state_digest(self._h, C.byref(o)))
        return o.value
