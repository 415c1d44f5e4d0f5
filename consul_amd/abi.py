"""ctypes mirror of include/swimsim.h (the C-ABI of the SWIM hot-path simulator).

Pure declarations: struct layouts, constants and function prototypes.  Nothing here decides
WHICH shared library is loaded — `consul_amd.lib` loads the HIP product library, and the test
suite binds the same prototypes onto the CPU oracle to check it.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 7
NONE = 0xFFFFFFFF
FRAME_MAGIC = 0x4D415246          # SWIM_FRAME_MAGIC: the header record of a frame of the framed exchange

OK, EINVAL, ENOMEM, ENODEV, ERANGE, EOVERFLOW, ESTATE, EIO = 0, -22, -12, -19, -34, -75, -71, -5
ERRNAMES = {EINVAL: "SWIM_EINVAL", ENOMEM: "SWIM_ENOMEM", ENODEV: "SWIM_ENODEV",
            ERANGE: "SWIM_ERANGE", EOVERFLOW: "SWIM_EOVERFLOW", ESTATE: "SWIM_ESTATE", EIO: "SWIM_EIO"}

STATE_ALIVE, STATE_SUSPECT, STATE_DEAD, STATE_LEFT = 0, 1, 2, 3
MSG_ALIVE, MSG_SUSPECT, MSG_DEAD, MSG_USER = 0, 1, 2, 3
MEMBER_NONE, MEMBER_ALIVE, MEMBER_LEAVING, MEMBER_LEFT, MEMBER_FAILED = 0, 1, 2, 3, 4
(EVENT_MEMBER_JOIN, EVENT_MEMBER_LEAVE, EVENT_MEMBER_FAILED, EVENT_MEMBER_UPDATE,
 EVENT_MEMBER_REAP, EVENT_USER, EVENT_QUERY) = range(7)
PRESET_LAN, PRESET_WAN, PRESET_LOCAL = 0, 1, 2
INFO_TILE_BUCKETS, INFO_MAILBOX_KIND, INFO_DEVICE_BYTES = 0, 1, 2      # swim_info keys
F_BUDDY_SUSPECT, F_NACK, F_SERF_EVENTS, F_FILTER_NOOP, F_PIGGYBACK, F_TCP_FALLBACK, F_COORDINATES, F_UNBOUNDED_QUEUE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20, 0x40, 0x80
F_DEFAULT = F_BUDDY_SUSPECT | F_NACK | F_FILTER_NOOP | F_PIGGYBACK | F_TCP_FALLBACK
SUBJECT_PULL, SUBJECT_PIGGY = 0xFFFFFFFE, 0xFFFFFFFD
INTENT_LEAVE, INTENT_PRUNE = 0x80000000, 0x40000000

u8, u32, u64, i32 = C.c_uint8, C.c_uint32, C.c_uint64, C.c_int32


class Config(C.Structure):
    _fields_ = [(n, u32) for n in (
        "abi_version", "n_nodes", "n_replicas", "n_initial",
        "gossip_nodes", "gossip_interval_ms", "probe_interval_ms", "probe_timeout_ms",
        "suspicion_mult", "retransmit_mult", "indirect_checks", "suspicion_max_timeout_mult",
        "awareness_max_mult", "gossip_to_dead_ms", "udp_buffer_size", "push_pull_interval_ms")] + [
        ("msg_len", u32 * 4), ("ctl_len", u32 * 4)] + [(n, u32) for n in (
        "quantum_ms", "phase_chunk", "queue_cap", "inbox_cap", "subject_cap", "view_cap", "mass_rows", "reap_interval_ms", "reconnect_timeout_ms", "tombstone_timeout_ms", "reconnect_interval_ms", "fold_interval_ms",
        "event_queue_cap", "event_buffer", "event_ids_per_ltime", "loss_q32", "flags", "watch_node", "trace_ticks",
        "shard_rank", "n_shards", "device", "rtt_scale_us", "rtt_height_us", "rtt_jitter_us")] + [("seed", u64)]


COORD_DIMS = 8


class Coordinate(C.Structure):
    """coordinate.Coordinate (serf/coordinate), seconds"""
    _fields_ = [("vec", C.c_double * COORD_DIMS), ("error", C.c_double), ("adjustment", C.c_double), ("height", C.c_double)]


class Derived(C.Structure):
    _fields_ = [(n, u32) for n in (
        "quantum_ms", "gossip_period", "probe_period", "probe_timeout_ticks", "phase_chunk",
        "retransmit_limit", "suspicion_k", "suspicion_min_ms", "suspicion_max_ms")] + [
        ("suspicion_timeout_ms", u32 * 8), ("node_scale_milli", u32),
        ("push_pull_scale", u32), ("push_pull_period_ticks", u32), ("packet_budget", u32),
        ("view_cap", u32), ("fold_period_ticks", u32), ("reap_period_ticks", u32), ("reconnect_period_ticks", u32)]


class Member(C.Structure):
    _fields_ = [("id", u32), ("incarnation", u32), ("state_change_ms", u32),
                ("state", u8), ("status", u8), ("n_confirm", u8), ("_pad", u8)]


class Event(C.Structure):
    _fields_ = [(n, u32) for n in ("time_ms", "replica", "type", "node", "incarnation", "observer")] + [("ltime", C.c_uint64)]


class Rumour(C.Structure):
    _fields_ = [("subject", u32), ("incarnation", u32), ("from_", u32),
                ("type", u8), ("transmits", u8), ("_pad", u8 * 2), ("seq", u32)]


class NodeInfo(C.Structure):
    _fields_ = [(n, u32) for n in (
        "incarnation", "probe_target", "probe_deadline_tick", "probe_cursor", "probe_epoch",
        "queue_len", "event_queue_len")] + [
        ("alive", u8), ("leaving", u8), ("awareness", u8), ("partition", u8),
        ("event_clock", C.c_uint64),
        ("queue", Rumour * 32)]


class Census(C.Structure):
    _fields_ = [("n_observers", u32), ("by_state", u32 * 4), ("n_current", u32),
                ("first_suspect_ms", u32), ("first_dead_ms", u32), ("all_dead_ms", u32),
                ("all_current_ms", u32)]


class Detection(C.Structure):
    _fields_ = [("pairs", u64), ("by_state", u64 * 4)]


class Edge(C.Structure):
    _fields_ = [("dst", u32), ("subject", u32), ("incarnation", u32), ("meta", u32)]


class Stats(C.Structure):
    _fields_ = [("ticks", u64), ("gossip_rounds", u64), ("node_rounds_active", u64),
                ("node_rounds_quiescent", u64), ("packets_sent", u64), ("packets_dropped", u64),
                ("msgs_sent", u64 * 4), ("msgs_applied", u64 * 4), ("probes", u64),
                ("probe_acks", u64), ("probe_indirect_acks", u64), ("probe_failures", u64),
                ("nacks_missed", u64), ("refutes", u64), ("suspicion_timeouts", u64),
                ("confirmations", u64), ("edges", u64), ("edges_remote", u64),
                ("queue_drops", u64), ("inbox_overflow", u64), ("subject_overflow", u64),
                ("event_drops", u64), ("user_events_delivered", u64),
                ("user_events_deduped", u64), ("user_events_stale", u64), ("msgs_filtered", u64), ("push_pulls", u64),
                ("piggybacks", u64), ("msgs_piggybacked", u64), ("probe_tcp_acks", u64),
                ("view_drops", u64), ("view_evictions", u64), ("intents_applied", u64), ("reaped", u64), ("joins", u64), ("join_failures", u64), ("folds", u64), ("fold_freed", u64),
                ("coord_updates", u64), ("coord_resets", u64), ("reconnects", u64), ("reconnects_reached", u64), ("inbox_peak", u64)]


class XchgHandle(C.Structure):
    _fields_ = [("bytes", u8 * 96)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("launches", u64), ("total_ms", C.c_double)]


P = C.POINTER
SimP = C.c_void_p

# name -> (restype, argtypes); every symbol include/swimsim.h declares
PROTOTYPES = {
    "swim_config_preset": (C.c_int, [P(Config), C.c_int]),
    "swim_config_derive": (C.c_int, [P(Config), P(Derived)]),
    "swim_create": (C.c_int, [P(Config), P(SimP)]),
    "swim_destroy": (C.c_int, [SimP]),
    "swim_backend": (C.c_char_p, []),
    "swim_last_error": (C.c_char_p, [SimP]),
    "swim_step": (C.c_int, [SimP, u32]),
    "swim_sync": (C.c_int, [SimP]),
    "swim_now": (C.c_int, [SimP, P(u32), P(u32)]),
    "swim_tick_begin": (C.c_int, [SimP]),
    "swim_outbound": (C.c_int, [SimP, u32, P(C.c_void_p), P(u32)]),
    "swim_outbound_capacity": (u32, [SimP, u32]),
    "swim_peer_activity": (C.c_int, [SimP, C.c_int]),
    "swim_activity": (C.c_int, [SimP, C.POINTER(C.c_int)]),
    "swim_stream": (C.c_int, [SimP, P(C.c_void_p)]),
    "swim_outbound_raw": (C.c_int, [SimP, u32, P(C.c_void_p), P(C.c_void_p)]),
    "swim_inbound": (C.c_int, [SimP, C.c_void_p, u32]),
    "swim_frame_records": (u32, [SimP]),
    "swim_frame_pack": (C.c_int, [SimP, C.c_void_p, u32]),
    "swim_frame_pack_fill": (C.c_int, [SimP, C.c_void_p, u32]),
    "swim_frame_deliver": (C.c_int, [SimP, C.c_void_p, u32]),
    "swim_tick_end": (C.c_int, [SimP]),
    "swim_tick_end_begin": (C.c_int, [SimP]),
    "swim_xchg_export": (C.c_int, [SimP, P(XchgHandle)]),
    "swim_xchg_connect": (C.c_int, [SimP, P(XchgHandle)]),
    "swim_xchg_step": (C.c_int, [SimP, u32]),
    "swim_inject_kill": (C.c_int, [SimP, u32, P(u32), C.c_size_t]),
    "swim_inject_revive": (C.c_int, [SimP, u32, P(u32), C.c_size_t]),
    "swim_inject_leave": (C.c_int, [SimP, u32, P(u32), C.c_size_t]),
    "swim_inject_update": (C.c_int, [SimP, u32, P(u32), C.c_size_t]),
    "swim_force_leave": (C.c_int, [SimP, u32, u32, u32, C.c_int, P(C.c_uint64)]),
    "swim_inject_join": (C.c_int, [SimP, u32, P(u32), C.c_size_t, u32]),
    "swim_inject_partition": (C.c_int, [SimP, u32, P(u8)]),
    "swim_set_loss": (C.c_int, [SimP, u32]),
    "swim_set_tcp_class": (C.c_int, [SimP, u32, P(u32), C.c_size_t, u8]),
    "swim_user_event": (C.c_int, [SimP, u32, u32, u32, P(C.c_uint64)]),
    "swim_watch": (C.c_int, [SimP, u32, u32]),
    "swim_members": (C.c_int, [SimP, u32, u32, P(Member), C.c_size_t, P(C.c_size_t)]),
    "swim_view": (C.c_int, [SimP, u32, u32, u32, P(Member)]),
    "swim_watch_events": (C.c_int, [SimP, u32, u32]),
    "swim_poll_events": (C.c_int, [SimP, P(Event), C.c_size_t, P(C.c_size_t)]),
    "swim_node_info_get": (C.c_int, [SimP, u32, u32, P(NodeInfo)]),
    "swim_event_queued": (C.c_int, [SimP, u32, u32, u32, u64, C.POINTER(C.c_int)]),
    "swim_census_get": (C.c_int, [SimP, u32, u32, P(Census)]),
    "swim_detection_get": (C.c_int, [SimP, u32, P(Detection)]),
    "swim_trace_read": (C.c_int, [SimP, u32, u32, u32, u32, P(u32)]),
    "swim_stats": (C.c_int, [SimP, P(Stats)]),
    "swim_coordinate_get": (C.c_int, [SimP, u32, u32, P(Coordinate)]),
    "swim_coordinate_distance": (C.c_double, [P(Coordinate), P(Coordinate)]),
    "swim_rtt_truth": (C.c_int, [SimP, u32, u32, u32, P(u32)]),
    "swim_debug_edges": (C.c_int, [SimP, P(Edge), C.c_size_t, P(C.c_size_t)]),
    "swim_info": (C.c_int, [SimP, u32, P(C.c_uint64)]),
    "swim_state_digest": (C.c_int, [SimP, P(u64)]),
    "swim_checkpoint_save": (C.c_int, [SimP, C.c_char_p]),
    "swim_checkpoint_load": (C.c_int, [SimP, C.c_char_p]),
    "swim_profile": (C.c_int, [SimP, C.c_int]),
    "swim_profile_read": (C.c_int, [SimP, P(KernelTime), C.c_size_t, P(C.c_size_t)]),
    "swim_transport_write_to": (C.c_int, [SimP, u32, u32, u32, P(Edge), C.c_size_t]),
    "swim_transport_poll": (C.c_int, [SimP, u32, u32, P(Edge), C.c_size_t, P(C.c_size_t)]),
    "swim_kat_philox4x32": (None, [P(u32), P(u32), P(u32)]),
    "swim_kat_probe_perm": (u32, [u64, u32, u32, u32, u32]),
    "swim_kat_remaining_suspicion_ms": (i32, [u32, u32, u32, u32, u32]),
    "swim_kat_phase_of": (None, [P(Config), u32, P(u32), P(u32)]),
    "swim_kat_awareness_apply": (u32, [u32, u32, C.c_int32]),
    "swim_kat_awareness_scale_ms": (u32, [u32, u32]),
}


def bind(cdll: C.CDLL) -> C.CDLL:
    """Attach restype/argtypes for every ABI symbol; raises AttributeError if one is missing."""
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(cdll, name)
        fn.restype = res
        fn.argtypes = args
    return cdll
