"""SWIM_F_UNBOUNDED_QUEUE on the checker (CPU): memberlist's TransmitLimitedQueue never drops a broadcast before its retransmit limit (queue.go,
pin /root/reference go.mod:80; Consul sizes only serf's event queue: internal/gossip/libserf/serf.go:24-27).  The checker's queue is a sorted
array either way (one walk per GetBroadcasts); with the flag it grows on demand.  Pinned here: a queue that grows == a queue of 4 096 slots that
never fills, bit for bit; what a bound costs config #4's answer (336 s at 32 entries, 31 s with none, at 8 192 nodes) in small;
the fold rule the flag adds (no fold while a rumour about the subject is queued).  The device side: tests/test_unbounded_queue_gpu.py and
tests/test_emulated_kernels.py."""
import numpy as np

from consul_amd import abi
from consul_amd.sim import Sim, preset

UQ = abi.F_DEFAULT | abi.F_UNBOUNDED_QUEUE


def mass(oracle, n, nv, seed, **kw):
    victims = np.random.default_rng(seed).choice(n, size=nv, replace=False).tolist()
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=n, seed=seed, inbox_cap=2 * nv + 256, view_cap=nv + 64, subject_cap=4, **kw))
    s.step_ms(1000); s.kill(0, victims)
    return s


def test_a_queue_that_grows_is_a_queue_that_never_fills(oracle):
    n, nv = 1024, 51
    a, b = mass(oracle, n, nv, 5, queue_cap=4096), mass(oracle, n, nv, 5, queue_cap=4, flags=UQ)
    for sec in range(30):
        a.step_ms(1000); b.step_ms(1000)
        assert a.digest() == b.digest(), sec
    assert a.stats() == b.stats() and a.stats()["queue_drops"] == 0
    pairs, by = a.detection(0)
    assert by[2] + by[3] == pairs == (n - nv) * nv
    qa, qb = a.node_info(0, 3), b.node_info(0, 3)
    assert qa.queue_len == qb.queue_len and bytes(qa) == bytes(qb)


def test_what_a_bound_costs(oracle):
    """The same failure with 8 slots per node: rumours are pruned (counted) and full detection takes several times as long."""
    n, nv = 1024, 51
    def seconds_to_full_detection(s, limit=400):
        for sec in range(1, limit):
            s.step_ms(1000)
            pairs, by = s.detection(0)
            if by[2] + by[3] == pairs:
                return sec
        return limit
    free, tight = mass(oracle, n, nv, 5, queue_cap=4, flags=UQ), mass(oracle, n, nv, 5, queue_cap=8)
    t_free, t_tight = seconds_to_full_detection(free), seconds_to_full_detection(tight)
    assert free.stats()["queue_drops"] == 0 and tight.stats()["queue_drops"] > 0
    assert t_free <= 25 and t_tight >= t_free + 10, (t_free, t_tight)      # (20 s against 38 s: the suspicion timeout + a dissemination; with 8 slots the rumours take turns)


def test_no_fold_while_a_rumour_is_queued(oracle):
    """A killed node that nobody gossips to any more is folded into the base row once everybody holds it dead for longer than GossipToTheDeadTime —
    with the flag only after the last queued rumour about it has retired, too (the device's rows hold the queue: DESIGN 4b)."""
    kw = dict(n_nodes=256, seed=3, view_cap=16, subject_cap=4, fold_interval_ms=1000, gossip_to_dead_ms=2000, inbox_cap=64)
    a = Sim(oracle, preset(oracle, abi.PRESET_LAN, queue_cap=64, **kw)); b = Sim(oracle, preset(oracle, abi.PRESET_LAN, queue_cap=4, flags=UQ, **kw))
    for s in (a, b):
        s.step_ms(1000); s.kill(0, [77]); s.revive(0, [77]); s.kill(0, [77])
    fa = fb = None
    for sec in range(1, 60):
        a.step_ms(1000); b.step_ms(1000)
        fa = fa or (sec if a.stats()["folds"] else None); fb = fb or (sec if b.stats()["folds"] else None)
    assert fa and fb and fb >= fa, (fa, fb)
    assert a.view(0, 5, 77).state == b.view(0, 5, 77).state == abi.STATE_DEAD


def test_sharded_checker_with_unbounded_queues_and_folds(oracle):
    """The fold rule across shards: a shard that still has a rumour about the subject queued poisons its census record, so all shards decide
    alike — 2 shards of the checker with unbounded queues against the unsharded checker, folds every second, to the end of a mass event."""
    from consul_amd.dist import LocalExchange, ShardedSim
    n, nv = 1024, 60
    victims = np.random.default_rng(2).choice(n, size=nv, replace=False).tolist()
    kw = dict(n_nodes=n, seed=2, queue_cap=4, inbox_cap=2 * nv + 256, view_cap=nv + 64, subject_cap=4, fold_interval_ms=1000, gossip_to_dead_ms=3000, flags=UQ)
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(1000); s.kill(0, victims)
    for sec in range(0, 60, 5):
        sh.step_ms(5000); ref.step_ms(5000)
        assert sh.digest() == ref.digest(), sec
    a, b = sh.stats(), ref.stats()
    for k in ("msgs_applied", "folds", "fold_freed", "queue_drops", "edges", "msgs_filtered"):
        assert a[k] == b[k], k
    assert b["folds"] == nv and b["queue_drops"] == 0
