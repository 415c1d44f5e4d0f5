"""GPU parity: the HIP hot path against the CPU oracle on the same seeded inputs.

Bit-exact bar: every integer of node state (digest over self state, queues, views, suspicion
timers), the per-tick edge list (after canonical sort), the census and the counters.
"""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

pytestmark = pytest.mark.gpu

STAT_KEYS = ["node_rounds_active", "node_rounds_quiescent", "packets_sent", "packets_dropped", "msgs_sent",
             "msgs_applied", "probes", "probe_acks", "probe_indirect_acks", "probe_failures", "nacks_missed",
             "refutes", "suspicion_timeouts", "confirmations", "edges", "msgs_filtered", "push_pulls", "queue_drops",
             "inbox_overflow", "piggybacks", "msgs_piggybacked",
             "subject_overflow", "view_drops", "folds", "fold_freed"]


def pair(hip, oracle, which=abi.PRESET_LAN, **kw):
    return Sim(hip, preset(hip, which, **kw)), Sim(oracle, preset(oracle, which, **kw))


def assert_same(a, b, subjects=(), tag=""):
    a.sync()
    assert a.digest() == b.digest(), f"state digest differs {tag}"
    sa, sb = a.stats(), b.stats()
    for k in STAT_KEYS:
        assert sa[k] == sb[k], f"stat {k}: hip {sa[k]} oracle {sb[k]} {tag}"
    for r, x in subjects:
        ca, cb = a.census(r, x), b.census(r, x)
        for f, _ in abi.Census._fields_:
            va, vb = getattr(ca, f), getattr(cb, f)
            va, vb = (list(va), list(vb)) if hasattr(va, "__len__") else (va, vb)
            assert va == vb, f"census[{r},{x}].{f}: hip {va} oracle {vb} {tag}"


def test_backend_is_hip(hip):
    assert hip.swim_backend() == b"hip-gfx950"


def test_single_failure_lockstep_small(hip, oracle):
    """config #1 shape: 128 nodes, LAN timers, kill node 17 at t=10s; compare every tick."""
    a, b = pair(hip, oracle, n_nodes=128, seed=1, trace_ticks=400)
    for s in (a, b):
        s.step_ms(10000)
        s.kill(0, [17])
    a.edges()        # (tile buckets: the first call switches the recording of what the receivers' filter drops on)
    for t in range(150):
        a.step(1); b.step(1)
        ea, eb = a.edges(), b.edges()
        assert np.array_equal(ea, eb), f"edge list differs at tick {a.now()[0]}"
        assert a.digest() == b.digest(), f"digest differs at tick {a.now()[0]}"
    assert_same(a, b, [(0, 17)])
    assert b.census(0, 17).all_dead_ms != abi.NONE
    assert np.array_equal(a.trace(0, 17, 100, 150), b.trace(0, 17, 100, 150))
    assert a.poll_events() == b.poll_events()
    ma, mb = a.members(0, 3), b.members(0, 3)
    assert np.array_equal(ma, mb)


@pytest.mark.parametrize("n,reps,seed", [(4096, 3, 11), (65536, 2, 5)])
def test_single_failure_replicas(hip, oracle, n, reps, seed):
    """config #2 shape: kill one uniformly drawn node per replica at t=5s, run past detection."""
    a, b = pair(hip, oracle, n_nodes=n, n_replicas=reps, seed=seed)
    rng = np.random.default_rng(seed)
    victims = [int(rng.integers(n)) for _ in range(reps)]
    for s in (a, b):
        s.step_ms(5000)
        for r, v in enumerate(victims):
            s.kill(r, [v])
    for chunk in range(8):
        a.step_ms(5000); b.step_ms(5000)
        assert_same(a, b, list(enumerate(victims)), tag=f"after {5 + 5 * (chunk + 1)}s")
    for r, v in enumerate(victims):
        c = a.census(r, v)
        assert c.all_dead_ms != abi.NONE and c.first_suspect_ms < c.first_dead_ms <= c.all_dead_ms


def test_update_rumour_wan(hip, oracle):
    """config #3 shape: WAN timers, one alive-update injected at node 0, infection curve."""
    for k in (2, 3, 5):
        a, b = pair(hip, oracle, abi.PRESET_WAN, n_nodes=32768, seed=3, gossip_nodes=k, trace_ticks=64)
        for s in (a, b):
            s.update(0, [0])
            s.step(60)
        assert_same(a, b, [(0, 0)], tag=f"k={k}")
        ta, tb = a.trace(0, 0, 0, 60), b.trace(0, 0, 0, 60)
        assert np.array_equal(ta, tb)
        assert ta[-1, 4] == 32767 and a.census(0, 0).all_current_ms != abi.NONE
        assert np.all(np.diff(ta[:, 4].astype(np.int64)) >= 0)      # infection is monotone


def test_loss_refute_and_partition(hip, oracle):
    """packet loss => false suspicions => refutes (incarnation bumps); then a partition."""
    a, b = pair(hip, oracle, n_nodes=2048, seed=9, subject_cap=1024, view_cap=1024, queue_cap=32, inbox_cap=256,
                loss_q32=int(0.10 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    for s in (a, b):
        s.step_ms(20000)
    assert_same(a, b, tag="lossy")
    st = b.stats()
    assert st["refutes"] > 0 and st["probe_failures"] > 0 and st["queue_drops"] > 0   # Prune() path too
    mask = np.zeros(2048, dtype=np.uint8); mask[:100] = 1
    for s in (a, b):
        s.set_loss(0.0)
        s.partition(0, mask)
        s.step_ms(15000)
    assert_same(a, b, tag="partitioned")


def test_leave_and_revive(hip, oracle):
    a, b = pair(hip, oracle, n_nodes=1024, seed=2, subject_cap=16)
    for s in (a, b):
        s.step_ms(1000)
        s.leave(0, [5, 900])
        s.step_ms(3000)
        s.kill(0, [5, 900, 33])
        s.step_ms(8000)            # 33 is suspected by now, not yet declared dead
        s.revive(0, [33])          # ...comes back, is probed with ping+suspect, refutes
        s.step_ms(40000)
    assert_same(a, b, [(0, 5), (0, 900), (0, 33)])
    assert a.view(0, 1, 5).state == abi.STATE_LEFT and a.view(0, 1, 5).status == abi.MEMBER_LEFT
    c = a.census(0, 33)
    assert c.by_state[abi.STATE_ALIVE] == c.n_observers       # 33 refuted its own death
    assert a.node_info(0, 33).incarnation > 1
    assert a.poll_events() == b.poll_events()


@pytest.mark.parametrize("n_shards", [2, 4])
def test_sharded_population_matches_unsharded(hip, oracle, n_shards):
    """the population block-partitioned over several simulators (all on this one
    device, records handed over in-process) must reproduce the unsharded oracle bit for bit."""
    from consul_amd.dist import LocalExchange, ShardedSim
    kw = dict(n_nodes=4096, n_replicas=2, seed=5, subject_cap=256, view_cap=256, queue_cap=16, inbox_cap=1024,
              loss_q32=int(0.05 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw))
                     for i in range(n_shards)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(3000)
        s.kill(0, [100, 3000]); s.kill(1, [7]); s.update(1, [2048])
        s.step_ms(30000)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in STAT_KEYS:
        if k != "subject_overflow":                 # (edges and msgs_filtered included: the receiving shard judges what crosses a boundary)
            assert a[k] == b[k], k
    assert a["edges_remote"] > 0
    sh.close()


def test_sharded_quiet_cluster_stays_off_the_wire_and_wakes_up(hip, oracle):
    """Piggy-back orders for nodes of other shards are only filed while somebody may have something queued
    (swim_peer_activity): a quiescent sharded cluster exchanges nothing, and every way of waking it up (failure,
    update, user-visible leave) still reproduces the unsharded oracle bit for bit."""
    from consul_amd.dist import LocalExchange, ShardedSim
    kw = dict(n_nodes=4096, seed=8, subject_cap=64, view_cap=64, queue_cap=16, inbox_cap=1024, push_pull_interval_ms=0)
    sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(3000)
    assert sh.stats()["edges_remote"] == 0 and sh.digest() == ref.digest()
    for s in (sh, ref):
        s.update(0, [5]); s.step(1); s.update(0, [4000]); s.step_ms(2000)
    assert sh.digest() == ref.digest()
    for s in (sh, ref):
        s.step_ms(20000)                                     # everything retires: quiet again
        s.kill(0, [2500]); s.step_ms(40000)
        s.leave(0, [77]); s.step_ms(10000)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in ("piggybacks", "msgs_piggybacked", "msgs_sent", "msgs_applied", "probe_failures", "suspicion_timeouts"):
        assert a[k] == b[k], k
    assert a["piggybacks"] > 0
    sh.close()


# ---- BASELINE's full sizes: golden fixtures + size-independent properties -----------------------------
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("k", [2, 3, 5])
def test_config3_full_size_matches_golden_curve(hip, k):
    """config #3: N = 1 048 576, DefaultWANConfig timers, fan-out sweep, one update rumour at node 0.
    The fixture was produced by the oracle in the build container (tools/make_golden.py)."""
    g = json.load(open(os.path.join(GOLDEN, "config3_infection_1m.json")))
    want = g["curves"][str(k)]
    s = Sim(hip, preset(hip, abi.PRESET_WAN, gossip_nodes=k, trace_ticks=64, **g["config"]))
    s.update(0, [0])
    s.step(45)
    s.sync()
    got = [int(x) for x in s.trace(0, 0, 0, 45)[:, 4]]
    assert got == want["infected"]
    assert got.index(1048575) + 1 == want["rounds_to_full"]
    assert s.census(0, 0).all_current_ms == want["all_current_ms"]
    assert f"{s.digest():#018x}" == want["digest"]
    assert all(b >= a for a, b in zip(got, got[1:]))               # infection is monotone
    st = s.stats()
    assert st["msgs_applied"][abi.MSG_ALIVE] == 1048575             # everybody adopted it exactly once


def test_full_size_sharded_equals_unsharded(hip):
    """1 048 576 nodes split over 2 and 4 simulators on this device: digests add up to the same value."""
    from consul_amd.dist import LocalExchange, ShardedSim
    kw = dict(n_nodes=1048576, seed=4, gossip_nodes=3, subject_cap=4)
    ref = Sim(hip, preset(hip, abi.PRESET_WAN, **kw))
    ref.update(0, [123456]); ref.kill(0, [777]); ref.step(30); ref.sync()
    want = ref.digest(); ref.close()
    for n_shards in (2, 4):
        sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_WAN, shard_rank=i, n_shards=n_shards, **kw))
                         for i in range(n_shards)], LocalExchange())
        sh.update(0, [123456]); sh.kill(0, [777]); sh.step(30); sh.sync()
        assert sh.digest() == want
        sh.close()


def test_runs_are_deterministic_and_seeds_differ(hip):
    kw = dict(n_nodes=65536, n_replicas=4, subject_cap=4)
    out = []
    for seed in (1, 1, 2):
        s = Sim(hip, preset(hip, abi.PRESET_LAN, seed=seed, **kw))
        s.step_ms(2000); s.kill(0, [4242]); s.kill(3, [99]); s.step_ms(28000); s.sync()
        out.append((s.digest(), s.census(0, 4242).all_dead_ms))
        s.close()
    assert out[0] == out[1] and out[0][0] != out[2][0]


def test_finer_quantum_parity(hip, oracle):
    """quantum_ms = 50 instead of the gcd (100): twice the ticks per round, finer stagger; still bit-exact."""
    a, b = pair(hip, oracle, n_nodes=8192, seed=6, quantum_ms=50)
    assert a.derived.gossip_period == 4 and a.derived.probe_period == 20
    for s in (a, b):
        s.step_ms(1500); s.kill(0, [5000]); s.step_ms(30000)
    assert_same(a, b, [(0, 5000)])
    assert a.census(0, 5000).all_dead_ms != abi.NONE


def test_push_pull_parity_and_heal(hip, oracle):
    """a14: push-pull on a short period so it dominates: kill, let everyone declare it dead, revive,
    and let the state exchange heal the views — bit-exact against the oracle, sharded too."""
    from consul_amd.dist import LocalExchange, ShardedSim
    kw = dict(n_nodes=4096, n_replicas=2, seed=8, subject_cap=16, push_pull_interval_ms=1000)
    a, b = pair(hip, oracle, **kw)
    sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    for s in (a, b, sh):
        s.step_ms(1000); s.kill(0, [33]); s.kill(1, [5]); s.update(1, [77]); s.step_ms(40000)
        s.revive(0, [33]); s.step_ms(30000)
    assert_same(a, b, [(0, 33), (1, 5), (1, 77)])
    sh.sync()
    assert sh.digest() == b.digest()
    c = a.census(0, 33)
    assert c.by_state[abi.STATE_ALIVE] == c.n_observers and a.node_info(0, 33).incarnation == 2
    assert a.stats()["push_pulls"] == b.stats()["push_pulls"] > 0
    sh.close()


# ---- BASELINE config #5's shape at test size: churn + Serf user-event flood + Lifeguard -------------------------
def _churn_and_flood(sims, n, seconds, rng_seed, events_per_s=3, churn=0.03):
    """Every simulated second: a uniformly drawn `churn` share of the nodes flips alive <-> dead (kill / revive) and
    `events_per_s` user events start at uniformly drawn live origins.  The same stimulus goes to every simulator."""
    rng = np.random.default_rng(rng_seed)
    dead = np.zeros(n, dtype=bool)
    eid = 1
    for sec in range(seconds):
        flip = rng.choice(n, size=int(n * churn), replace=False)
        kill, revive = [int(x) for x in flip if not dead[x]], [int(x) for x in flip if dead[x]]
        dead[flip] = ~dead[flip]
        origins = [int(x) for x in rng.choice(np.flatnonzero(~dead), size=events_per_s)]
        for s in sims:
            if kill: s.kill(0, kill)
            if revive: s.revive(0, revive)
            for j, o in enumerate(origins):
                s.user_event(0, o, eid + j)
        eid += events_per_s
        for s in sims:
            s.step_ms(1000)


@pytest.mark.parametrize("fanout", [3, 5])
def test_churn_and_event_flood(hip, oracle, fanout):
    """Config #5's shape (Lifeguard on, churn, user-event flood) at 2 048 nodes, compared every two seconds.
    fan-out 5 runs the wide-array kernel variant with the Serf queue."""
    kw = dict(n_nodes=2048, seed=21, gossip_nodes=fanout, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, subject_cap=2048, view_cap=2048,
              queue_cap=16, event_queue_cap=16, inbox_cap=512, push_pull_interval_ms=0)
    a, b = pair(hip, oracle, **kw)
    for step in range(8):
        _churn_and_flood((a, b), 2048, 2, rng_seed=100 + step)
        a.sync()
        assert a.digest() == b.digest(), f"after {2 * (step + 1)} s"
    sa, sb = a.stats(), b.stats()
    for k in STAT_KEYS + ["user_events_delivered", "user_events_deduped", "user_events_stale", "event_drops"]:
        assert sa[k] == sb[k], k
    assert sb["user_events_delivered"] > 0 and sb["refutes"] > 0 and sb["suspicion_timeouts"] > 0 and sb["piggybacks"] > 0


def test_churn_and_event_flood_sharded(hip, oracle):
    """The same on two HIP shards (sharded kernel variant with the Serf queue) against the unsharded oracle."""
    from consul_amd.dist import LocalExchange, ShardedSim
    kw = dict(n_nodes=2048, seed=22, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, subject_cap=2048, view_cap=2048, queue_cap=16,
              event_queue_cap=16, inbox_cap=1024, push_pull_interval_ms=0)
    sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    _churn_and_flood((sh, ref), 2048, 12, rng_seed=7)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in ("user_events_delivered", "user_events_deduped", "msgs_applied", "piggybacks", "probe_failures", "refutes"):
        assert a[k] == b[k], k
    sh.close()


def test_queues_that_do_not_fit_the_lds_are_refused(hip):
    """The gossip role stages 16 B x 256 lanes x (queue_cap + event_queue_cap) in LDS: 64 slots would need 256 KB."""
    cfg = preset(hip, abi.PRESET_LAN, n_nodes=1024, queue_cap=32, event_queue_cap=32, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS)
    with pytest.raises(Exception) as e:
        Sim(hip, cfg)
    assert "ERANGE" in str(e.value) or "-34" in str(e.value)


def test_tcp_fallback_under_loss_parity(hip, oracle):
    """Default flags (TCP fallback ping on) with 20 % packet loss, a real failure and a partition: the lossy probes are
    saved by TCP on both libraries alike, the dead node and the partitioned ones are still found out."""
    a, b = pair(hip, oracle, n_nodes=2048, seed=13, subject_cap=256, view_cap=256, queue_cap=16, inbox_cap=256, loss_q32=int(0.20 * 2**32))
    mask = np.zeros(2048, dtype=np.uint8); mask[1000:1040] = 1
    for s in (a, b):
        s.step_ms(8000)
        s.kill(0, [321])
        s.step_ms(12000)
        s.partition(0, mask)
        s.step_ms(20000)
    assert_same(a, b, [(0, 321)])
    st = b.stats()
    assert st["probe_tcp_acks"] > 0 and a.stats()["probe_tcp_acks"] == st["probe_tcp_acks"] and st["probe_failures"] > 0


def test_no_tcp_ping_across_datacenters_parity(hip, oracle):
    """swim_set_tcp_class (DisableTcpPingsForNode, agent/consul/server_serf.go:222-232): three datacenters under 25 % loss and
    a real failure — both libraries skip the same TCP pings, raise the same false suspicions and refute them alike."""
    a, b = pair(hip, oracle, n_nodes=1536, seed=21, subject_cap=512, view_cap=256, queue_cap=16, inbox_cap=256, loss_q32=int(0.25 * 2**32))
    for s in (a, b):
        s.set_tcp_class(0, range(512, 1024), 1); s.set_tcp_class(0, range(1024, 1536), 2)
        s.step_ms(6000)
        s.kill(0, [700])
        s.step_ms(20000)
    assert_same(a, b, [(0, 700)])
    st = b.stats()
    assert st["probe_tcp_acks"] > 0 and st["probe_failures"] > 0 and st["refutes"] > 0 and a.stats()["probe_tcp_acks"] == st["probe_tcp_acks"]


def test_revived_node_fires_overdue_suspicion_timers(hip, oracle):
    """A node that went down while it suspected somebody resumes its old views when it comes back, and the suspicion
    timer that ran out meanwhile fires in its first tick (found by tools/fuzz_parity.py: the expire role's gate only
    knew the observers that were running at the last census)."""
    a, b = pair(hip, oracle, n_nodes=64, seed=4, subject_cap=16)
    for s in (a, b):
        s.step_ms(2000); s.kill(0, [9]); s.step_ms(3000)
    holder = next(o for o in range(64) if o != 9 and b.view(0, o, 9).state == abi.STATE_SUSPECT)
    for s in (a, b):
        s.kill(0, [holder]); s.step_ms(60000)          # everybody else has long declared 9 dead; holder's timer is overdue
        s.revive(0, [holder])
    t0 = b.stats()["suspicion_timeouts"]
    for t in range(30):
        a.step(1); b.step(1); a.sync()
        assert a.digest() == b.digest(), f"tick {t} after the revive"
        assert a.stats()["suspicion_timeouts"] == b.stats()["suspicion_timeouts"]
    assert b.stats()["suspicion_timeouts"] > t0 and b.view(0, holder, 9).state == abi.STATE_DEAD


def test_randomised_parity_cases(hip, oracle):
    """A fixed slice of tools/fuzz_parity.py (random configurations, shards, stimulus and transport-bridge operations):
    none of the cases that run to the end may differ from the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    # the cases of seed 131 whose configuration is accepted and which run to the end without overflowing a bounded structure
    # (that depends on the draw alone: established with `tools/fuzz_parity.py --backend oracle --seed 131`); every one of
    # them must match the oracle
    cases = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 16, 17, 18, 20, 21, 23, 24, 25, 26, 29, 30, 32, 33, 34, 36]
    tally = {}
    for k in cases:
        res = fz.run_case(k, hip, oracle, 131, False)
        tally[res] = tally.get(res, 0) + 1
    assert tally == {"ok": len(cases)}, tally
    # a second slice (seed 977), drawn after round 3 added serf's reconnect(), narrow event-buffer slots and — on the product
    # side only — the dense pair store to the draw: same rule
    cases = [1, 3, 6, 7, 8, 10, 11, 12, 14, 15, 16, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29]
    tally = {}
    for k in cases:
        res = fz.run_case(k, hip, oracle, 977, False)
        tally[res] = tally.get(res, 0) + 1
    assert tally == {"ok": len(cases)}, tally
