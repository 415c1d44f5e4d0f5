"""serf/coordinate on the device (k_coord_update / k_coord_commit) against the checker, bit for bit: f64 with one rounding per
operation on both sides, so the coordinates are compared with ==, not with a tolerance.  Then the property that matters, on the
HIP library alone, at BASELINE config #2's cluster size."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

pytestmark = pytest.mark.gpu
FLAGS = abi.F_DEFAULT | abi.F_COORDINATES
FIELDS = [f"v{i}" for i in range(abi.COORD_DIMS)] + ["error", "adjustment", "height"]


def bits(c):
    return [float(x).hex() for x in c.vec] + [float(c.error).hex(), float(c.adjustment).hex(), float(c.height).hex()]


def same_coordinates(a, b, replicas, nodes):
    for r in range(replicas):
        for i in nodes:
            ca, cb = bits(a.coordinate(r, i)), bits(b.coordinate(r, i))
            assert ca == cb, (r, i, [f for f, x, y in zip(FIELDS, ca, cb) if x != y])


def test_coordinates_match_the_checker_bit_for_bit(hip, oracle):
    """512-node clusters, jitter on, a failure and a restart on the way: digests (which include the raw bits of every
    coordinate), counters and sampled coordinates after every phase"""
    kw = dict(n_nodes=512, n_replicas=2, seed=17, flags=FLAGS, rtt_jitter_us=300, subject_cap=4, view_cap=16)
    a, b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    sample = list(range(0, 512, 37)) + [511]
    for phase, ticks in enumerate((1, 30, 400, 800)):
        for s in (a, b):
            if phase == 2:
                s.kill(0, [100]); s.kill(1, [7])
            if phase == 3:
                s.revive(0, [100])
            s.step(ticks); s.sync()
        assert a.digest() == b.digest(), f"phase {phase}"
        same_coordinates(a, b, 2, sample)
        sa, sb = a.stats(), b.stats()
        assert (sa["coord_updates"], sa["coord_resets"], sa["probe_acks"]) == (sb["coord_updates"], sb["coord_resets"], sb["probe_acks"])
    assert a.stats()["coord_updates"] > 100000
    a.close(); b.close()


def test_latency_filter_and_restarts_in_a_tiny_cluster(hip, oracle):
    """8 nodes: every peer comes round every 7 probes, so the per-peer latency filter (median of three) is exercised; nodes that
    start later (serf.Create + Join) begin at NewCoordinate"""
    kw = dict(n_nodes=8, n_initial=6, seed=4, flags=FLAGS, rtt_jitter_us=5000, rtt_scale_us=20000, view_cap=8, subject_cap=8)
    a, b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for step in (50, 500, 1500):
        for s in (a, b):
            if step == 1500:
                s.join(0, [6, 7], via=2)
            s.step(step); s.sync()
        assert a.digest() == b.digest()
        same_coordinates(a, b, 1, range(8))
    a.close(); b.close()


def test_predicted_round_trip_times_at_65536_nodes(hip):
    """HIP alone: 4 clusters of 65 536 nodes, 400 s of probing (400 updates per node).  Median relative error of the predicted
    round-trip time over random pairs falls from ~1 to under 15 %, the error estimates fall with it, nothing resets."""
    n = 65536
    s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, n_replicas=4, seed=2, flags=FLAGS, rtt_jitter_us=500,
                        subject_cap=2, view_cap=4, queue_cap=4, inbox_cap=24, push_pull_interval_ms=0))
    rng = np.random.default_rng(0)
    pairs = [(int(x), int(y)) for x, y in rng.integers(n, size=(150, 2)) if x != y]

    def err(r):
        e = []
        for x, y in pairs:
            t = s.rtt_truth(r, x, y) * 1e-6
            e.append(abs(s.distance(s.coordinate(r, x), s.coordinate(r, y)) - t) / t)
        return float(np.median(e))
    s.step(10); s.sync()
    e0 = err(0)
    s.step(4000); s.sync()
    e1 = [err(r) for r in range(4)]
    assert e0 > 0.6 and max(e1) < 0.15, (e0, e1)
    st = s.stats()
    assert st["coord_updates"] == st["probe_acks"] and st["coord_resets"] == 0
    assert np.mean([s.coordinate(0, x).error for x, _ in pairs[:50]]) < 0.5
    s.close()
