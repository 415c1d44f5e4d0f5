"""serf/coordinate (Vivaldi network coordinates, SWIM_F_COORDINATES) — the checker's restatement against
the reference's own table for librtt.ComputeDistance, closed forms of the first update, convergence on the latency model.
The HIP twin of the update is checked bit for bit against this in tests/test_coordinates_gpu.py."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from consul_amd import abi, lib
from consul_amd.sim import Sim, SwimError, preset

FLAGS = abi.F_DEFAULT | abi.F_COORDINATES


def generate_coordinate(rtt_seconds: float) -> abi.Coordinate:
    """librtt.GenerateCoordinate (internal/gossip/librtt/rtt.go:59-64): NewCoordinate(DefaultConfig()), Vec[0] = rtt, Height = 0"""
    c = abi.Coordinate()
    c.error, c.adjustment, c.height = 1.5, 0.0, 0.0
    c.vec[0] = rtt_seconds
    return c


def product_host_lib():
    """swim_coordinate_distance of the PRODUCT library is plain host arithmetic: callable without a GPU"""
    return abi.bind(C.CDLL(os.environ.get("SWIMSIM_LIB", lib.LIB_PATH)))


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_compute_distance_table_of_the_reference(oracle, which):
    """internal/gossip/librtt/rtt_test.go:16-75 (TestRTT_ComputeDistance), compared with == like require.Equal does"""
    l = oracle if which == "oracle" else product_host_lib()
    dist = lambda a, b: float(l.swim_coordinate_distance(C.byref(a) if a is not None else None, C.byref(b) if b is not None else None))
    ms = 1e-3
    assert dist(generate_coordinate(0), generate_coordinate(10 * ms)) == 0.010
    assert dist(generate_coordinate(10 * ms), generate_coordinate(10 * ms)) == 0.0
    assert dist(generate_coordinate(8 * ms), generate_coordinate(10 * ms)) == 0.002
    assert dist(generate_coordinate(10 * ms), generate_coordinate(8 * ms)) == 0.002
    assert dist(None, generate_coordinate(8 * ms)) == math.inf
    assert dist(generate_coordinate(8 * ms), None) == math.inf
    assert dist(None, None) == math.inf
    # DistanceTo adds both adjustments unless the sum is not positive (coordinate.go), and goes through whole nanoseconds
    a, b = generate_coordinate(0), generate_coordinate(10 * ms)
    a.adjustment, b.adjustment = 0.001, 0.0005
    assert dist(a, b) == 0.0115
    a.adjustment = -0.02
    assert dist(a, b) == 0.010
    a.adjustment, b.adjustment, a.height, b.height = 0.0, 0.0, 1e-10, 0.0          # below a nanosecond: truncated away
    assert dist(a, b) == 0.010


def test_first_update_follows_the_closed_form(oracle):
    """Two nodes, both at NewCoordinate: the first acked probe moves the prober by VivaldiCC * weight * (rtt - dist) along a
    random direction (coincident points), sets the error estimate and the first adjustment sample (client.go updateVivaldi,
    updateAdjustment; DefaultConfig: CE = CC = 0.25, ErrorMax 1.5, HeightMin 10 us, window 20)."""
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=2, seed=5, flags=FLAGS, rtt_scale_us=40000, rtt_height_us=0))
    rtt = s.rtt_truth(0, 0, 1) * 1e-6
    assert rtt > 0
    for _ in range(40):                                  # until node 0 has probed once (its probe tick depends on the stagger)
        s.step(1)
        if s.stats()["coord_updates"]:
            break
    st = s.stats()
    assert 1 <= st["coord_updates"] <= 2 and st["coord_resets"] == 0
    moved = [i for i in (0, 1) if any(v != 0.0 for v in s.coordinate(0, i).vec)]
    assert moved
    for i in moved:
        c = s.coordinate(0, i)
        dist0 = 2e-5                                     # two HeightMin, through whole nanoseconds
        weight = 0.5
        assert c.error == pytest.approx(0.25 * weight * abs(dist0 - rtt) / rtt + 1.5 * (1 - 0.25 * weight), rel=1e-12)
        force = 0.25 * weight * (rtt - dist0)
        mag = math.sqrt(sum(v * v for v in c.vec))
        grav = ((mag + 2e-5) / 150.0) ** 2               # the pull back towards the origin, second order
        assert mag == pytest.approx(force, abs=2 * grav + 1e-15)
        assert c.height == pytest.approx(10e-6, rel=1e-6)     # coincident points: the height term of ApplyForce is skipped
        assert c.adjustment == pytest.approx((rtt - (force + 2e-5)) / 40.0, rel=1e-9)
    s.close()


def median_relative_error(s, n, pairs=400, seed=0):
    rng = np.random.default_rng(seed)
    errs = []
    for _ in range(pairs):
        a, b = (int(x) for x in rng.integers(n, size=2))
        if a == b:
            continue
        truth = s.rtt_truth(0, a, b) * 1e-6
        errs.append(abs(s.distance(s.coordinate(0, a), s.coordinate(0, b)) - truth) / truth)
    return float(np.median(errs))


def test_coordinates_converge_on_the_latency_model(oracle):
    """Vivaldi's whole point: after a few hundred probes per node the predicted round-trip times are within a few percent"""
    n = 64
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=n, seed=3, flags=FLAGS, rtt_jitter_us=200))
    s.step(10)
    assert median_relative_error(s, n) > 0.6            # everybody still near the origin
    s.step(3000)                                         # 300 s: 300 probes per node
    e1 = median_relative_error(s, n)
    s.step(12000)
    e2 = median_relative_error(s, n)
    assert e1 < 0.2 and e2 < 0.05 and e2 < e1
    st = s.stats()
    assert st["coord_updates"] == st["probe_acks"] and st["coord_resets"] == 0
    s.close()


def test_a_dead_node_stops_updating_and_a_fresh_process_starts_over(oracle):
    n = 16
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=n, seed=9, flags=FLAGS, n_initial=12, view_cap=16, subject_cap=8))
    s.step(600)
    before = s.coordinate(0, 3)
    assert any(v != 0.0 for v in before.vec)
    fresh = s.coordinate(0, 13)                          # never started: NewCoordinate
    assert all(v == 0.0 for v in fresh.vec) and fresh.error == 1.5 and fresh.height == 10e-6
    s.kill(0, [3]); s.step(300)
    after = s.coordinate(0, 3)
    assert list(after.vec) == list(before.vec) and after.adjustment == before.adjustment
    s.join(0, [3, 13], via=0); s.step(5)
    again = s.coordinate(0, 3)                           # restarted: serf.Create makes a new coordinate client
    assert again.error > 1.0 and math.sqrt(sum(v * v for v in again.vec)) < math.sqrt(sum(v * v for v in before.vec))
    s.step(600)
    assert any(v != 0.0 for v in s.coordinate(0, 13).vec)
    s.close()


def test_coordinates_are_refused_on_sharded_handles(oracle):
    with pytest.raises(SwimError):
        Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=64, flags=FLAGS, n_shards=2, shard_rank=0))
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=64))
    with pytest.raises(SwimError):
        s.coordinate(0, 1)                               # the flag is off: no coordinates to report
    s.close()
