"""serf's reconnect() (swim_config.reconnect_interval_ms): what heals a partition that outlasted
GossipToTheDeadTime.  By then both sides hold each other Dead for so long that nobody gossips to, probes or push-pulls with the
other side any more: without reconnect the split is permanent (checked), with it every node keeps trying one of its Failed
members per interval — memberlist.Join([addr]), a state exchange — and the first that gets through makes the member refute."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

KW = dict(n_nodes=512, seed=3, view_cap=512, queue_cap=16, inbox_cap=4096, gossip_to_dead_ms=10000, push_pull_interval_ms=5000)


def split_and_heal(sim, n, on_second=None):
    half = np.zeros(n, dtype=np.uint8); half[n // 2:] = 1
    sim.step_ms(1000); sim.partition(0, half)
    for sec in range(140):                      # both halves declare each other dead (~110 s), then 20 s more: past GossipToTheDeadTime
        sim.step_ms(1000)
        if on_second:
            on_second(sec)
    pairs, by = sim.detection(0)
    assert by[2] + by[3] == pairs == 2 * (n // 2) ** 2
    sim.partition(0, np.zeros(n, dtype=np.uint8))
    for sec in range(140, 200):
        sim.step_ms(1000)
        if on_second:
            on_second(sec)


def held_dead(sim, n):
    return sum(1 for x in range(n // 2, n) if sim.view(0, 0, x).state >= abi.STATE_DEAD)


def test_without_reconnect_a_long_split_is_permanent_and_with_it_heals(oracle):
    n = KW["n_nodes"]
    a = Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW))
    split_and_heal(a, n)
    st = a.stats()
    assert held_dead(a, n) == n // 2 and st["refutes"] == 0 and st["reconnects"] == 0
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, reconnect_interval_ms=5000, **KW))
    split_and_heal(b, n)
    st = b.stats()
    assert held_dead(b, n) == 0 and st["refutes"] == n and 0 < st["reconnects_reached"] < st["reconnects"]
    # every node is alive in everybody's eyes again: nothing is out of anybody's reach, and nobody is held dead
    assert b.detection(0)[0] == 0
    m = b.members(0, 7)
    assert all(int(x["status"]) == abi.MEMBER_ALIVE for x in m)


def test_the_gate_follows_failed_over_alive(oracle):
    """serf: prob = failed / alive — with one failed member in 256 a node tries about once in 255 intervals."""
    n = 256
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=n, seed=5, reconnect_interval_ms=1000, push_pull_interval_ms=0))
    s.step_ms(1000); s.kill(0, [9]); s.step_ms(40000)          # everybody holds 9 dead by now
    r0 = s.stats()["reconnects"]
    s.step_ms(100000)
    tried = s.stats()["reconnects"] - r0                         # 255 nodes x 100 intervals x 1/254
    assert 60 <= tried <= 150 and s.stats()["reconnects_reached"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [0, 512])
def test_reconnect_on_hip_matches_the_checker(hip, oracle, rows):
    n = KW["n_nodes"]
    a = Sim(hip, preset(hip, abi.PRESET_LAN, reconnect_interval_ms=5000, mass_rows=rows, **dict(KW, view_cap=8 if rows else 512)))
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, reconnect_interval_ms=5000, **KW))
    half = np.zeros(n, dtype=np.uint8); half[n // 2:] = 1
    for s in (a, b):
        s.step_ms(1000); s.partition(0, half)
    for sec in range(200):
        if sec == 140:
            for s in (a, b):
                s.partition(0, np.zeros(n, dtype=np.uint8))
        a.step_ms(1000); b.step_ms(1000)
        assert a.digest() == b.digest(), f"second {sec}"
    sa, sb = a.stats(), b.stats()
    for k in ("reconnects", "reconnects_reached", "refutes", "msgs_applied", "push_pulls", "edges", "packets_sent"):
        assert sa[k] == sb[k], k
    assert sb["refutes"] == n and a.detection(0) == b.detection(0) == (0, [0, 0, 0, 0])
