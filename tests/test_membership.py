"""Membership that changes: nodes nobody has heard of, serf.Join, the join push-pull, restarts, and
estNumNodes() feeding retransmitLimit / suspicionTimeout — on the CPU oracle.  GPU parity: tests/test_scale_gpu.py."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset


def lan(oracle, **kw):
    return Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))


def grow(s, n, first, gap_ms=200):
    """nodes first..n-1 join one after the other, each through the node that joined before it (the first ones through 0..2)"""
    for x in range(first, n):
        s.join(0, [x], via=x % first if x < first + 7 else x - 1)
        s.step_ms(gap_ms)


def test_cluster_grows_from_three_by_joins(oracle):
    """server.go:1461 / client.go:222 shape: 3 members, 61 serf.Join calls; everybody ends up knowing everybody, the watch
    node saw one EventMemberJoin per joiner, and every new member ends up in the base row (folded)."""
    n = 64
    s = lan(oracle, n_nodes=n, n_initial=3, seed=2, view_cap=64, inbox_cap=256, fold_interval_ms=2000, watch_node=0)
    m = s.members(0, 0)
    assert [int(x) for x in m["status"][:4]] == [abi.MEMBER_ALIVE] * 3 + [abi.MEMBER_NONE]     # never heard of node 3
    s.step_ms(1000)
    grow(s, n, 3)
    s.step_ms(20000)
    st = s.stats()
    assert st["joins"] == n - 3 and st["join_failures"] == 0 and st["folds"] == n - 3 and st["view_drops"] == 0
    assert st["msgs_applied"][abi.MSG_ALIVE] >= (n - 3) * 3                # (a later joiner finds earlier ones in the base row already)
    for o in (0, 30, n - 1):
        m = s.members(0, o)
        assert (m["status"] == abi.MEMBER_ALIVE).all() and (m["incarnation"] == 1).all()
    ev = [e for e in s.poll_events() if e[2] == abi.EVENT_MEMBER_JOIN]
    assert sorted(e[3] for e in ev) == list(range(3, n))


def test_est_num_nodes_scales_the_timers_like_upstream(oracle):
    """suspicionTimeout(mult, n, interval) with n = the observer's own member count: in a 4-member cluster (of a 64-node id
    space) a failure is declared after max = 6 x 4 s at the latest and min = 4 s at the earliest; the same failure in the
    fixed 64-member population takes 4 x log10(64) = 7.224 s at least."""
    small = lan(oracle, n_nodes=64, n_initial=4, seed=3, watch_node=0)
    full = lan(oracle, n_nodes=64, seed=3, watch_node=0)
    assert small.derived.suspicion_min_ms == full.derived.suspicion_min_ms == 7224          # (derived = the fixed population's)
    for s in (small, full):
        s.step_ms(2000); s.kill(0, [2]); s.step_ms(60000)
    cs, cf = small.census(0, 2), full.census(0, 2)
    assert cs.all_dead_ms != abi.NONE and cf.all_dead_ms != abi.NONE
    assert 4000 <= cs.first_dead_ms - cs.first_suspect_ms <= 24000
    assert cf.first_dead_ms - cf.first_suspect_ms >= 7224
    # retransmitLimit = 4 * ceil(log10(n+1)): 4 transmissions per rumour with 4 members, 8 with 64
    assert small.stats()["msgs_sent"][abi.MSG_SUSPECT] < full.stats()["msgs_sent"][abi.MSG_SUSPECT]


def test_restart_comes_back_with_a_higher_incarnation(oracle):
    """config #5's "kill / rejoin with incarnation bump" on a fixed population: the restarted process knows nothing of its
    own (it holds the base row), announces alive@2, and does the join push-pull."""
    s = lan(oracle, n_nodes=256, seed=4, fold_interval_ms=2000, push_pull_interval_ms=0)
    s.step_ms(1000); s.kill(0, [9]); s.step_ms(25000)
    assert s.census(0, 9).by_state[abi.STATE_DEAD] == 255
    s.join(0, [9], via=100)
    assert s.node_info(0, 9).incarnation == 2 and s.node_info(0, 9).queue_len == 1
    s.step_ms(20000)
    c = s.census(0, 9)
    assert c.by_state[abi.STATE_ALIVE] == 255 and c.n_current == 255
    st = s.stats()
    assert st["joins"] == 1 and st["refutes"] == 0 and s.view(0, 200, 9).incarnation == 2


def test_join_through_an_unreachable_member_fails(oracle):
    s = lan(oracle, n_nodes=32, n_initial=8, seed=5)
    s.step_ms(500); s.kill(0, [3])
    s.join(0, [20], via=3)            # via is down: memberlist.Join returns an error, the node is up but alone
    s.join(0, [21], via=21)           # via = itself
    s.step_ms(5000)
    st = s.stats()
    assert st["joins"] == 0 and st["join_failures"] == 2
    assert s.view(0, 0, 20).status == abi.MEMBER_NONE and s.node_info(0, 20).alive == 1
    with pytest.raises(Exception):
        lan(oracle, n_nodes=32, n_initial=33)
