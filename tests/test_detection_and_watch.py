"""swim_detection_get (BASELINE config #4's deliverable) against its own definition, pair by pair; swim_watch_events: an EventCh per
agent."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, SwimError, preset


def brute_force(s, n, down, group):
    """over all ordered pairs (o, x != o), o running, x down or in another group: o's view of x by state — through swim_view"""
    pairs, by = 0, [0, 0, 0, 0]
    for o in range(n):
        if o in down:
            continue
        for x in range(n):
            if x == o or not (x in down or group[x] != group[o]):
                continue
            pairs += 1
            by[s.view(0, o, x).state] += 1
    return pairs, by


def detection_scenario(lib):
    n = 96
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=n, seed=6, view_cap=n, fold_interval_ms=20000))
    group = np.zeros(n, dtype=np.uint8); group[80:] = 1
    down = {3, 40, 41, 90}
    s.step_ms(1000); s.kill(0, sorted(down)); s.partition(0, group)
    seen = []
    for sec in range(0, 70, 7):
        s.step_ms(7000)
        got = s.detection(0)
        assert got == brute_force(s, n, down, group), f"t = {sec + 8} s"
        seen.append(got)
    assert seen[0][1][abi.STATE_DEAD] < seen[-1][1][abi.STATE_DEAD] and seen[-1][0] == seen[0][0]
    # 92 running observers: 76 on the majority side (each cannot reach 16 + 4 - 1 = 19: the other side incl. its dead one, and the
    # 3 dead of its own), 15 on the minority side (80 + 1 - ... counted by the brute force above); the closed form must agree
    return seen


def test_detection_census_equals_its_definition_on_the_checker(oracle):
    detection_scenario(oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [0, 96])
def test_detection_census_equals_its_definition_on_hip(hip, oracle, rows):
    n = 96
    a = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, seed=6, view_cap=n, fold_interval_ms=20000, mass_rows=rows))
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=n, seed=6, view_cap=n, fold_interval_ms=20000))
    group = np.zeros(n, dtype=np.uint8); group[80:] = 1
    for s in (a, b):
        s.step_ms(1000); s.kill(0, [3, 40, 41, 90]); s.partition(0, group)
    for sec in range(0, 70, 7):
        a.step_ms(7000); b.step_ms(7000)
        assert a.detection(0) == b.detection(0) and a.digest() == b.digest()
    assert a.detection(0) == brute_force(a, n, {3, 40, 41, 90}, group)


def watchers(lib):
    """Two agents with an EventCh of their own besides the watch node: each gets the failure and the graceful leave once, in order,
    tagged with its id; polling returns the streams sorted by (time, observer)."""
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=64, seed=8, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, watch_node=2))
    s.watch_events(0, 30); s.watch_events(0, 7); s.watch_events(0, 30)          # (twice is once)
    s.step_ms(1000); s.kill(0, [11]); s.leave(0, [12]); s.user_event(0, 20, 777)
    s.step_ms(40000)
    ev = s.poll_events(65536)
    by_obs = {}
    for e in ev:
        by_obs.setdefault(e[6], []).append(e)
    assert sorted(by_obs) == [2, 7, 30]
    for o, evs in by_obs.items():
        kinds = [(e[2], e[3]) for e in evs]
        assert kinds.count((abi.EVENT_MEMBER_FAILED, 11)) == 1 and kinds.count((abi.EVENT_MEMBER_LEAVE, 12)) == 1 and kinds.count((abi.EVENT_USER, 777)) == 1
        assert [e[0] for e in evs] == sorted(e[0] for e in evs)                   # each observer's own events in time order
    assert [(e[0], e[6]) for e in ev] == sorted((e[0], e[6]) for e in ev)         # the stream: by time, then observer
    return ev


def test_an_event_channel_per_agent_on_the_checker(oracle):
    watchers(oracle)
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=256, seed=8))
    for i in range(1, 65):
        s.watch_events(0, i)                                                       # SWIM_EVENT_WATCHERS = 64 besides the watch node (0)
    with pytest.raises(SwimError) as e:
        s.watch_events(0, 100)
    assert e.value.rc == abi.EOVERFLOW
    with pytest.raises(SwimError) as e:
        s.watch_events(0, 256)
    assert e.value.rc == abi.ERANGE


@pytest.mark.gpu
def test_an_event_channel_per_agent_on_hip_matches_the_checker(hip, oracle):
    assert watchers(hip) == watchers(oracle)
