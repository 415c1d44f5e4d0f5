"""Serf user events: Lamport clock, EventBuffer window, dedupe, rebroadcast.

Consul fires them through serf.UserEvent(name, payload, coalesce=false)
(agent/consul/server_ce.go:125-131) and consumes them at server_serf.go:283 / client_serf.go:98; the
reference's own test of this path is TestClientServer_UserEvent (agent/consul/client_test.go:756-830):
an event reaches every member exactly once."""
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset


def cluster(lib, **kw):
    base = dict(n_nodes=256, seed=4, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, watch_node=3)
    base.update(kw)
    return Sim(lib, preset(lib, abi.PRESET_LAN, **base))


def scenario(s):
    lts = [s.user_event(0, 10, 1001), s.user_event(0, 10, 1002), s.user_event(0, 200, 2001)]
    s.step_ms(4000)
    lts.append(s.user_event(0, 77, 3001))
    s.step_ms(4000)
    return lts


def test_event_reaches_every_member_exactly_once(oracle):
    s = cluster(oracle)
    lts = scenario(s)
    assert lts[0] == 1 and lts[1] == 2 and lts[2] == 1          # Lamport time: per-origin clock at fire time (serf.Create starts it at 1)
    assert lts[3] >= 3                                          # node 77 witnessed ltime 1 before firing
    st = s.stats()
    assert st["user_events_delivered"] == 4 * 256               # each event delivered once per member
    assert st["user_events_deduped"] > 0                        # ...however many copies arrived
    ev = [e for e in s.poll_events() if e[2] == abi.EVENT_USER]
    assert sorted(e[3] for e in ev) == [1001, 1002, 2001, 3001]  # the watched member saw each once
    for i in range(256):
        ni = s.node_info(0, i)
        assert ni.event_clock >= lts[3] + 1 and ni.event_queue_len == 0   # witnessed, queues drained


def test_events_share_the_packet_with_membership_rumours(oracle):
    """memberlist getBroadcasts: system queue first, then the delegate's user messages in what is left."""
    s = cluster(oracle, udp_buffer_size=2 + (2 + 48) + (3 + 64))   # room for one suspect/dead + one event
    s.kill(0, [9])
    s.step_ms(3000)
    s.user_event(0, 1, 42)
    s.step_ms(6000)
    st = s.stats()
    assert st["user_events_delivered"] == 255                   # the dead node never hears it
    assert st["msgs_sent"][abi.MSG_USER] > 0 and st["msgs_sent"][abi.MSG_SUSPECT] > 0


def test_old_events_fall_out_of_the_window(oracle):
    """handleUserEvent drops an event whose LTime is more than EventBuffer behind the local clock."""
    s = cluster(oracle, n_nodes=64, event_buffer=4, event_queue_cap=16)
    for i in range(12):                                         # a burst: LTimes 1..12 leave node 5 together
        assert s.user_event(0, 5, 100 + i) == i + 1
    s.step_ms(6000)
    st = s.stats()
    # the whole burst rides one packet and is applied in ascending LTime order, so a first copy is always
    # inside the window; every LATER copy of an early event finds the clock >4 ahead and is dropped as stale
    assert st["user_events_stale"] > 0
    assert st["user_events_delivered"] == 12 * 64


@pytest.mark.gpu
def test_user_events_hip_matches_oracle(hip, oracle):
    a, b = cluster(hip, n_nodes=4096, n_replicas=2), cluster(oracle, n_nodes=4096, n_replicas=2)
    for s in (a, b):
        lts = scenario(s)
        s.kill(1, [50])
        s.user_event(1, 3, 777)
        s.step_ms(8000)
    a.sync()
    assert a.digest() == b.digest()
    sa, sb = a.stats(), b.stats()
    for k in ("user_events_delivered", "user_events_deduped", "user_events_stale", "event_drops", "msgs_sent",
              "packets_sent", "edges", "msgs_applied"):
        assert sa[k] == sb[k], k
    assert a.poll_events() == b.poll_events()
    assert a.node_info(0, 100).event_clock == b.node_info(0, 100).event_clock


def test_a_user_event_id_is_never_read_as_an_intent(oracle):
    """force-leave intents share the broadcast queue with user events and are told apart by bits 31-30 of the id
    word.  An application id with those bits set used to vanish (no delivery) and to force-leave — even prune — whatever member its
    low bits named.  Ids are 30 bits now: anything larger is refused, never reinterpreted."""
    from consul_amd.sim import SwimError
    s = cluster(oracle, n_nodes=64)
    s.kill(0, [3]); s.step_ms(40000)                         # node 3 is Failed everywhere
    assert s.view(0, 9, 3).status == abi.MEMBER_FAILED
    for bad in (0x80000003, 0xC0000003, 0x92345678, 0x40000000):
        with pytest.raises(SwimError) as e:
            s.user_event(0, 10, bad)
        assert e.value.rc == abi.ERANGE
    s.step_ms(5000)
    assert s.view(0, 9, 3).status == abi.MEMBER_FAILED and s.stats()["intents_applied"] == 0 and s.stats()["reaped"] == 0
    lt = s.user_event(0, 10, 0x3FFFFFFF)                     # the largest id there is: an ordinary event
    s.step_ms(5000)
    assert lt != abi.NONE and s.stats()["user_events_delivered"] == 63 and s.view(0, 9, 3).status == abi.MEMBER_FAILED
    s.force_leave(0, 10, 3)                                  # the real thing still works
    s.step_ms(5000)
    assert s.view(0, 9, 3).status == abi.MEMBER_LEFT


@pytest.mark.gpu
def test_event_id_range_on_hip(hip):
    from consul_amd.sim import SwimError
    s = cluster(hip, n_nodes=64)
    with pytest.raises(SwimError) as e:
        s.user_event(0, 10, 0x80000003)
    assert e.value.rc == abi.ERANGE
    assert s.user_event(0, 10, 0x3FFFFFFF) != abi.NONE
