"""serf's intent ordering (VERDICT r3 missing 4): every member entry carries a statusLTime, a leave or join intent stamped
no later than it is ignored, a member answers a leave intent about ITSELF with a join intent (the refutation), a newer join intent takes a
Leaving mark back, and serf.Join broadcasts one.  Upstream: serf v0.10.4 serf.go handleNodeLeaveIntent / handleNodeJoinIntent / Leave / Join
(absent from /root/reference: go.mod:85); Consul's consumers: agent/consul/server_serf.go:270-297 (lanNodeFailed / left), agent/agent.go Leave.
The checker here; tests/test_serf_intents_gpu.py runs the same scripts on the HIP library beside it."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

KW = dict(n_nodes=256, seed=31, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, subject_cap=16, view_cap=64, queue_cap=16, event_queue_cap=16,
          inbox_cap=256, push_pull_interval_ms=0)


def statuses(s, x, observers):
    return [int(s.view(0, o, x).status) for o in observers]


def script_refuted_leave(s):
    """A leave intent about a member that is alive and not leaving: everybody marks it Leaving — until the member hears of it and
    refutes (broadcastJoin); then it is Alive again everywhere, and a later failure reads as Failed, not Left."""
    out = {}
    s.step_ms(2000)
    s.force_leave(0, 3, 77, False)                       # node 3 says: 77 is leaving (it is not)
    s.step_ms(200)
    out["early"] = statuses(s, 77, (3,))
    s.step_ms(6000)
    out["after"] = statuses(s, 77, range(0, 256, 17))
    out["self"] = int(s.view(0, 77, 77).status)
    s.kill(0, [77]); s.step_ms(40000)
    out["failed"] = statuses(s, 77, range(0, 256, 17))
    out["digest"] = s.digest()
    return out


def test_a_leave_intent_about_a_live_member_is_refuted(oracle):
    o = script_refuted_leave(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    assert o["early"] == [abi.MEMBER_LEAVING]
    assert set(o["after"]) == {abi.MEMBER_ALIVE} and o["self"] == abi.MEMBER_ALIVE
    assert set(o["failed"]) == {abi.MEMBER_FAILED}        # (a member still marked Leaving would have become Left)


def script_stale_leave(s):
    """Node 9 is cut off, issues a leave intent about 77 on its old clock, and comes back after the pool has applied a NEWER join intent
    of 77 (a refutation): the stale intent changes nothing, anywhere."""
    out = {}
    s.step_ms(2000)
    cut = np.zeros(256, dtype=np.uint8); cut[9] = 1
    s.partition(0, cut)
    s.force_leave(0, 3, 77, False); s.step_ms(8000)       # leave intent + 77's refutation spread in the majority: statusLTime(77) = the refutation's
    out["lt_old"] = s.force_leave(0, 9, 77, False)        # 9 has heard of neither: its clock is behind
    s.partition(0, np.zeros(256, dtype=np.uint8)); s.step_ms(10000)
    out["after"] = statuses(s, 77, range(0, 256, 17))
    out["at9"] = int(s.view(0, 9, 77).status)
    out["digest"] = s.digest()
    return out


def test_a_stale_leave_intent_is_ignored(oracle):
    o = script_stale_leave(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    assert set(o["after"]) == {abi.MEMBER_ALIVE}
    # 9 applied its own intent and came back after the refutation had finished its transmissions: it keeps the mark until a state exchange
    # tells it better — serf's push-pull hands statusLTimes over as join intents; this simulator's push-pull carries memberlist's state only
    #  — or until 77 hears 9's copy... which it does: 9 gossips its intent to 77 among others, 77 refutes again with a newer join.
    assert o["at9"] in (abi.MEMBER_LEAVING, abi.MEMBER_ALIVE)


def script_rejoin_after_force_leave(s):
    """TestAgent_ForceLeave's sequel (agent/agent_endpoint_test.go:2524-2566): a failed member is force-left, then the process comes back
    and joins — its join intent is newer than the leave intent (serf.Join witnesses the pool's clock first), so it is a member again and
    copies of the old leave intent that still travel do not mark it Leaving."""
    out = {}
    s.step_ms(2000); s.kill(0, [77]); s.step_ms(30000)
    out["failed"] = statuses(s, 77, (0, 100, 200))
    out["lt_leave"] = s.force_leave(0, 3, 77, False); s.step_ms(4000)
    out["left"] = statuses(s, 77, (0, 100, 200))
    s.join(0, [77], via=5); s.step_ms(500)
    s.force_leave(0, 3, 77, False)                        # hmm: a NEW intent, newer than the join: this one does apply
    s.step_ms(300)
    out["leaving_again"] = statuses(s, 77, (3,))
    s.step_ms(10000)
    out["end"] = statuses(s, 77, range(0, 256, 17))       # ... and is refuted by 77 itself
    out["digest"] = s.digest()
    return out


def test_rejoin_after_force_leave(oracle):
    o = script_rejoin_after_force_leave(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    assert set(o["failed"]) == {abi.MEMBER_FAILED} and set(o["left"]) == {abi.MEMBER_LEFT}
    assert o["leaving_again"] == [abi.MEMBER_LEAVING]
    assert set(o["end"]) == {abi.MEMBER_ALIVE}


def script_graceful_leave(s):
    """serf.Leave(): the member's own leave intent first (everybody: Leaving; nobody refutes, the member least of all), then memberlist's
    leave (dead{Node == From}): Left everywhere, never Failed."""
    out = {}
    s.step_ms(2000)
    out["lt"] = s.force_leave(0, 77, 77, False)           # node == origin: Leave()'s intent
    s.step_ms(3000)
    out["leaving"] = statuses(s, 77, range(0, 256, 17))
    s.leave(0, [77]); s.step_ms(3000)
    out["left"] = statuses(s, 77, [o for o in range(0, 256, 17) if o != 77])
    out["digest"] = s.digest()
    return out


def test_graceful_leave_broadcasts_its_intent_first(oracle):
    o = script_graceful_leave(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    assert set(o["leaving"]) == {abi.MEMBER_LEAVING}
    assert set(o["left"]) == {abi.MEMBER_LEFT}


JOIN_KW = dict(n_nodes=1024, n_initial=1000, seed=17, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, subject_cap=64, view_cap=256, queue_cap=16,
               event_queue_cap=16, inbox_cap=1024, push_pull_interval_ms=0)


def script_joins(s):
    """A failed member is force-left and comes back through a node of the OTHER half of the id space (another shard, when there are two);
    then two members that start late, one via each half.  Returns (digest, a few statuses) after every step."""
    out = []

    def mark():
        s.sync()
        out.append((s.digest(), int(s.view(0, 10, 700).status), int(s.view(0, 900, 700).status), int(s.view(0, 10, 1011).status)))
    s.step_ms(2000); mark()
    s.kill(0, [700]); s.step_ms(30000); mark()
    s.force_leave(0, 3, 700, False); s.step_ms(4000); mark()
    s.join(0, [700], via=5); s.step_ms(6000); mark()
    s.join(0, [1010], via=600); s.step_ms(6000); mark()
    s.join(0, [1011], via=2); s.step_ms(6000); mark()
    return out


def test_joins_with_intents_on_the_checker(oracle):
    o = script_joins(Sim(oracle, preset(oracle, abi.PRESET_LAN, **JOIN_KW)))
    assert o[2][1] == abi.MEMBER_LEFT and o[3][1] == abi.MEMBER_ALIVE and o[3][2] == abi.MEMBER_ALIVE and o[5][3] == abi.MEMBER_ALIVE


def event_queued_tracks_one_broadcast(lib):
    """swim_event_queued (round 5): serf.Leave() waits on the notify channel of ITS intent — under steady user-event traffic the
    queue never empties, the intent still retires once it has used up its transmissions."""
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=64, seed=4, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, event_queue_cap=16))
    s.step_ms(1000)
    lt = s.force_leave(0, 9, 9)                                  # node 9's own leave intent (serf.Leave step 1)
    intent = abi.INTENT_LEAVE | 9
    assert s.event_queued(0, 9, intent, lt) and not s.event_queued(0, 9, intent, lt + 1) and not s.event_queued(0, 10, intent, lt)
    retired_at, quiet_queue_seen = None, False
    for tick in range(1, 120):
        s.user_event(0, 9, 1000 + tick)                           # steady traffic from the same node: its queue never runs empty
        s.step(1)
        if not s.event_queued(0, 9, intent, lt):
            retired_at = tick
            break
        quiet_queue_seen |= s.node_info(0, 9).event_queue_len == 0
    assert retired_at is not None and not quiet_queue_seen and s.node_info(0, 9).event_queue_len > 0
    s.close()


def test_event_queued_tracks_one_broadcast_on_the_checker(oracle):
    event_queued_tracks_one_broadcast(oracle)


@pytest.mark.gpu
def test_event_queued_tracks_one_broadcast_on_hip(hip):
    event_queued_tracks_one_broadcast(hip)
