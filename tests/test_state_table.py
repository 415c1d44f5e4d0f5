"""memberlist's state machine as a TABLE (state.go aliveNode / suspectNode / deadNode), in the manner of
upstream's own state_test.go cases (TestMemberList_AliveNode_*, _SuspectNode_*, _DeadNode_*): an observer is put into every
state of a subject at incarnation 5, one message of every kind arrives with an older, the same or a newer incarnation, and the
resulting view is compared with what the published rules say.  The messages come in through the transport bridge
(swim_transport_write_to: a "real" node attached as member 0 writes packets to the observer), so this pins the handlers
themselves on BOTH libraries, not an end-to-end outcome.  The observer sits alone behind a partition with a probe interval of
100 s: nothing but the injected messages reaches it or leaves it while a case runs."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

A, S, D, L = abi.STATE_ALIVE, abi.STATE_SUSPECT, abi.STATE_DEAD, abi.STATE_LEFT
N, OBS, SUBJ, REAL, ACCUSER = 16, 1, 5, 0, 9

# (state before, message kind, incarnation of the message relative to 5) -> (state after, incarnation after)
TABLE = {
    # aliveNode: only a strictly newer incarnation is news; it clears a suspicion and brings back the dead
    (A, "alive", -1): (A, 5), (A, "alive", 0): (A, 5), (A, "alive", +1): (A, 6),
    (S, "alive", -1): (S, 5), (S, "alive", 0): (S, 5), (S, "alive", +1): (A, 6),
    (D, "alive", -1): (D, 5), (D, "alive", 0): (D, 5), (D, "alive", +1): (A, 6),
    (L, "alive", -1): (L, 5), (L, "alive", 0): (L, 5), (L, "alive", +1): (A, 6),
    # suspectNode: an older incarnation is ignored; an equal or newer one suspects an Alive member; a running timer only takes the
    # confirmation (the incarnation stays); a Dead / Left member is not suspected, whatever the incarnation (the order dependence
    # SURVEY §7 hard part 5 names)
    (A, "suspect", -1): (A, 5), (A, "suspect", 0): (S, 5), (A, "suspect", +1): (S, 6),
    (S, "suspect", -1): (S, 5), (S, "suspect", 0): (S, 5), (S, "suspect", +1): (S, 5),
    (D, "suspect", -1): (D, 5), (D, "suspect", 0): (D, 5), (D, "suspect", +1): (D, 5),
    (L, "suspect", -1): (L, 5), (L, "suspect", 0): (L, 5), (L, "suspect", +1): (L, 5),
    # deadNode: an older incarnation is ignored; otherwise Alive and Suspect become Dead; Dead / Left stay what they are
    (A, "dead", -1): (A, 5), (A, "dead", 0): (D, 5), (A, "dead", +1): (D, 6),
    (S, "dead", -1): (S, 5), (S, "dead", 0): (D, 5), (S, "dead", +1): (D, 6),
    (D, "dead", -1): (D, 5), (D, "dead", 0): (D, 5), (D, "dead", +1): (D, 5),
    (L, "dead", -1): (L, 5), (L, "dead", 0): (L, 5), (L, "dead", +1): (L, 5),
    # dead{Node == From}: a graceful leave
    (A, "leave", 0): (L, 5), (S, "leave", +1): (L, 6), (D, "leave", +1): (D, 5),
}
KIND = {"alive": (abi.MSG_ALIVE, 0), "suspect": (abi.MSG_SUSPECT, ACCUSER), "dead": (abi.MSG_DEAD, ACCUSER), "leave": (abi.MSG_DEAD, SUBJ)}


def observer_in_state(lib, before):
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=N, seed=2, probe_interval_ms=100000, probe_timeout_ms=500, push_pull_interval_ms=0, view_cap=N))
    mask = np.zeros(N, dtype=np.uint8); mask[OBS] = 1
    s.partition(0, mask)
    def write(kind, inc):
        typ, frm = KIND[kind]
        s.transport_write_to(0, REAL, OBS, [(SUBJ, inc, typ, frm)]); s.step(1)
    write("alive", 5)
    if before == S: write("suspect", 5)
    if before == D: write("dead", 5)
    if before == L: write("leave", 5)
    v = s.view(0, OBS, SUBJ)
    assert (v.state, v.incarnation) == (before, 5)
    return s, write


def run_table(lib):
    for (before, kind, rel), want in sorted(TABLE.items()):
        s, write = observer_in_state(lib, before)
        write(kind, 5 + rel)
        v = s.view(0, OBS, SUBJ)
        assert (v.state, v.incarnation) == want, f"{('alive', 'suspect', 'dead', 'left')[before]}@5 + {kind}@{5 + rel}: got state {v.state} inc {v.incarnation}, want {want}"
        s.close()


def test_state_table_on_the_checker(oracle):
    run_table(oracle)


@pytest.mark.gpu
def test_state_table_on_hip(hip):
    run_table(hip)


def confirmations(lib):
    """suspicion.Confirm: a new accuser counts once, the first accuser and a repeated one do not, and after k = SuspicionMult - 2
    confirmations nothing counts any more."""
    s, write = observer_in_state(lib, S)                       # suspected by ACCUSER
    def accuse(frm):
        s.transport_write_to(0, REAL, OBS, [(SUBJ, 5, abi.MSG_SUSPECT, frm)]); s.step(1)
        return s.view(0, OBS, SUBJ).n_confirm
    assert accuse(ACCUSER) == 0                                # the first accuser again
    assert accuse(10) == 1 and accuse(10) == 1                 # a new one counts, once
    assert accuse(11) == 2                                     # k = 2 at SuspicionMult 4 ...
    assert accuse(12) == 2                                     # ... reached: the timer is at its minimum, nobody else counts
    s.close()


def test_confirmations_on_the_checker(oracle):
    confirmations(oracle)


@pytest.mark.gpu
def test_confirmations_on_hip(hip):
    confirmations(hip)


def refutation(lib):
    """A node that hears it is suspected or dead at incarnation >= its own refutes with the next one (skipping past the accuser's);
    an older accusation is ignored; an alive about itself with a newer incarnation (somebody else's memory of an earlier life) too."""
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=N, seed=2, probe_interval_ms=100000, probe_timeout_ms=500, push_pull_interval_ms=0))
    mask = np.zeros(N, dtype=np.uint8); mask[OBS] = 1
    s.partition(0, mask)
    def tell(typ, inc, frm=ACCUSER):
        s.transport_write_to(0, REAL, OBS, [(OBS, inc, typ, frm)]); s.step(1)
        return s.node_info(0, OBS).incarnation
    assert tell(abi.MSG_SUSPECT, 0) == 1                       # older than its own incarnation 1: ignored
    assert tell(abi.MSG_SUSPECT, 1) == 2                       # nextIncarnation
    assert tell(abi.MSG_DEAD, 7) == 8                          # skipIncarnation past the accuser's
    assert tell(abi.MSG_ALIVE, 8, 0) == 8 and tell(abi.MSG_ALIVE, 11, 0) == 12
    assert s.stats()["refutes"] == 3 and s.node_info(0, OBS).awareness == 3      # Lifeguard: having to refute costs a point each time
    s.close()


def test_refutation_on_the_checker(oracle):
    refutation(oracle)


@pytest.mark.gpu
def test_refutation_on_hip(hip):
    refutation(hip)


def merge_state(lib):
    """After upstream's TestMemberList_MergeState (state_test.go): three members alive at incarnation 1, the first of them suspected;
    a remote state list arrives that holds the first ALIVE at incarnation 2, the second SUSPECT at 1, the third DEAD at 1 and a fourth,
    unknown member alive at 2.  mergeState turns the list into aliveNode / suspectNode calls — a remote Dead is never adopted: it
    becomes a suspicion from the receiver itself — so afterwards the first is alive at 2 (the newer incarnation clears the
    suspicion), the second and the third are suspect at 1, the fourth has joined at 2, and the receiver's EventCh has exactly one
    NodeJoin, the fourth's.  The list comes in as ONE packet through the transport bridge, the way BridgeTransport::PushPull hands it over."""
    t1, t2, t3, t4 = 5, 6, 7, N - 1
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=N, n_initial=N - 1, seed=2, probe_interval_ms=100000, probe_timeout_ms=500, push_pull_interval_ms=0,
                        view_cap=N, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, watch_node=OBS))
    mask = np.zeros(N, dtype=np.uint8); mask[OBS] = 1
    s.partition(0, mask)
    s.transport_write_to(0, REAL, OBS, [(t1, 1, abi.MSG_SUSPECT, ACCUSER)]); s.step(1)
    assert (s.view(0, OBS, t1).state, s.view(0, OBS, t4).state) == (S, D)      # (a node nobody has heard of reads as the base row: Dead at incarnation 0)
    s.poll_events()
    s.transport_write_to(0, REAL, OBS, [(t1, 2, abi.MSG_ALIVE, 0), (t2, 1, abi.MSG_SUSPECT, OBS), (t3, 1, abi.MSG_SUSPECT, OBS), (t4, 2, abi.MSG_ALIVE, 0)])
    s.step(1)
    got = {x: (s.view(0, OBS, x).state, s.view(0, OBS, x).incarnation) for x in (t1, t2, t3, t4)}
    assert got == {t1: (A, 2), t2: (S, 1), t3: (S, 1), t4: (A, 2)}, got
    joins = [e for e in s.poll_events() if e[2] == abi.EVENT_MEMBER_JOIN]
    assert [(e[3], e[6]) for e in joins] == [(t4, OBS)], joins
    s.close()


def test_merge_state_on_the_checker(oracle):
    merge_state(oracle)


@pytest.mark.gpu
def test_merge_state_on_hip(hip):
    merge_state(hip)


def queue_prune_and_invalidation(lib):
    """TransmitLimitedQueue as upstream's queue_test.go pins it (recalled: TestTransmitLimited_Prune, TestTransmitLimited_QueueBroadcast), on the
    queue of ONE node: with room for two broadcasts, four queued one after the other leave the two NEWEST ("Prune(2)": 'test' and 'foo' go,
    'bar' and 'baz' stay); a broadcast about a node invalidates the older one about the same node and joins the queue at its end; what is
    picked for a packet is what has been transmitted least, newest first.  The messages arrive in one packet through the transport bridge and
    are applied in subject order, so the "newest" of a packet is the highest subject."""
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=N, seed=2, probe_interval_ms=100000, probe_timeout_ms=500, push_pull_interval_ms=0, view_cap=N, queue_cap=2))
    mask = np.zeros(N, dtype=np.uint8); mask[OBS] = 1
    s.partition(0, mask)                                       # (nothing but the injected packets reaches the observer, nothing it gossips arrives anywhere)
    queue = lambda: [(q.subject, q.incarnation, q.type) for q in list(s.node_info(0, OBS).queue)[:s.node_info(0, OBS).queue_len]]
    s.transport_write_to(0, REAL, OBS, [(x, 6, abi.MSG_ALIVE, 0) for x in (5, 6, 7, 8)]); s.step(1)
    assert queue() == [(7, 6, abi.MSG_ALIVE), (8, 6, abi.MSG_ALIVE)] and s.stats()["queue_drops"] == 2
    s.close()
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=N, seed=2, probe_interval_ms=100000, probe_timeout_ms=500, push_pull_interval_ms=0, view_cap=N, queue_cap=4))
    s.partition(0, mask)
    s.transport_write_to(0, REAL, OBS, [(5, 6, abi.MSG_ALIVE, 0), (6, 6, abi.MSG_ALIVE, 0)]); s.step(1)
    assert queue() == [(5, 6, abi.MSG_ALIVE), (6, 6, abi.MSG_ALIVE)]
    s.transport_write_to(0, REAL, OBS, [(5, 6, abi.MSG_SUSPECT, ACCUSER)]); s.step(1)       # about node 5 again: the alive about it is invalidated
    assert queue() == [(6, 6, abi.MSG_ALIVE), (5, 6, abi.MSG_SUSPECT)] and s.stats()["queue_drops"] == 0
    s.close()


def test_queue_prune_and_invalidation_on_the_checker(oracle):
    queue_prune_and_invalidation(oracle)


@pytest.mark.gpu
def test_queue_prune_and_invalidation_on_hip(hip):
    queue_prune_and_invalidation(hip)
