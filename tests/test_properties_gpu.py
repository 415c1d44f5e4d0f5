"""Protocol invariants and closed forms checked on the HIP library ALONE — no oracle in the loop.

HIP <-> oracle parity is bit-exact, but both were written by the same hand from the same reading of memberlist (the
reference's hot path is Go in two modules that are not in /root/reference and cannot be built here).  These tests pin the
product to things that do not depend on that reading:

  * invariants of SWIM / memberlist that hold for ANY legal execution (incarnations never go back, the dead do not come
    back without a higher incarnation, a live node that is suspected refutes, nothing is transmitted more often than
    retransmitLimit allows);
  * closed forms: first-detection time is the minimum of ~N uniform probe slots plus one probe interval, a suspicion
    without refutation runs at least suspicionTimeout(min), and a single rumour spreads like the push-gossip recurrence.
"""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

pytestmark = pytest.mark.gpu


def test_views_are_monotone_and_the_dead_stay_dead_without_a_new_incarnation(hip):
    """30 % packet loss without the TCP fallback: suspicions, refutations, a few real deaths.  Sampled every tick at
    eight observers: an observer's incarnation of a subject never decreases; Dead/Left -> Alive/Suspect only with a strictly
    higher incarnation; at equal incarnation the state only moves alive -> suspect -> dead."""
    n = 512
    s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, seed=23, loss_q32=int(0.30 * 2**32), view_cap=256, queue_cap=16, inbox_cap=256,
                        flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK))
    obs = [0, 17, 101, 255, 256, 300, 444, 511]
    prev = {o: s.members(0, o) for o in obs}
    rank = {abi.STATE_ALIVE: 0, abi.STATE_SUSPECT: 1, abi.STATE_DEAD: 2, abi.STATE_LEFT: 2}
    for t in range(300):
        if t == 50:
            s.kill(0, [33, 77])
        s.step(1)
        for o in obs:
            cur = s.members(0, o)
            inc0, inc1 = prev[o]["incarnation"].astype(np.int64), cur["incarnation"].astype(np.int64)
            assert (inc1 >= inc0).all(), f"tick {t} observer {o}: an incarnation went back"
            same = inc1 == inc0
            r0 = np.vectorize(rank.get)(prev[o]["state"]); r1 = np.vectorize(rank.get)(cur["state"])
            assert (r1[same] >= r0[same]).all(), f"tick {t} observer {o}: a state moved back at the same incarnation"
            prev[o] = cur
    st = s.stats()
    assert st["refutes"] > 0 and st["suspicion_timeouts"] > 0


def test_a_suspected_live_node_refutes_and_nothing_exceeds_the_retransmit_limit(hip):
    """Every live node that anybody suspects bumps its incarnation (refute), nobody is declared dead, and the suspicions fade
    well before suspicionTimeout(min); no queued rumour ever shows more transmits than retransmitLimit - 1."""
    n = 256
    s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, seed=29, loss_q32=int(0.35 * 2**32), view_cap=256, queue_cap=16, inbox_cap=256,
                        flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK))
    limit = s.derived.retransmit_limit
    s.step_ms(4000)
    suspected = set()
    for o in range(0, n, 8):
        m = s.members(0, o)
        suspected |= {int(x) for x in np.flatnonzero(m["state"] == abi.STATE_SUSPECT)}
    assert suspected, "35 % loss raised no suspicion"
    s.set_loss(0.0)
    for t in range(60):                                  # 6 s: far below suspicion min (9.6 s at N = 256)
        s.step(1)
        for i in range(0, n, 16):
            ni = s.node_info(0, i)
            assert all(ni.queue[j].transmits < limit for j in range(ni.queue_len))
    assert s.derived.suspicion_min_ms > 9000
    for x in suspected:
        assert s.node_info(0, x).incarnation >= 2, f"node {x} was suspected and never refuted"
    lingering = 0
    for o in range(0, n, 8):
        m = s.members(0, o)
        assert (m["state"] != abi.STATE_DEAD).all()
        lingering += int((m["state"] == abi.STATE_SUSPECT).sum())
    # gossip alone gives no hard guarantee that every refutation reaches every observer within 6 s (push-pull would): nearly all
    assert lingering <= 0.05 * (n // 8) * n
    assert s.stats()["suspicion_timeouts"] == 0


def test_detection_times_follow_the_closed_forms(hip):
    """BASELINE config #2 on 32 seeds (65 536 nodes, one failure at t = 5 s).  Independent of any oracle:
    first suspicion = (first probe of the victim after the kill) + ProbeInterval, and with ~1 probe of the victim per
    second cluster-wide the first probe is exponential with mean ~1 s; nobody refutes, so the first Dead verdict comes
    suspicionTimeout(min) = 19.264 s after the first suspicion at the earliest (k = 2 confirmations bring it down to the
    minimum), and everybody knows within a few gossip rounds of that."""
    n, reps = 65536, 32
    s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, n_replicas=reps, seed=1, subject_cap=2, view_cap=4, queue_cap=4, inbox_cap=24))
    assert s.derived.suspicion_min_ms == 19264 and s.derived.suspicion_max_ms == 115584 and s.derived.retransmit_limit == 20
    rng = np.random.default_rng(1)
    victims = [int(rng.integers(n)) for _ in range(reps)]
    s.step_ms(5000)
    for r, v in enumerate(victims):
        s.kill(r, [v])
    s.step_ms(35000); s.sync()
    c = [s.census(r, v) for r, v in enumerate(victims)]
    assert all(x.all_dead_ms != abi.NONE for x in c)
    first = np.array([x.first_suspect_ms - 5000 for x in c], dtype=float)
    # the failed probe's verdict falls one ProbeInterval after the ping (awareness 0): never earlier
    assert first.min() >= 1000
    wait = (first - 1000) / 1000.0                        # seconds until somebody's probe order reached the victim
    assert 0.45 <= wait.mean() <= 1.7, wait.mean()        # exponential(1 s): mean 1, sd of the mean of 32 samples 0.18
    assert wait.max() <= 8.0                              # P(exp(1) > 8) = 3e-4 per cluster
    gap = np.array([x.first_dead_ms - x.first_suspect_ms for x in c])
    assert gap.min() >= 19264 and gap.max() <= 19264 + 3000
    tail = np.array([x.all_dead_ms - x.first_dead_ms for x in c])
    assert 0 <= tail.min() and tail.max() <= 4000        # log4(65536) = 8 rounds of 200 ms + the retransmit tail


@pytest.mark.parametrize("k", [2, 3, 5])
def test_single_rumour_at_a_million_nodes_follows_the_push_gossip_recurrence(hip, k):
    """BASELINE config #3 at full size against I' = I + (N - I)(1 - (1 - 1/N)^(k I)): the simulated
    curve reaches 50 % and 99 % within one round of the recurrence.  gossip() alone (piggy-back off: the recurrence knows
    nothing of pings carrying rumours)."""
    n = 1 << 20
    s = Sim(hip, preset(hip, abi.PRESET_WAN, n_nodes=n, gossip_nodes=k, seed=5, trace_ticks=48, subject_cap=2, view_cap=2, queue_cap=4,
                        inbox_cap=32, flags=abi.F_DEFAULT & ~abi.F_PIGGYBACK))
    s.update(0, [0])
    s.step(40); s.sync()
    got = s.trace(0, 0, 0, 40)[:, 4].astype(float) + 1      # + the origin itself
    i, model = 1.0, []
    for _ in range(40):
        i = i + (n - i) * (1 - (1 - 1 / n) ** (k * i))
        model.append(i)
    model = np.array(model)
    for frac in (0.5, 0.99):
        assert abs(int(np.argmax(got >= frac * n)) - int(np.argmax(model >= frac * n))) <= 1, (k, frac)
    assert got[-1] == n and (np.diff(got) >= 0).all()
