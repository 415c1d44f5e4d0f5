"""serf sizes a node's event queue max(2N, 4096) (internal/gossip/libserf/serf.go:22-27); the product library's holds 32 entries, the
checker's as many as serf's.  What the depth is worth (tools/event_queue_depth.py is the full experiment, profiles/r04_event_queue_depth.txt
its result): without churn every event reaches every stable observer with ANY depth — Prune() drops the copies that have been
transmitted most — and under config #5's churn the packets' byte budget (memberlist's own broadcasts first), not the queue, bounds
an event's reach while the flood lasts."""
import numpy as np

from consul_amd import abi
from consul_amd.sim import Sim, preset


def flood(lib, n, eq, churn, secs=8, E=20):
    kw = dict(n_nodes=n, seed=6, view_cap=n, queue_cap=16, event_queue_cap=eq, event_ids_per_ltime=62, inbox_cap=4096, subject_cap=4,
              flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, watch_node=abi.NONE)
    s = Sim(lib, preset(lib, abi.PRESET_LAN, **kw))
    rng = np.random.default_rng(6)
    dead = np.zeros(n, dtype=bool)
    stable = [int(x) for x in rng.choice(n, size=4, replace=False)]
    for w in stable:
        s.watch_events(0, w)
    churnable = np.setdiff1d(np.arange(n), stable)
    fired, got = [], {w: set() for w in stable}
    s.step_ms(1000)
    for sec in range(secs):
        k = int(n * churn)
        if k:
            flip = rng.choice(churnable, size=k, replace=False)
            kill, rev = flip[~dead[flip]], flip[dead[flip]]
            dead[flip] = ~dead[flip]
            if len(kill):
                s.kill(0, kill.tolist())
            if len(rev):
                s.revive(0, rev.tolist())
        live = np.flatnonzero(~dead)
        for tenth in range(10):
            for o in rng.choice(live, size=E // 10, replace=False):
                eid = int(rng.integers(1 << 30))
                fired.append((sec, eid, s.user_event(0, int(o), eid)))
            s.step_ms(100)
            for e in s.poll_events(65536):
                if e[2] == abi.EVENT_USER:
                    got[e[6]].add((e[3], e[4]))
    old = [(eid, lt) for (t, eid, lt) in fired if t < secs - 4]
    cov = [sum(1 for x in old if x in got[w]) / len(old) for w in stable]
    st = s.stats()
    return min(cov), st["event_drops"], st["queue_drops"]


def test_without_churn_every_depth_delivers_everything(oracle):
    shallow = flood(oracle, 1024, 16, 0.0)
    deep = flood(oracle, 1024, 2048, 0.0)              # serf's own size for this cluster: max(2N, 4096) capped at what the checker holds
    assert shallow[0] == deep[0] == 1.0
    assert shallow[1] > 10000 and deep[1] == 0         # thousands of pruned copies cost nothing; the deep queue prunes none


def test_the_checker_takes_serfs_depth_and_the_product_library_says_what_it_holds(oracle, hip):
    ok = preset(oracle, abi.PRESET_LAN, n_nodes=256, event_queue_cap=4096, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS)
    Sim(oracle, ok).close()
    bad = preset(oracle, abi.PRESET_LAN, n_nodes=256, event_queue_cap=8193, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS)
    d = abi.Derived()
    assert oracle.swim_config_derive(bad, d) != 0
    # the product library validates on the host: 32 is its limit (no GPU needed to hear it)
    h32 = preset(hip, abi.PRESET_LAN, n_nodes=256, queue_cap=4, event_queue_cap=32, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS)
    h33 = preset(hip, abi.PRESET_LAN, n_nodes=256, queue_cap=4, event_queue_cap=33, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS)
    assert hip.swim_config_derive(h32, d) == 0 and hip.swim_config_derive(h33, d) != 0


def test_under_churn_the_packets_not_the_queue_bound_an_events_reach(oracle):
    shallow = flood(oracle, 2048, 16, 0.10)
    deep = flood(oracle, 2048, 4096, 0.10)
    assert deep[1] == 0 and shallow[1] > 0 and shallow[2] == deep[2] > 0      # the memberlist queues overflow alike: the churn's own rumours
    assert shallow[0] < 1.0 and deep[0] < 1.0                                 # neither hands a stable observer everything while the flood lasts ...
    assert abs(shallow[0] - deep[0]) < 0.10                                   # ... and the 256-fold deeper queue changes little of it
