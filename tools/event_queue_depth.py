#!/usr/bin/env python3
"""How deep must a node's serf event queue be?  (VERDICT r3 missing 2 / weak 6: config #5's leg delivers an event to 37 % of the nodes
and drops 4 M queued events — is that the 16-entry queue, where serf holds max(2N, 4096), internal/gossip/libserf/serf.go:22-27?)

Runs on the CHECKER, whose event queue can be as deep as serf's (the product library's holds 32): config #5's shape — 10 %/s churn
(kill / revive) and 20 user events/s for 20 s, then 20 quiet seconds — with four STABLE observers (never killed) whose EventCh is read like a
consumer would; reported: which share of the events fired at least 5 s before the end of the flood each of them was handed, at the end of
the flood and after the quiet tail, for several queue depths, with and without the churn.  Usage: event_queue_depth.py [n_nodes]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consul_amd import abi  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402

ora = abi.bind(ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))


def run(n, eq, churn, secs=20, tail=20, E=20):
    kw = dict(n_nodes=n, seed=6, view_cap=n, queue_cap=16, event_queue_cap=eq, event_ids_per_ltime=62, inbox_cap=4096, subject_cap=4, gossip_nodes=3,
              fold_interval_ms=5000, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, watch_node=abi.NONE)
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **kw))
    rng = np.random.default_rng(6)
    dead = np.zeros(n, dtype=bool)
    stable = [int(x) for x in rng.choice(n, size=4, replace=False)]
    for w in stable:
        s.watch_events(0, w)
    churnable = np.setdiff1d(np.arange(n), stable)
    s.step_ms(1000)
    s0, fired, got, t0 = s.stats(), [], {w: set() for w in stable}, time.time()

    def drain():
        while True:
            ev = s.poll_events()
            for e in ev:
                if e[2] == abi.EVENT_USER:
                    got[e[6]].add((e[3], e[4]))
            if len(ev) < 4096:
                break
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
                fired.append((sec * 10 + tenth, eid, s.user_event(0, int(o), eid)))
            s.step_ms(100)
            drain()
    old = [(eid, lt) for (t, eid, lt) in fired if t < (secs - 5) * 10]
    cov = lambda: [round(sum(1 for x in old if x in got[w]) / len(old), 3) for w in stable]
    at_end = cov()
    st = s.stats()
    for _ in range(tail * 10):
        s.step_ms(100)
        drain()
    d = {k: st[k] - s0[k] for k in ("event_drops", "queue_drops", "view_drops", "user_events_delivered")}
    print(f"n {n:6d}  event queue {eq:5d}  churn {churn:4.2f}/s : stable observers hold {at_end} of the events at the end of the flood, {cov()} after "
          f"{tail} quiet seconds; event_drops {d['event_drops']}, memberlist queue_drops {d['queue_drops']}, view_drops {d['view_drops']} ({time.time() - t0:.0f} s)", flush=True)
    s.close()


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    for churn in (0.10, 0.0):
        for eq in (16, 32, 64, 4096):
            run(n, eq, churn)
