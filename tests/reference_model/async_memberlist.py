"""An ASYNCHRONOUS reference model of memberlist's failure detector — test infrastructure, pure Python.

Why it exists: the CPU checker (oracle/) and the HIP library share the lock-step determinisation of DESIGN §3 / §8 (integer
ticks, round-trip time 0, piggy-backed broadcasts one tick late, stagger per 256-node chunk, one canonical order of a tick's
arrivals).  A test that compares those two can never see whether that determinisation bends the DYNAMICS.  This model makes none
of those choices: it is event driven on a continuous clock (a heap of timed events), every node has its own randomly staggered
probe and gossip tickers, every packet its own latency, maps and queues are unbounded, arrival order is whatever the latencies
make it.  It follows SURVEY.md Appendix A (the published algorithm of hashicorp/memberlist v0.6.0: state.go probe / probeNode /
suspectNode / aliveNode / deadNode, suspicion.go, awareness.go, queue.go, util.go) for the part config #1 exercises:
a fixed population, one node stops, nobody leaves or joins.  tests/test_async_reference.py compares distributions (time to first
suspicion, to the first Dead verdict, to everybody knowing) over a few hundred seeds with the lock-step simulator.

It is NOT the reference (that is Go code this image cannot build: DESIGN §2); it is a second, differently shaped reading of the
same published algorithm, by the same author.  Parity stays "partial"."""
from __future__ import annotations

import heapq
import math
import random

ALIVE, SUSPECT, DEAD = 0, 1, 2
MSG_LEN = {"alive": 128, "suspect": 48, "dead": 48}           # the simulator's modelled sizes (swim_config.msg_len)
CTL_LEN = {"ping": 86, "indirect": 122, "ack": 108, "nack": 13}


class Cluster:
    def __init__(self, n=128, seed=1, gossip_interval=0.2, gossip_nodes=3, probe_interval=1.0, probe_timeout=0.5, suspicion_mult=4,
                 suspicion_max_mult=6, retransmit_mult=4, indirect_checks=3, awareness_max=8, gossip_to_dead=30.0, udp=1400,
                 latency=(0.0002, 0.002), loss=0.0):
        self.n, self.rng, self.loss = n, random.Random(seed), loss
        self.probe_failures = self.refutes = self.timeouts = 0
        self.gi, self.k, self.pi, self.pt = gossip_interval, gossip_nodes, probe_interval, probe_timeout
        self.ic, self.aw_max, self.g2d, self.budget, self.lat = indirect_checks, awareness_max, gossip_to_dead, udp - 2, latency
        self.limit = retransmit_mult * math.ceil(math.log10(n + 1))
        scale = max(1.0, math.log10(max(1, n)))
        self.s_min = suspicion_mult * int(scale * 1000) * probe_interval / 1000.0
        self.s_max = suspicion_max_mult * self.s_min
        self.s_k = suspicion_mult - 2 if n - 2 >= suspicion_mult - 2 else 0
        self.now, self.heap, self.seq = 0.0, [], 0
        self.up = [True] * n
        self.nodes = [Node(self, i) for i in range(n)]
        for nd in self.nodes:                                    # triggerFunc: a random initial sleep within the interval
            self.at(self.rng.random() * self.pi, nd.probe_tick)
            self.at(self.rng.random() * self.gi, nd.gossip_tick)
        self.log = []                                            # (time, observer, subject, new state)

    def at(self, t, fn, *a):
        self.seq += 1
        heapq.heappush(self.heap, (t, self.seq, fn, a))

    def send(self, src, dst, fn, *a):                            # a UDP packet: arrives after its own latency if dst runs
        if self.up[src] and not (self.loss and self.rng.random() < self.loss):
            self.at(self.now + self.rng.uniform(*self.lat), self._deliver, dst, fn, a)

    def _deliver(self, dst, fn, a):
        if self.up[dst]:
            fn(*a)

    def run(self, until, stop=None):
        while self.heap and self.heap[0][0] <= until:
            t, _, fn, a = heapq.heappop(self.heap)
            self.now = t
            fn(*a)
            if stop and stop():
                return


class Node:
    def __init__(self, c: Cluster, i: int):
        self.c, self.i = c, i
        self.inc, self.awareness = 1, 0
        self.view = {j: [ALIVE, 1, 0.0] for j in range(c.n)}     # state, incarnation, state change
        self.timers, self.tgen = {}, 0                           # subject -> dict(confirmers, k, start, gen)
        self.queue, self.qid = [], 0                             # limitedBroadcast: dict(name, type, inc, frm, transmits, id)
        self.order, self.pidx = [], 0                            # the shuffled probe list and probeIndex
        self.busy_until = 0.0

    # ---- TransmitLimitedQueue ----------------------------------------------------------------------------------------
    def broadcast(self, typ, subject, inc, frm):
        self.queue = [b for b in self.queue if b["name"] != subject]      # a named broadcast invalidates the older one
        self.qid += 1
        self.queue.append(dict(name=subject, type=typ, inc=inc, frm=frm, transmits=0, id=self.qid))

    def get_broadcasts(self, overhead, limit):
        out, used = [], 0
        for b in sorted(self.queue, key=lambda b: (b["transmits"], -MSG_LEN[b["type"]], -b["id"])):
            free = limit - used - overhead
            if MSG_LEN[b["type"]] > free:
                continue
            used += overhead + MSG_LEN[b["type"]]
            out.append(b)
        for b in out:
            b["transmits"] += 1
        self.queue = [b for b in self.queue if b["transmits"] < self.c.limit]
        return [(b["type"], b["name"], b["inc"], b["frm"]) for b in out]

    def handle(self, msgs):
        for typ, subject, inc, frm in msgs:
            getattr(self, typ + "_node")(subject, inc, frm)

    # ---- state.go -----------------------------------------------------------------------------------------------------
    def set_state(self, x, st, inc):
        v = self.view[x]
        if v[0] != st:
            v[2] = self.c.now
            self.c.log.append((self.c.now, self.i, x, st))
        v[0], v[1] = st, inc

    def alive_node(self, x, inc, frm):
        v = self.view[x]
        if x == self.i:
            if inc > self.inc:
                self.refute(inc)
            return
        if inc <= v[1]:
            return
        self.timers.pop(x, None)
        self.broadcast("alive", x, inc, 0)
        self.set_state(x, ALIVE, inc)

    def suspect_node(self, x, inc, frm):
        v = self.view[x]
        if inc < v[1]:
            return
        t = self.timers.get(x)
        if t is not None:                                        # suspicion.Confirm(from)
            if len(t["conf"]) - 1 >= t["k"] or frm in t["conf"]:
                return
            t["conf"].add(frm)
            n = len(t["conf"]) - 1
            frac = math.log(n + 1.0) / math.log(t["k"] + 1.0)
            timeout = max(math.floor(1000.0 * (self.c.s_max - frac * (self.c.s_max - self.c.s_min))) / 1000.0, self.c.s_min)
            self.tgen += 1; t["gen"] = self.tgen                 # timer.Reset: the earlier expiry is void
            self.c.at(max(t["start"] + timeout, self.c.now), self.suspicion_fired, x, t["gen"])
            self.broadcast("suspect", x, inc, frm)
            return
        if v[0] != ALIVE:
            return
        if x == self.i:
            self.refute(inc)
            return
        self.broadcast("suspect", x, inc, frm)
        self.set_state(x, SUSPECT, inc)
        k = self.c.s_k
        self.tgen += 1                                           # (unique per node: an expiry scheduled for an EARLIER suspicion of x is void)
        t = dict(conf={frm}, k=k, start=self.c.now, gen=self.tgen)
        self.timers[x] = t
        self.c.at(self.c.now + (self.c.s_min if k < 1 else self.c.s_max), self.suspicion_fired, x, t["gen"])

    def suspicion_fired(self, x, gen):
        t = self.timers.get(x)
        if t is None or t["gen"] != gen or not self.c.up[self.i]:
            return
        if self.view[x][0] == SUSPECT:
            self.c.timeouts += 1
            self.dead_node(x, self.view[x][1], self.i)

    def dead_node(self, x, inc, frm):
        v = self.view[x]
        if inc < v[1]:
            return
        self.timers.pop(x, None)
        if v[0] == DEAD:
            return
        if x == self.i:
            self.refute(inc)
            return
        self.broadcast("dead", x, inc, frm)
        self.set_state(x, DEAD, inc)

    def refute(self, accused):
        self.c.refutes += 1
        self.inc = max(self.inc + 1, accused + 1)
        self.view[self.i][1] = self.inc                          # state.Incarnation of the own entry: older accusations are stale now
        self.awareness = min(self.awareness + 1, self.c.aw_max - 1)
        self.broadcast("alive", self.i, self.inc, 0)

    # ---- gossip() --------------------------------------------------------------------------------------------------------
    def k_random(self, k, ok):
        out, n = [], self.c.n
        for _ in range(3 * n):
            if len(out) >= k:
                break
            x = self.c.rng.randrange(n)
            if x != self.i and x not in out and ok(x):
                out.append(x)
        return out

    def gossip_tick(self):
        c = self.c
        c.at(c.now + c.gi, self.gossip_tick)
        if not c.up[self.i]:
            return
        for p in self.k_random(c.k, lambda x: self.view[x][0] != DEAD or c.now - self.view[x][2] <= c.g2d):
            msgs = self.get_broadcasts(2, c.budget)
            if not msgs:
                return
            c.send(self.i, p, c.nodes[p].handle, msgs)

    # ---- probe() / probeNode ------------------------------------------------------------------------------------------------
    def probe_tick(self):
        c = self.c
        c.at(c.now + c.pi, self.probe_tick)
        if not c.up[self.i] or c.now < self.busy_until:          # time.Ticker drops ticks while probeNode blocks
            return
        target, checked = None, 0
        while checked < c.n:
            if self.pidx >= len(self.order):                     # resetNodes: forget the long dead, reshuffle
                self.order = [j for j in range(c.n) if not (self.view[j][0] == DEAD and c.now - self.view[j][2] > c.g2d)]
                c.rng.shuffle(self.order)
                self.pidx = 0
                checked += 1
                continue
            j = self.order[self.pidx]; self.pidx += 1
            if j == self.i or self.view[j][0] == DEAD:
                checked += 1
                continue
            target = j
            break
        if target is None:
            return
        interval = c.pi * (self.awareness + 1)                   # awareness.ScaleTimeout(ProbeInterval)
        self.busy_until = c.now + interval
        st = dict(acked=False, nacks=0, expected=0, inc=self.view[target][1])    # probe() hands probeNode a COPY of the nodeState
        piggy = self.get_broadcasts(2, c.budget - CTL_LEN["ping"])
        extra = [("suspect", target, st["inc"], self.i)] if self.view[target][0] != ALIVE else []
        c.send(self.i, target, c.nodes[target].on_ping, self.i, extra + piggy, st, self)
        c.at(c.now + c.pt, self.probe_indirect, target, st)
        c.at(c.now + interval, self.probe_conclude, target, st)

    def on_ping(self, frm, msgs, st, prober):
        self.handle(msgs)
        ack = self.get_broadcasts(2, self.c.budget - CTL_LEN["ack"])
        self.c.send(self.i, frm, prober.on_ack, st, ack)

    def on_ack(self, st, msgs):
        self.handle(msgs)
        if not st["acked"]:
            st["acked"] = True
            self.awareness = max(self.awareness - 1, 0)
            self.busy_until = self.c.now                         # probeNode returns

    def probe_indirect(self, target, st):
        c = self.c
        if st["acked"] or not c.up[self.i]:
            return
        for h in self.k_random(c.ic, lambda x: x != target and self.view[x][0] == ALIVE):
            st["expected"] += 1
            c.send(self.i, h, c.nodes[h].on_indirect, self, target, st)

    def on_indirect(self, prober, target, st):
        c, inner = self.c, dict(acked=False)
        c.send(self.i, target, c.nodes[target].on_relay_ping, self, inner, prober, st)
        c.at(c.now + c.pt, self.relay_timeout, prober, inner, st)

    def on_relay_ping(self, helper, inner, prober, st):
        self.c.send(self.i, helper.i, helper.on_relay_ack, inner, prober, st)

    def on_relay_ack(self, inner, prober, st):
        inner["acked"] = True
        self.c.send(self.i, prober.i, prober.on_ack, st, [])

    def relay_timeout(self, prober, inner, st):
        if not inner["acked"] and self.c.up[self.i]:
            self.c.send(self.i, prober.i, prober.on_nack, st)

    def on_nack(self, st):
        st["nacks"] += 1

    def probe_conclude(self, target, st):
        if st["acked"] or not self.c.up[self.i]:
            return
        delta = (st["expected"] - st["nacks"]) if st["expected"] else 1
        self.c.probe_failures += 1
        self.awareness = min(max(self.awareness + delta, 0), self.c.aw_max - 1)
        self.suspect_node(target, st["inc"], self.i)              # ... and accuses with that copy's incarnation


def config1(seed, n=128, victim=17, kill_at=10.0, horizon=70.0):
    """BASELINE config #1: n nodes, DefaultLANConfig, `victim` stops at `kill_at`; seconds after the stop until the first
    Suspect view, the first Dead view and everybody holding it Dead (None = not within the horizon)."""
    c = Cluster(n=n, seed=seed)
    c.run(kill_at)
    c.up[victim] = False
    dead = set()

    def all_know():
        return len(dead) == n - 1

    first_s = first_d = allk = None
    mark = len(c.log)
    while c.heap and c.heap[0][0] <= kill_at + horizon:
        c.run(c.heap[0][0])
        for t, o, x, st in c.log[mark:]:
            if x != victim:
                continue
            if st == SUSPECT and first_s is None:
                first_s = t - kill_at
            if st == DEAD:
                dead.add(o)
                if first_d is None:
                    first_d = t - kill_at
        mark = len(c.log)
        if all_know():
            allk = c.now - kill_at
            break
    return first_s, first_d, allk


def lossy(seed, n=128, loss=0.2, seconds=60.0):
    """Nobody stops, every packet is lost with probability `loss`, no TCP fallback ping: probes fail now and then, the accused
    refute.  -> (failed probes, refutations, suspicion timers that ran out, mean awareness score at the end)."""
    c = Cluster(n=n, seed=seed, loss=loss)
    c.run(seconds)
    return c.probe_failures, c.refutes, c.timeouts, sum(nd.awareness for nd in c.nodes) / n


def update(seed, n=128, who=5, at=5.0, horizon=20.0):
    """memberlist.UpdateNode on `who` at `at`: it bumps its incarnation and queues alive{}; seconds until the last member holds
    the new incarnation (the dissemination time of ONE rumour through gossip and piggy-backing)."""
    c = Cluster(n=n, seed=seed)
    c.run(at)
    nd = c.nodes[who]
    nd.inc += 1
    nd.view[who][1] = nd.inc
    nd.broadcast("alive", who, nd.inc, 0)
    done = [None]

    def everybody():
        if all(m.view[who][1] == nd.inc for m in c.nodes):
            done[0] = c.now - at
            return True
        return False
    c.run(at + horizon, stop=everybody)
    return done[0]
