#!/usr/bin/env python3
"""Tick-by-tick HIP vs oracle on a scripted scenario; prints the first divergence in detail (node self state, members rows,
counters).  usage: tools/diag_parity.py <scenario>   (scenarios: restart, grow)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset

ora = abi.bind(C.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))
hip = lib.load()


def fields(ni):
    q = [(e.subject, e.incarnation, e.from_, e.type, e.transmits, e.seq) for e in list(ni.queue)[: ni.queue_len]]
    return dict(inc=ni.incarnation, target=ni.probe_target, deadline=ni.probe_deadline_tick, cursor=ni.probe_cursor, epoch=ni.probe_epoch,
                qlen=ni.queue_len, alive=ni.alive, leaving=ni.leaving, awareness=ni.awareness, queue=sorted(q))


def compare(a, b, n, tag):
    a.sync()
    if a.digest() == b.digest():
        return True
    print(f"DIVERGENCE {tag}")
    sa, sb = a.stats(), b.stats()
    print("  stats (hip, oracle):", {k: (sa[k], sb[k]) for k in sa if sa[k] != sb[k]})
    shown = 0
    for i in range(n):
        fa, fb = fields(a.node_info(0, i)), fields(b.node_info(0, i))
        if fa != fb:
            print(f"  node {i}: " + "; ".join(f"{k}: hip {fa[k]} oracle {fb[k]}" for k in fa if fa[k] != fb[k])); shown += 1
            if shown >= 6: break
    if not shown:
        for o in range(n):
            ma, mb = a.members(0, o), b.members(0, o)
            if not np.array_equal(ma, mb):
                d = [(x.tolist(), y.tolist()) for x, y in zip(ma, mb) if x.tolist() != y.tolist()][:4]
                print(f"  observer {o} rows differ (hip, oracle): {d}"); shown += 1
                if shown >= 4: break
    return False


def restart():
    n = 512
    kw = dict(n_nodes=n, seed=31, view_cap=64, queue_cap=16, inbox_cap=512, push_pull_interval_ms=0)
    a, b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(ora, preset(ora, abi.PRESET_LAN, **kw))
    rng = np.random.default_rng(8); dead = np.zeros(n, dtype=bool)
    for sec in range(6):
        flip = rng.choice(n, size=n * 3 // 100, replace=False)
        kill, back = flip[~dead[flip]], flip[dead[flip]]; dead[flip] = ~dead[flip]
        via = int(rng.choice(np.flatnonzero(~dead)))
        for s in (a, b):
            if len(kill): s.kill(0, kill.tolist())
            if len(back): s.join(0, back.tolist(), via=via)
        print(f"sec {sec}: kill {kill.tolist()} back {back.tolist()} via {via}")
        if not compare(a, b, n, f"right after the stimulus of second {sec}"): return
        for t in range(10):
            a.step(1); b.step(1)
            if not compare(a, b, n, f"second {sec} tick {t}"): return
    print("no divergence")


def grow():
    n = 32
    kw = dict(n_nodes=n, n_initial=3, seed=2, view_cap=32, inbox_cap=256, fold_interval_ms=2000, watch_node=0)
    a, b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(ora, preset(ora, abi.PRESET_LAN, **kw))
    for s in (a, b): s.step_ms(1000)
    for x in range(3, n):
        for s in (a, b): s.join(0, [x], via=x % 3 if x < 10 else x - 1)
        if not compare(a, b, n, f"right after join of {x}"): return
        for t in range(2):
            a.step(1); b.step(1)
            if not compare(a, b, n, f"joiner {x} tick {t}"): return
    for t in range(100):
        a.step(1); b.step(1)
        if not compare(a, b, n, f"settling tick {t}"): return
    print("no divergence")


if __name__ == "__main__":
    {"restart": restart, "grow": grow}[sys.argv[1]]()
