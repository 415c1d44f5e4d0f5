#!/usr/bin/env python3
"""The unsharded counterpart of bench.py's sharded config-4 leg (run_config4_sharded): same population, same victims, same 1 s + 30 s,
one handle — prints the detection census and the counters the leg reports, to hold the leg's `pairs / suspect_fraction / dead_fraction /
view_drops / inbox_peak` against (a rumour that crosses a shard boundary is judged by the receiving shard, so they must be EQUAL).
  python tools/config4_sharded_check.py --nodes 524288 [--seed 1]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consul_amd import abi, lib as L
from consul_amd.sim import Sim, preset

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=262144)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
hip = L.load()
n, nv = a.nodes, a.nodes // 20
kw = dict(n_nodes=n, seed=a.seed, view_cap=8, mass_rows=nv + 8, queue_cap=16, inbox_cap=8192, subject_cap=4, gossip_nodes=3)
victims = np.random.default_rng(a.seed).choice(n, size=nv, replace=False)
s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
s.step_ms(1000); s.kill(0, victims.tolist()); s.sync()
t0 = time.perf_counter()
s.step_ms(30000); s.sync()
dt = time.perf_counter() - t0
st = s.stats(); pairs, by = s.detection(0)
print(json.dumps({"n_nodes": n, "victims": nv, "wall_s": round(dt, 2), "pairs": pairs, "suspect_fraction": by[1] / pairs, "dead_fraction": (by[2] + by[3]) / pairs,
                  "edges": st["edges"], "msgs_filtered": st["msgs_filtered"], "view_drops": st["view_drops"], "queue_drops": st["queue_drops"],
                  "inbox_overflow": st["inbox_overflow"], "inbox_peak": st["inbox_peak"]}))
