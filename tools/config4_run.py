#!/usr/bin/env python3
"""BASELINE config #4 on one GPU (or the checker): N nodes, a uniformly drawn share cut off / stopped at once at t = 1 s, run
until every survivor holds every victim dead (swim_detection_get).  Prints a timeline; used for sizing runs and by hand.
  python tools/config4_run.py --nodes 524288 --mode kill [--oracle] [--seconds 600]"""
import argparse, ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consul_amd import abi
from consul_amd.sim import Sim, preset

ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=65536)
ap.add_argument("--share", type=float, default=0.05)
ap.add_argument("--mode", choices=("kill", "partition"), default="kill")
ap.add_argument("--oracle", action="store_true")
ap.add_argument("--seconds", type=int, default=600)
ap.add_argument("--queue-cap", type=int, default=32)
ap.add_argument("--inbox-cap", type=int, default=0)
ap.add_argument("--view-cap", type=int, default=0)
ap.add_argument("--mass-rows", type=int, default=-1)
ap.add_argument("--push-pull-ms", type=int, default=30000)
ap.add_argument("--every", type=int, default=10)
ap.add_argument("--seed", type=int, default=11)
ap.add_argument("--profile", action="store_true")
ap.add_argument("--unbounded", action="store_true", help="SWIM_F_UNBOUNDED_QUEUE: memberlist's queue as it is upstream (the device: implied by the pair store)")
a = ap.parse_args()
n, nv = a.nodes, int(a.nodes * a.share)
if a.oracle:
    lib = abi.bind(C.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))
    kw = dict(view_cap=a.view_cap or (nv + 64 if a.mode == "kill" else n))
else:
    from consul_amd import lib as L
    lib = L.load()
    rows = a.mass_rows if a.mass_rows >= 0 else (nv + 8 if a.mode == "kill" else n)
    kw = dict(view_cap=a.view_cap or 8, mass_rows=rows)
if a.unbounded:
    kw["flags"] = abi.F_DEFAULT | abi.F_UNBOUNDED_QUEUE
cfg = dict(n_nodes=n, seed=a.seed, queue_cap=a.queue_cap, inbox_cap=a.inbox_cap or min(2 * nv + 256, 8192), subject_cap=8,
           push_pull_interval_ms=a.push_pull_ms, **kw)
t0 = time.time()
s = Sim(lib, preset(lib, abi.PRESET_LAN, **cfg))
print("config", cfg, "create %.1fs" % (time.time() - t0), flush=True)
victims = np.random.default_rng(44).choice(n, size=nv, replace=False)
s.step_ms(1000)
if a.mode == "kill":
    s.kill(0, victims.tolist())
else:
    m = np.zeros(n, dtype=np.uint8); m[victims] = 1; s.partition(0, m)
s.sync()
if a.profile:
    s.profile(True)
t0 = time.time(); done = None
for sec in range(1, a.seconds + 1):
    s.step_ms(1000)
    if sec % a.every == 0 or sec == a.seconds:
        s.sync(); st = s.stats(); pairs, by = s.detection(0)
        print(json.dumps({"t_s": sec + 1, "wall_s": round(time.time() - t0, 2), "pairs": pairs, "alive": by[0], "suspect": by[1], "dead": by[2] + by[3],
                          "view_drops": st["view_drops"], "queue_drops": st["queue_drops"], "inbox_peak": st["inbox_peak"], "push_pulls": st["push_pulls"],
                          "timeouts": st["suspicion_timeouts"], "applied": st["msgs_applied"][:3], "edges": st["edges"]}), flush=True)
        if by[2] + by[3] == pairs and pairs:
            done = sec; break
if a.profile:
    print({k: (v[0], round(v[1], 2)) for k, v in s.profile_read().items()})
print("full detection after", done, "s of simulated time;", "wall %.1f s" % (time.time() - t0))
s.close()
