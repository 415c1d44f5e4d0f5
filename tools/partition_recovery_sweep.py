#!/usr/bin/env python3
"""How long (SIMULATED seconds) the recovery after BASELINE configs[3]'s partition takes, by population size, on the checker: 5 % cut off at t = 1 s
for 60 s, heal, then serf reconnect (30 s) + push-pull + refutations + folds until eight observers (four of either side) hold everybody alive again —
bench.py's config4_partition leg at sizes the checker can run.  python tools/partition_recovery_sweep.py 2048 4096 8192"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consul_amd import abi
from consul_amd.sim import Sim, preset
lib = abi.bind(C.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))
qcap = int(os.environ.get("QUEUE_CAP", "32"))
for n in [int(a) for a in sys.argv[1:]] or [2048]:
    nv = n // 20
    rng = np.random.default_rng(5)
    mask = np.zeros(n, dtype=np.uint8); mask[rng.choice(n, size=nv, replace=False)] = 1
    minority, majority = np.flatnonzero(mask), np.flatnonzero(mask == 0)
    watchers = [int(x) for x in majority[:4]] + [int(x) for x in minority[:4]]
    s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=n, seed=5, view_cap=n, queue_cap=qcap, inbox_cap=2 * n, subject_cap=4, gossip_nodes=3,
                        fold_interval_ms=5000, reconnect_interval_ms=30000))
    t0 = time.time()
    s.step_ms(1000); s.partition(0, mask); s.step_ms(60000); s.partition(0, np.zeros(n, dtype=np.uint8))
    sec, rec = 60, None
    while sec < 60 + 1800:
        s.step_ms(30000); sec += 30
        left = [int(sum(1 for m in s.members(0, w) if int(m["status"]) != abi.MEMBER_ALIVE)) for w in watchers]
        st = s.stats()
        print(json.dumps({"n": n, "queue_cap": qcap, "t_s": sec, "wall_s": round(time.time() - t0, 1), "not_alive_seen_by_watchers": left, "refutes": st["refutes"], "folds": st["folds"], "queue_drops": st["queue_drops"]}), flush=True)
        if not any(left):
            rec = sec; break
    print(f"n {n} queue_cap {qcap}: recovered for the watchers at {rec} s of simulated time", flush=True)
