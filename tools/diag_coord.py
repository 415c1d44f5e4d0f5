#!/usr/bin/env python3
"""Lock-step comparison of the coordinates (SWIM_F_COORDINATES) of the HIP library and the oracle: first differing field."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset
hip = lib.load(); ora = abi.bind(C.CDLL(os.path.join(os.path.dirname(__file__), "..", "oracle", "_build", "libswim_oracle.so")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kw = dict(n_nodes=n, seed=17, flags=abi.F_DEFAULT | abi.F_COORDINATES, rtt_jitter_us=300, subject_cap=4, view_cap=16)
a, b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(ora, preset(ora, abi.PRESET_LAN, **kw))
names = [f"v{i}" for i in range(8)] + ["error", "adjustment", "height"]
def vals(c): return list(c.vec) + [c.error, c.adjustment, c.height]
for t in range(40):
    a.step(1); b.step(1); a.sync()
    sa, sb = a.stats(), b.stats()
    bad = 0
    for i in range(n):
        va, vb = vals(a.coordinate(0, i)), vals(b.coordinate(0, i))
        for f, x, y in zip(names, va, vb):
            if x.hex() != y.hex():
                if bad < 6: print(f"tick {t} node {i} {f}: hip {x!r} ({x.hex()})  oracle {y!r} ({y.hex()})")
                bad += 1
    print(f"tick {t}: updates hip {sa['coord_updates']} oracle {sb['coord_updates']} acks {sa['probe_acks']}/{sb['probe_acks']} mismatching fields {bad} digest {'same' if a.digest() == b.digest() else 'DIFFERENT'}")
    if bad: break
