#!/usr/bin/env python3
"""BASELINE config #2 at the driver's arguments with the 32 clusters spread over G library handles (one stream each):
independent clusters need no common launch, and kernels of different handles overlap on the GPU.
usage: tools/handles_ab.py [G ...]   -> one line per G: node-rounds/s over the timed window"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset

hip = lib.load()
N, REPS, SEED, STEPS, WARM = 65536, 32, 1, int(os.environ.get("STEPS", 20)), int(os.environ.get("WARM", 5))
rng = np.random.default_rng(SEED); victims = [int(rng.integers(N)) for _ in range(REPS)]
for G in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    per = REPS // G
    sims = [Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=N, n_replicas=per, seed=SEED + g * per, subject_cap=2, view_cap=4,
                            queue_cap=4, inbox_cap=24)) for g in range(G)]
    gp = sims[0].derived.gossip_period
    for s in sims: s.step(gp)
    for s in sims: s.sync()
    for s in sims: s.step((WARM - 1) * gp)
    for s in sims: s.sync()
    for g, s in enumerate(sims):
        for r in range(per): s.kill(r, [victims[g * per + r]])
    for s in sims: s.sync()
    t0 = time.perf_counter()
    # interleave the handles' launches so that every stream has work queued early
    left = STEPS * gp
    while left:
        n = min(left, 16)
        for s in sims: s.step(n)
        left -= n
    for s in sims: s.sync()
    dt = time.perf_counter() - t0
    dig = 0
    for s in sims: dig ^= s.digest()
    print(f"G={G} value {REPS * N * STEPS / dt:.3e} node-rounds/s  ms/step {1000 * dt / STEPS:.4f}  digest-xor {dig:#018x}", flush=True)
    for s in sims: s.close()
