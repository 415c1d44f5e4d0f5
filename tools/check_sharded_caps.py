#!/usr/bin/env python3
"""The bench's N-GPU configuration on ONE device: N shards in this process (records handed over in memory), same
per-node capacities as `bench.py --gpus N`.  Checks that no bounded structure overflows over the whole scenario and that
the sharded run lands on the same state as the unsharded one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset
from consul_amd.dist import LocalExchange, ShardedSim
import bench

n_shards = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
hip = lib.load()
kw = dict(n_nodes=65536, n_replicas=reps, seed=1, subject_cap=4, gossip_nodes=3, queue_cap=4, inbox_cap=96)
victims = bench.victims_for(1, reps, 65536)
sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw)) for i in range(n_shards)], LocalExchange())
ref = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
t0 = time.time()
for s in (sh, ref):
    s.step(50)
    for r, v in enumerate(victims):
        s.kill(r, [v])
    s.step(400)
    s.sync()
st = sh.stats()
print(f"{n_shards} shards x {reps} replicas: digest equal = {sh.digest() == ref.digest()}, edges_remote = {st['edges_remote']}, "
      f"inbox_overflow = {st['inbox_overflow']}, queue_drops = {st['queue_drops']}, piggybacks = {st['piggybacks']}, {time.time() - t0:.1f} s")
assert sh.digest() == ref.digest() and st["inbox_overflow"] == 0
print("OK")
