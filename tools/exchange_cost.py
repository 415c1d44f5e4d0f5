#!/usr/bin/env python3
"""Where a sharded tick's host time goes (world size 1, RCCL): eager split tick vs + count gather vs full exchange."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset
from consul_amd.dist import ShardedSim, TorchExchange
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
hip = lib.load()
kw = dict(n_nodes=65536, n_replicas=32, seed=1, subject_cap=4, queue_cap=4, inbox_cap=24)
T = 2000
def timed(fn, n=T):
    fn(50); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)); print("graph replay (swim_step)          %7.1f us/tick" % timed(lambda n: (s.step(n), s.sync()))); s.close()
s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
def split(n):
    for _ in range(n): s.tick_begin(); s.tick_end()
    s.sync()
print("eager split tick, no exchange      %7.1f us/tick" % timed(split)); s.close()
s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)); ex = TorchExchange(dist.group.WORLD, 0); sh = ShardedSim(s, ex)
print("split tick + TorchExchange         %7.1f us/tick" % timed(lambda n: (sh.step(n), s.sync())))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); sh.step(1000); s.sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
dist.destroy_process_group()
