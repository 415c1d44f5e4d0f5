#!/usr/bin/env python3
"""bench.py — SWIM hot-path throughput on MI355X (BASELINE.json metric: gossip rounds/sec x nodes).

Workload at N=1 = BASELINE.json configs[1]: 65 536-node clusters, memberlist DefaultLANConfig,
fan-out k=3, single failure injection, one cluster replica per seed 1..64 batched on the GPU
(4 194 304 virtual nodes = lanes; `batch_scaling` in the line has the same window with 32 and 128), spread over three
library handles (HIP streams).  Warm-up = the pre-failure phase (default 25 rounds = 5 s of
simulated time), then one uniformly drawn node per replica is killed and the timed region runs K
gossip rounds (default 200 = 40 s simulated: probe failure -> suspicion -> confirmations -> dead ->
dissemination -> quiescence).  A "step" is one gossip round = GossipInterval/quantum ticks of the
whole pipeline (timers, probe, gossip select/emit, delivery, merge) for every node.

At N>1 every replica's node population is block-partitioned over the N ranks (one process per
GPU) and cross-shard gossip records cross once per tick — by default through the library's own
exchange (peer-mapped mailboxes over xGMI, no host round trip: `--exchange library`), with the
split tick + ONE equal-split RCCL all_to_all_single of frames per tick, the counts in the frames' headers, nothing
read back by the host (`--exchange rccl`) timed beside it as `exchange.other`; the replica
count grows with N so per-GPU work is fixed (weak scaling).

One JSON line on stdout (rank 0).  `roofline` is for the kernel that dominates the timed region,
timed with HIP events on the simulator's stream in a second, instrumented pass of the same region
(`traffic` is null: HBM counters need rocprofv3 --pmc passes, see profiles/README.md); `cpu_baseline` is the
plain-C oracle on the host cores, one thread and all cores, on a bounded sample of the same scenario.
Legs that do not depend on --steps/--warmup, so that every line carries them: `detection` (config #2's
deliverable: kill at t = 5 s, run until every replica's survivors all know), `config4` (524 288 nodes on this
GPU = one GPU's share of BASELINE configs[3], 5 % stopped at once, every (survivor, victim) view kept in the dense
pair store, run to full detection), `config4_partition` (the same config as written — a partition, both directions — then heal,
serf's reconnect and folds, at 65 536 nodes), `config5` (churn + user-event flood) and `convergence` (config #3).
Numbers that are quoted from committed profiles rather than measured by this run sit under keys named `quoted_from_profiles`.
"""
from __future__ import annotations

import argparse
import bisect
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from consul_amd import abi  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_FILES = ("r06_pmc_driver.json", "r05_pmc_driver.json", "r04_pmc_driver.json", "r03_pmc_driver.json")     # HBM bytes per launch, newest first (tools/pmc_traffic_pass.sh over the driver's window)


def _pmc():
    for name in PMC_FILES:
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)) as f:
                return name, json.load(f)
        except (OSError, ValueError):
            continue
    return None, {}


def traffic_note() -> str:
    """What the committed --pmc passes say, per tick kernel — built from the file, so it cannot lag behind it."""
    name, d = _pmc()
    if not name:
        return "no committed --pmc passes"
    parts = [f"{k} {v['fetch_bytes_per_launch_raw'] / 1e6 + v['write_bytes_per_launch'] / 1e6:.0f} MB as counted "
             f"({v['fetch_bytes_per_launch_raw'] / 1e6:.0f} fetch + {v['write_bytes_per_launch'] / 1e6:.0f} write) / {v['hbm_bytes_per_launch'] / 1e6:.0f} MB with the guide's x2 on FETCH_SIZE"
             for k, v in d.items() if k in ("k_begin", "k_deliver", "k_resolve")]
    return (f"profiles/{name} (driver window, one handle, per launch; FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, not measured by this run): "
            + "; ".join(parts) + ".  The x2 is calibrated for wide coalesced reads; these kernels scatter 16-64 B, so both figures are quoted.")


def victims_for(seed: int, reps: int, n: int):
    rng = np.random.default_rng(seed)
    return [int(rng.integers(n)) for _ in range(reps)]


class MultiSim:
    """The clusters of config #2 are independent simulations: spread over several library handles (one HIP stream each) their
    tick kernels overlap on the GPU — the tail of one handle's launch runs next to the head of another's.  Replica r of a
    handle seeded s is replica 0 of a handle seeded s + r, so every cluster is the one it would be in a single handle."""

    def __init__(self, sims, first):
        self.sims, self.first, self.derived = sims, first, sims[0].derived       # first[g] = the cluster handle g starts with

    def _where(self, r):
        g = bisect.bisect_right(self.first, r) - 1
        return self.sims[g], r - self.first[g]

    def step(self, n):
        for s in self.sims:                                # asynchronous: every stream has its launches queued before any is awaited
            s.step(n)

    def sync(self):
        for s in self.sims:
            s.sync()

    def kill(self, r, ids):
        sim, k = self._where(r)
        sim.kill(k, ids)

    def census(self, r, x):
        sim, k = self._where(r)
        return sim.census(k, x)

    def close(self):
        for s in self.sims:
            s.close()


def algorithmic_bytes(kernel: str, st: dict) -> float:
    """SURVEY.md §8(d) per-unit bytes, split by the kernel that moves them.

    per active node-round: emit side 16 (header) + 8m (queue slots) + k(4+4m) (edges out);
    delivery side k(4+4m) (edges in); merge k*m*8 (view RMW) + 16+8m (header/slot write-back);
    per quiescent node-round 16; per probe 40.  k, m are the measured packets/node and msgs/packet.

    The 8-byte view access of a rumour is charged to the kernel that makes it: with SWIM_F_FILTER_NOOP the sender's side
    reads the receiver's view and drops the rumour when it would change nothing, so a FILTERED rumour's 8 bytes belong to
    k_begin (gossip) or k_deliver (a broadcast carried by a ping/ack), and only what is delivered reaches k_resolve.
    The sum over the three kernels is what it was without that split.
    """
    active, quiet = st["node_rounds_active"], st["node_rounds_quiescent"]
    pkts, msgs = st["packets_sent"], sum(st["msgs_sent"])
    pb, pbm = st.get("piggybacks", 0), st.get("msgs_piggybacked", 0)   # carriers with a load / broadcasts they carried
    applied = sum(st["msgs_applied"])
    gm = msgs - pbm                                                     # broadcasts sent by gossip() itself
    filtered = min(st.get("msgs_filtered", 0), msgs)
    f_gossip = filtered * gm / max(msgs, 1)                             # (the counter does not say where a rumour was dropped:
    f_carried = filtered - f_gossip                                     #  split in proportion to what each side sent)
    if kernel == "k_begin":       # fused: gossip select/emit + probe (+ timers); a piggy-back order is 8 B of the probe's 40
        return (16.0 * (active + quiet) + 8.0 * (gm / max(pkts, 1)) * active + 4.0 * pkts + 4.0 * gm
                + 40.0 * st["probes"] + 8.0 * f_gossip)
    if kernel == "k_deliver":     # gossip packets + the carried broadcasts (and their carriers' orders)
        return 4.0 * (pkts + pb) + 4.0 * msgs + 8.0 * f_carried
    if kernel == "k_resolve":     # merges + the piggy-back pick (8 B queue slot per carried broadcast, read and written)
        return 8.0 * (msgs - filtered) + 24.0 * applied + 16.0 * pbm
    return 0.0


def diff_stats(a: dict, b: dict) -> dict:
    out = {}
    for k, v in b.items():
        out[k] = [x - y for x, y in zip(v, a[k])] if isinstance(v, list) else v - a[k]
    return out


def run_cpu_baseline(args, ticks_per_round: int) -> dict:
    """The oracle (a scalar C port) on a bounded sample of config #2's scenario, independent of --steps/--warmup:
    25 quiet rounds, the failure, 200 timed rounds per cluster; `per_thread` clusters one after the other on each thread.
    Two figures (SURVEY §8(d) comparator A): one thread, and every host core.  Replica r of seed s is by construction
    replica 0 of seed s+r, so each cluster is an independent one-replica handle (ctypes drops the GIL in swim_step)."""
    from concurrent.futures import ThreadPoolExecutor
    so = os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build_oracle()
    ora = abi.bind(C.CDLL(so))
    warm, rounds, per_thread = 25, 200, 1
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    # One cluster (65 536 nodes, ~150 MB of randomly accessed checker state) per thread.  More threads are not more throughput on this
    # host: 2.45e8 node-rounds/s on 16 threads, 2.05e8 on 32, 1.44e8 on 64, 1.10e8 on 128, 5.6e7 on all 256 (profiles/r04_cpu_baseline_sweep.txt,
    # r04_bench_driver.json) — so the leg sweeps 8 / 16 / 32 threads and reports the BEST as `value` (`cores` = the thread count that
    # reached it); --cpu-threads pins one count.  The clusters are created by the pool's threads too.
    sweep_counts = [max(1, min(ncpu, args.cpu_threads))] if args.cpu_threads else sorted({max(1, min(ncpu, c)) for c in (8, 16, 32)})

    def sample(threads):
        nonlocal per_thread
        reps = threads * per_thread
        victims = victims_for(args.seed, reps, args.nodes)
        sims = [None] * reps

        def prep(r):
            sims[r] = Sim(ora, preset(ora, abi.PRESET_LAN, n_nodes=args.nodes, n_replicas=1, seed=args.seed + r,
                                      subject_cap=args.subject_cap, gossip_nodes=args.fanout))
            sims[r].step(warm * ticks_per_round); sims[r].kill(0, [victims[r]])

        def timed(t):
            for r in range(t, reps, threads):
                sims[r].step(rounds * ticks_per_round)

        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(prep, range(reps)))
            t0 = time.perf_counter()
            list(ex.map(timed, range(threads)))
            dt = time.perf_counter() - t0
        for x in sims:
            x.close()
        return reps * args.nodes * rounds / dt, dt, reps

    per_thread = 4
    v1, dt1, n1 = sample(1)
    per_thread = 1
    sweep = {}
    for c in sweep_counts:
        sweep[c] = sample(c)
    cores = max(sweep, key=lambda c: sweep[c][0])
    vn, dtn, repsn = sweep[cores]
    return {"value": vn, "unit": "node-rounds/s", "cores": cores, "kind": "port",
            "thread_sweep": {str(c): {"value": v[0], "wall_s": round(v[1], 2)} for c, v in sweep.items()},
            "quoted_from_profiles": {"more_threads": "profiles/r04_cpu_baseline_sweep.txt: 64 threads 1.44e8, 128 threads 1.10e8, all 256 threads 5.6e7 node-rounds/s on the same host"},
            "one_thread": {"value": v1, "cores": 1, "wall_s": round(dt1, 2)},
            "sample": f"{repsn} clusters x {args.nodes} nodes x {rounds} rounds after the failure (config #2's scenario, "
                      f"kill after {warm} rounds), {per_thread} per thread: {dtn:.1f} s wall on {cores} of {ncpu} host threads; "
                      f"one thread: {n1} clusters one after the other, {dt1:.1f} s"}


def traffic_from_pmc(kernel: str, virtual_nodes: int):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs of the
    same window, FETCH_SIZE doubled as the guide prescribes for gfx950; tools/pmc_traffic_pass.sh), if they were taken on this
    workload — counters cannot be collected inside this process, so the line quotes the file; None when there is none to quote."""
    k = _pmc()[1].get(kernel)
    if not k or k.get("workload_nodes") != virtual_nodes:
        return None
    return {"hbm_bytes_per_launch": k["hbm_bytes_per_launch"], "as_counted": k["fetch_bytes_per_launch_raw"] + k["write_bytes_per_launch"]}


def roofline_of(prof: dict, st: dict, wall_s=None, virtual_nodes=0) -> dict:
    """`roofline` object for the kernel that took most of the instrumented region (HIP events around every launch)."""
    total_ms = sum(ms for _, ms in prof.values())
    # The headline kernel: among the kernels that take at least a fifth of the region's kernel time, the one FURTHEST below its
    # roofline (round 2 reported whichever kernel led by time, and the fraction swung 10x when k_begin and k_resolve traded
    # places by a microsecond)
    def frac_of(k):
        return algorithmic_bytes(k, st) / max(prof[k][1], 1e-9) / 1e6 / HBM_PEAK_GBS
    heavy = [k for k in prof if prof[k][1] >= 0.2 * total_ms and algorithmic_bytes(k, st) > 0] or [max(prof, key=lambda k: prof[k][1])]
    dom = min(heavy, key=frac_of)
    launches, ms = prof[dom]
    bytes_per_launch = algorithmic_bytes(dom, st) / max(launches, 1)
    avg_s = ms / 1000.0 / max(launches, 1)
    achieved = bytes_per_launch / avg_s / 1e9
    pipe_bytes = sum(algorithmic_bytes(k, st) for k in prof)
    out = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
           "headline_rule": "lowest algorithmic fraction among kernels with >= 20 % of the region's kernel time",
           # HBM bytes per launch come from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE), not from this run:
           "traffic_source": traffic_note(),
           # what the chip delivers for the access pattern these kernels have (tools/scatter_roofline.hip, 8 GB working set,
           # 8 waves per SIMD): dependent 64-byte-line gathers 3.1 TB/s, 16-byte gathers 0.66 TB/s (41 G accesses/s), 16-byte
           # scattered stores 0.36 TB/s, returning 4-byte atomics 20-27 G/s
           "quoted_from_profiles": {"scatter_ceiling": {"line_64B_GBps": 3112.0, "gather_16B_GBps": 661.0, "gather_16B_Gacc_per_s": 41.3, "store_16B_GBps": 357.0,
                                                        "atomic_4B_Gacc_per_s": 19.7, "source": "profiles/r03_scatter_roofline.txt (tools/scatter_roofline.hip; not re-measured by this run)"}},
           "avg_launch_us": 1e6 * avg_s, "launches": launches, "algorithmic_bytes_per_launch": bytes_per_launch,
           "kernel_time_share": {k: round(v[1] / total_ms, 4) for k, v in prof.items()},
           "per_kernel": {k: {"avg_launch_us": 1e3 * v[1] / max(v[0], 1), "launches": v[0],
                              "algorithmic_bytes_per_launch": algorithmic_bytes(k, st) / max(v[0], 1),
                              "frac": (algorithmic_bytes(k, st) / max(v[1], 1e-9) / 1e6) / HBM_PEAK_GBS}
                          for k, v in prof.items() if algorithmic_bytes(k, st) > 0},
           "pipeline": {"algorithmic_GBps_over_kernel_time": pipe_bytes / (total_ms / 1000.0) / 1e9,
                        "kernel_ms_per_round": None}}
    if wall_s:
        out["pipeline"]["algorithmic_GBps_over_wall"] = pipe_bytes / wall_s / 1e9
    t = traffic_from_pmc(dom, virtual_nodes)
    if t:            # (same window, same workload, a separate run: launches there and here process the same ticks)
        out["traffic"] = t["hbm_bytes_per_launch"]
        out["traffic_as_counted"] = t["as_counted"]
        out["traffic_over_algorithmic"] = {"with_x2_on_fetch": t["hbm_bytes_per_launch"] / bytes_per_launch, "as_counted": t["as_counted"] / bytes_per_launch}
    return out


def sim_derived_min_timeout(n: int) -> float:
    """suspicionTimeout(SuspicionMult 4, n, ProbeInterval 1 s) of memberlist's util.go in ms: 4 * max(1, log10 n) * 1 s, the node scale
    truncated to milliseconds as upstream's integer Duration arithmetic does — the timer's minimum, reached after k confirmations."""
    import math
    return 4 * math.floor(1000.0 * max(1.0, math.log10(max(n, 1))))


def run_detection(hip, cfg_kw, victims, G, quantum_ms) -> dict:
    """BASELINE configs[1]'s deliverable, independent of --steps/--warmup: all alive for 5 s, one uniformly drawn node per
    cluster killed, run until the survivors of every cluster all hold it dead (60 s of simulated time at most)."""
    sim = Sim(hip, preset(hip, abi.PRESET_LAN, **cfg_kw))
    t_kill = 5000
    sim.step_ms(t_kill)
    for r, v in enumerate(victims):
        sim.kill(r, [v])
    t0 = time.perf_counter()
    census, ran = [], 0
    for chunk in range(12):
        sim.step_ms(5000); sim.sync(); ran += 5000
        census = [sim.census(r, v) for r, v in enumerate(victims)]
        if all(c.all_dead_ms != abi.NONE for c in census):
            break
    dt = time.perf_counter() - t0
    sim.close()

    def spread(vals):
        v = sorted(int(x) - t_kill for x in vals if x != abi.NONE)
        if not v:
            return {"n": 0}
        return {"n": len(v), "min": v[0], "p25": v[len(v) // 4], "median": v[len(v) // 2], "p75": v[(3 * len(v)) // 4], "max": v[-1]}
    gaps = sorted(int(c.first_dead_ms) - int(c.first_suspect_ms) for c in census if c.first_dead_ms != abi.NONE and c.first_suspect_ms != abi.NONE)
    return {"workload": "kill one uniformly drawn node per cluster at t = 5 s, run to all-know-dead", "clusters": len(victims),
            "simulated_ms_after_failure": ran, "wall_s": round(dt, 3), "rounds_per_sec": ran / quantum_ms / G / dt,
            "ms_after_failure": {"first_suspect": spread(c.first_suspect_ms for c in census),
                                 "first_dead": spread(c.first_dead_ms for c in census),
                                 "all_know_dead": spread(c.all_dead_ms for c in census)},
            # what these times can be held against without the Go reference: the closed form of Lifeguard's timer, and an
            # asynchronous second model of the algorithm at the sizes pure Python reaches
            "compared_with": {
                "first_dead_minus_first_suspect_ms": {"min": gaps[0], "median": gaps[len(gaps) // 2], "max": gaps[-1]} if gaps else None,
                "suspicion_timeout_min_ms": int(sim_derived_min_timeout(cfg_kw["n_nodes"])),
                "quoted_from_profiles": {
                    "async_model": "tests/test_async_reference.py — event-driven, continuous time, per-packet latency (tests/reference_model/): medians async / "
                                   "lock-step at 128 nodes first Dead 10.10 / 10.20 s, everybody knows 10.51 / 10.80 s; at 1 024 nodes 13.97 / 13.80 s, 14.61 / 14.70 s",
                    # north_star asks for +-1 round against memberlist; against the asynchronous model the lock-step determinisation is on time
                    # for first suspicion and first Dead and LATE on the last leg (a verdict is merged at the end of its tick and broadcasts
                    # on pings arrive a tick later): say so next to the numbers
                    "all_know_dead_is_late_by_gossip_rounds": 1.45,
                                                                      "with_a_50_ms_tick": "all three legs within one gossip round of the asynchronous model (swim_config.quantum_ms = 50; tests/test_async_reference.py)"}}}


def run_config4(hip, args, device) -> dict:
    """BASELINE configs[3] on ONE GPU's share, with nothing dropped: N nodes (default 524 288 = 1/8 of the 4 194 304), LAN timers,
    k = 3, 5 % of them (uniformly drawn) unreachable at once at t = 1 s; run until every survivor holds every victim dead
    (swim_detection_get).  The victims are STOPPED: for the survivors — whose detection the config measures — a minority that
    is cut off and one that is down are the same thing (no packet crosses either way), while the cut-off minority's own views
    of the majority would be another 13 G pairs beside the survivors' 13 G.  Every survivor's view of every victim
    lives in the dense pair store (mass_rows): view_drops must be 0.  memberlist's TransmitLimitedQueue is UNBOUNDED, and since round 6 so is
    this leg's (SWIM_F_UNBOUNDED_QUEUE: the rumour a node has queued about a victim lives in the pair, GetBroadcasts selects over the node's
    column — DESIGN 4b): queue_drops must be 0 too.  --bounded-queue runs the leg as rounds 3-5 did (queue_cap slots with Prune())."""
    n, share = args.config4_nodes, 0.05
    nv = int(n * share)
    uq = not getattr(args, "bounded_queue", False)
    kw = dict(n_nodes=n, seed=args.seed, view_cap=8, mass_rows=nv + 8, queue_cap=8 if uq else args.config4_queue_cap,
              inbox_cap=(1 << (nv + 5000).bit_length()) if uq else min(2 * nv + 256, 8192),      # (a state exchange hands a node most of a table: pooled rows)
              subject_cap=4, gossip_nodes=3, device=device, flags=abi.F_DEFAULT | (abi.F_UNBOUNDED_QUEUE if uq else 0))
    victims = np.random.default_rng(args.seed).choice(n, size=nv, replace=False)
    s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    G, q = s.derived.gossip_period, s.derived.quantum_ms
    s.step_ms(1000); s.kill(0, victims.tolist()); s.sync()
    s0 = s.stats()
    if uq:
        s.profile(True)                              # HIP events around every launch of the leg: what the implied queue's kernels cost and move
    t0 = time.perf_counter()
    curve, done, heavy = [], None, None
    budget = args.config4_budget_s
    sec = 0
    while sec < 4000:
        step = 10 if sec < 100 else (20 if uq else 50)
        s.step_ms(1000 * step); sec += step
        s.sync()
        pairs, by = s.detection(0)
        wall = time.perf_counter() - t0
        curve.append({"t_s": sec, "wall_s": round(wall, 2), "dead_fraction": round((by[2] + by[3]) / max(pairs, 1), 6), "suspect_fraction": round(by[1] / max(pairs, 1), 6)})
        if sec == 60:
            heavy = wall
        if pairs and by[2] + by[3] == pairs:
            done = sec
            break
        if wall > budget:
            break
    dt = time.perf_counter() - t0
    st = diff_stats(s0, s.stats())
    ticks = sec * 1000 // q
    out = {"workload": f"BASELINE configs[3], one GPU's share: {n} nodes, {nv} stopped at once at t = 1 s, LAN timers, k = 3; run to full detection "
                       f"(every survivor holds every victim dead); dense pair store, " + ("memberlist's unbounded queue (implied by the pair store)" if uq else f"queue_cap {kw['queue_cap']}"),
           "n_nodes": n, "victims": nv, "pairs": int((n - nv)) * nv,
           "detection_complete": done is not None,
           "rounds_to_full_detection": done * 1000 // q // G if done else None, "simulated_s_to_full_detection": done,
           "simulated_s": sec, "wall_s": round(dt, 2), "rounds_per_sec": sec * 1000 / q / G / dt, "value": n * (sec * 1000 / q / G) / dt, "unit": "node-rounds/s",
           "first_60_s": {"wall_s": round(heavy, 2), "ms_per_round": 1000.0 * heavy / (60000 / q / G)} if heavy else None,
           "view_drops": st["view_drops"], "view_evictions": st["view_evictions"], "queue": "unbounded (SWIM_F_UNBOUNDED_QUEUE)" if uq else f"{kw['queue_cap']} slots, Prune()",
           "queue_drops": st["queue_drops"],
           "inbox_peak": s.stats()["inbox_peak"], "inbox_overflow": st["inbox_overflow"], "push_pulls": st["push_pulls"],
           "edges": st["edges"], "msgs_applied": sum(st["msgs_applied"]), "suspicion_timeouts": st["suspicion_timeouts"],
           # what would cross xGMI if this population were one of 8 shards: 7/8 of the records, 16 bytes each
           "a2a_bytes_per_tick_if_one_of_8_shards": {"mean": 16.0 * 7 / 8 * st["edges"] / max(ticks, 1)},
           "pair_store_GB": round((20.0 if uq else 12.0) * (nv + 8) * n / 1e9, 1),
           "what_the_answer_is_a_property_of": ("memberlist's own queue: unbounded, nothing pruned (queue_drops 0).  What takes the time at this size is the PACKET: every node holds a rumour "
                                                "about every one of the victims, each wants retransmitLimit = 24 transmissions, and a 1 398-byte packet takes 27 of them — 81 per gossip "
                                                "round and node, three hundred rounds for one pass over the queue; detection ends when the last (survivor, victim) pair has either heard "
                                                "the rumour or run out its own suspicion timer.  The same shape on the checker: 31 s at 8 192 nodes / 409 stopped, 46 s at 16 384 / 819 "
                                                "(profiles/r05_queue_cap_sweep.txt); on the device 160 s at 65 536, 560 s at 262 144, 820 s at 524 288 (profiles/r06_*)") if uq else
                                               ("the bounded queue, NOT memberlist (--bounded-queue): memberlist's TransmitLimitedQueue is unbounded; with 32 slots and Prune() the rumours take "
                                                "turns for several times as long (checker, 8 192 nodes: 336 s against 31 s)"),
           "quoted_from_profiles": {"queue_cap_cost_on_the_device": {"measured_at": "262144 nodes / 13107 stopped (profiles/r03_config4_queue_cap.txt)",
                                                                     "simulated_s_to_full_detection": {"8": 1400, "16": 1200, "32": 1100}},
                                    "queue_cap_sweep_on_the_checker": {"measured_at": "8192 nodes / 409 stopped (profiles/r05_queue_cap_sweep.txt)",
                                                                       "simulated_s_to_full_detection": {"8": 396, "16": 361, "32": 336, "64": 266, "128": 306, "256": 51, "4096": 31},
                                                                       "queue_drops_per_applied_message": {"32": 1.68, "256": 0.50, "4096": 0.0},
                                                                       "and_at_16384_nodes_819_stopped": {"32": 406, "128": 406, "512": 56, "2048": 46, "4096": 46}}},
           "curve": curve[:12] + curve[12::4]}
    if uq:
        prof = s.profile_read()
        rows_padded = -(-(nv + 8) // 256) * 256          # the queue words of an observer's column, in 1 KB runs (swim_device.h SW_IQ_RB)
        g = prof.get("k_gossip_iq", (0, 0.0)); pg = prof.get("k_piggy_iq", (0, 0.0))
        scan_bytes = 4.0 * rows_padded * (n - nv) / G    # per launch: every gossip-due survivor reads its whole column once
        out["implied_queue_kernels"] = {
            "kernel_ms_total": {k: round(v[1], 1) for k, v in prof.items() if v[0]}, "launches": int(g[0]),
            "k_gossip_iq": {"avg_launch_ms": g[1] / max(g[0], 1), "column_bytes_scanned_per_launch": scan_bytes,
                            "achieved_GBps": scan_bytes / max(g[1] / max(g[0], 1), 1e-9) / 1e6, "frac_of_8TBps": scan_bytes / max(g[1] / max(g[0], 1), 1e-9) / 1e6 / 8000.0,
                            "bound": "hbm by design (4 bytes per pair and scan); measured: vector-ALU issue — SQ counters (profiles/r06_iq_issue_bound.txt, before the "
                                     "selection): VALU instructions 62 % of the SIMDs' issue slots, all instruction types 113 %; since then the compactions are a selection "
                                     "(not a sort) and a row is judged on its raw word in five instructions; the phase clock has a node at 60 % scan, 22 % compactions, 4 % "
                                     "waiting for the loads"},
            "k_piggy_iq": {"avg_launch_ms": pg[1] / max(pg[0], 1)},
            "quoted_from_profiles": {"hbm_traffic": "profiles/r06_pmc_config4_262k.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; this shape at 262 144 nodes, the first 30 "
                                                    "simulated seconds, per launch; not measured by this run): k_gossip_iq 5.91 GB fetched as counted (11.8 GB with the guide's x2 on FETCH_SIZE) + "
                                                    "1.60 GB written (1.2 GB of it the out-of-line selection's saved registers: scratch) against 6.6 GB of column scan — 0.9x / 1.8x the algorithmic "
                                                    "bytes: nothing is re-read; k_piggy_iq 1.62 GB (3.2 GB) + 0.52 GB; k_resolve<MASS> 3.95 GB (7.9 GB) + 1.55 GB",
                                     "kernel_trace": "profiles/r06_config4_262k_kernel_stats.csv (rocprofv3 --kernel-trace --stats, same command): k_resolve<MASS> 5.68 ms, k_gossip_iq 3.41 ms, "
                                                     "k_piggy_iq 1.14 ms per launch in that phase (before the selection and without k_piggy_iq's device-scope fence: 4.93 and 2.92 ms)"}}
    s.close()
    return out


def run_config4_partition(hip, args, device) -> dict:
    """BASELINE configs[3] AS WRITTEN — a partition, both directions — and its recovery phase, at a size where both directions fit
    the dense pair store (a row for EVERY node: N^2 x 12 bytes; 65 536 nodes = 51.5 GB): 5 % of the nodes are cut off at t = 1 s, the
    cut heals 60 s later (the majority holds ~98 % of the minority dead by then, the minority has started on the majority), then
    serf's reconnect() (one Failed member per node and 30 s, agent/consul/config.go:640-641), push-pull, refutations and folds bring
    everybody back.  tests/test_scale_gpu.py pins this very scenario (other seed) against the checker's fixture at 65 536 nodes.
    Reported: the detection census at the heal, then how many members eight observers (four of either side) still hold not-alive,
    every 30 s until none is left (750 s of simulated time, 117 s of wall time at 65 536 nodes: profiles/r05_config4_partition_65k.json) or the
    push-pull period (30 s x pushPullScale = 360 s at this size) has passed two and a half times."""
    n, cut_s = args.config4p_nodes, 60
    nv = n // 20
    rng = np.random.default_rng(args.seed + 4)
    mask = np.zeros(n, dtype=np.uint8); mask[rng.choice(n, size=nv, replace=False)] = 1
    minority, majority = np.flatnonzero(mask), np.flatnonzero(mask == 0)
    watchers = [int(x) for x in majority[:4]] + [int(x) for x in minority[:4]]
    uq = not getattr(args, "bounded_queue", False)
    kw = dict(n_nodes=n, seed=args.seed + 4, view_cap=8, mass_rows=n, queue_cap=8 if uq else 32, inbox_cap=n // 2, subject_cap=4, gossip_nodes=3,
              fold_interval_ms=5000, reconnect_interval_ms=30000, device=device, flags=abi.F_DEFAULT | (abi.F_UNBOUNDED_QUEUE if uq else 0))
    s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    G, q = s.derived.gossip_period, s.derived.quantum_ms
    s.step_ms(1000); s.partition(0, mask); s.sync()
    s0 = s.stats()
    t0 = time.perf_counter()
    budget, gave_up = args.config4p_budget_s, False
    for _ in range(cut_s // 10):                    # (10 s at a time: the wall-time budget is looked at in between)
        s.step_ms(10000); s.sync()
        if time.perf_counter() - t0 > budget:
            gave_up = True
            break
    t_cut = time.perf_counter() - t0
    if gave_up:
        out = {"n_nodes": n, "cut_off": nv, "gave_up": f"the cut alone passed the leg's wall-time budget ({budget:.0f} s; --config4p-budget-s)", "wall_s": round(t_cut, 2)}
        s.close()
        return out
    pairs, by = s.detection(0)
    at_heal = {"pairs_out_of_reach": pairs, "dead_fraction": (by[2] + by[3]) / max(pairs, 1), "suspect_fraction": by[1] / max(pairs, 1)}
    s.partition(0, np.zeros(n, dtype=np.uint8))
    curve, sec, recovered = [], cut_s, None
    while sec < cut_s + 900 and not gave_up:
        for _ in range(3):
            s.step_ms(10000); s.sync()
            gave_up = gave_up or time.perf_counter() - t0 > budget
        sec += 30
        left = [int(sum(1 for m in s.members(0, w) if int(m["status"]) != abi.MEMBER_ALIVE)) for w in watchers]
        st = diff_stats(s0, s.stats())
        curve.append({"t_s": sec, "wall_s": round(time.perf_counter() - t0, 2), "not_alive_seen_by_watchers": left, "refutes": st["refutes"],
                      "reconnects_reached": st["reconnects_reached"], "rows_freed_by_folds": st["fold_freed"]})
        if not any(left):
            recovered = sec
            break
    dt = time.perf_counter() - t0
    st = diff_stats(s0, s.stats())
    out = {"gave_up_on_wall_time_budget_s": budget if gave_up and recovered is None else None,
           "workload": f"BASELINE configs[3] as written, {n} nodes on one GPU: {nv} cut off (partition, both directions) at t = 1 s for {cut_s} s, then heal + "
                       "serf reconnect (30 s) + push-pull + folds; dense pair store with a row for every node, " + ("memberlist's unbounded queue (implied by the pair store)" if uq else "queue_cap 32"),
           "n_nodes": n, "cut_off": nv, "cut_s": cut_s, "at_heal": at_heal, "watchers": "4 majority + 4 minority observers",
           "recovered_for_the_watchers_at_s": recovered, "simulated_s": sec, "wall_s": round(dt, 2), "wall_s_of_the_cut": round(t_cut, 2),
           "rounds_per_sec": sec * 1000 / q / G / dt, "value": n * (sec * 1000 / q / G) / dt, "unit": "node-rounds/s",
           "refutes": st["refutes"], "reconnects": st["reconnects"], "reconnects_reached": st["reconnects_reached"], "push_pulls": st["push_pulls"],
           "folds": st["folds"], "rows_freed_by_folds": st["fold_freed"], "view_drops": st["view_drops"], "queue_drops": st["queue_drops"],
           "inbox_peak": s.stats()["inbox_peak"], "inbox_overflow": st["inbox_overflow"], "pair_store_GB": round((20.0 if uq else 12.0) * n * n / 1e9, 1), "curve": curve}
    s.close()
    return out


def run_config4_sharded(hip, args, rank, world, device, dist, gather_handles, barrier, allreduce_max) -> dict:
    """BASELINE configs[3] across the ranks: one population block-partitioned over the GPUs, 5 % stopped at once, the 30 s after
    the failure (the phase in which every node learns of every victim: the exchange carries ~(world-1)/world of all records).
    262 144 nodes / 13 107 victims whatever the number of ranks (SWIMSIM_BENCH_C4S_NODES overrides).  Since round 4 a rumour that
    crosses a shard boundary is judged by the no-op filter of the RECEIVING shard (SW_EDGE_JUDGE), so a remote state exchange delivers
    what an unsharded one would and the shards' counters add up to the unsharded run's; before, such an exchange arrived with every
    explicit view of its sender, which is what had bounded the population.  The size stays where it was all the same: the one attempt at
    524 288 nodes on two ranks SHARING one device did not finish within 15 minutes (two processes time-slicing one GPU around the
    exchange's spin-wait; profiles/r04_config4_sharded_524k.txt) and no larger population has run on more than one device.  (The full
    4 194 304 with 209 715 victims needs 1.3 TB of pair store per rank: DESIGN §4a, §10.)
    A failure of this leg (an overflowing bounded structure raises, never passes silently) is reported in the line, not fatal."""
    from consul_amd.dist import LibraryExchange, ShardedSim
    n = int(os.environ.get("SWIMSIM_BENCH_C4S_NODES", 0)) or 262144      # (the override: tests on one device)
    nv = n // 20
    uq = not getattr(args, "bounded_queue", False)
    kw = dict(n_nodes=n, seed=args.seed, view_cap=8, mass_rows=nv + 8, queue_cap=8 if uq else 16, inbox_cap=16384 if uq else 8192, subject_cap=4, gossip_nodes=3,
              flags=abi.F_DEFAULT | (abi.F_UNBOUNDED_QUEUE if uq else 0),
              device=device, shard_rank=rank, n_shards=world)
    victims = np.random.default_rng(args.seed).choice(n, size=nv, replace=False)
    what = (f"{n} nodes block-partitioned over {world} GPUs, {nv} stopped at once, LAN timers, k = 3: the 30 s after the failure; "
            "dense pair store per rank, the library's mailbox exchange")
    s, err, dt, mine = None, None, 0.0, None

    def agree(payload=None):
        """Every rank arrives here whether its part failed or not (one all-gather per phase: it is also the barrier); -> the payloads,
        or None if some rank failed (then every rank gives up together)."""
        got = [None] * world
        dist.all_gather_object(got, (err, payload))
        return None if any(g[0] for g in got) else [g[1] for g in got], next((g[0] for g in got if g[0]), None), sum(1 for g in got if g[0])

    def give_up(first, n_failed):
        try:
            if s is not None:
                s.close()
        except Exception:                                   # noqa: BLE001 (already failing: report the first error)
            pass
        return {"workload": what, "n_nodes": n, "victims": nv, "error": first, "ranks_failed": n_failed}

    try:                                                    # phase 1: allocate (the pair store and the inboxes are tens of GB)
        s = ShardedSim(Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), LibraryExchange(gather_handles))
        G, q = s.sim.derived.gossip_period, s.sim.derived.quantum_ms
    except Exception as e:                                  # noqa: BLE001
        err = f"rank {rank} (create): {e}"[:300]
    ok, first, n_failed = agree()
    if ok is None:
        return give_up(first, n_failed)
    try:                                                    # phase 2: connect the mailboxes (first step), the failure
        s.step_ms(1000); s.kill(0, victims.tolist()); s.sync()
    except Exception as e:                                  # noqa: BLE001 (a rank that stops makes the others' exchange time out: they get here too)
        err = f"rank {rank} (start): {e}"[:300]
    ok, first, n_failed = agree()
    if ok is None:
        return give_up(first, n_failed)
    try:                                                    # phase 3: the timed 30 s (agree() was the barrier); an overflow is raised at the END of the call, on the rank it happened on
        t0 = time.perf_counter()
        s.step_ms(30000); s.sync()
        dt = time.perf_counter() - t0
        mine = (s.sim.stats(), s.sim.detection(0))
    except Exception as e:                                  # noqa: BLE001
        err = f"rank {rank} (run): {e}"[:300]
    ok, first, n_failed = agree((dt, mine))
    if ok is None:
        return give_up(first, n_failed)
    s.close()
    got = [(None,) + g for g in ok]
    dt = max(g[1] for g in got)
    got = [g[2] for g in got]
    st = {k: sum(g[0][k] for g in got) for k in ("edges", "edges_remote", "view_drops", "queue_drops", "inbox_overflow")}
    pairs = sum(g[1][0] for g in got); dead = sum(g[1][1][2] + g[1][1][3] for g in got); susp = sum(g[1][1][1] for g in got)
    ticks = 31000 // q
    return {"workload": what,
            "n_nodes": n, "victims": nv, "wall_s": round(dt, 2), "rounds_per_sec": 30000 / q / G / dt, "value": n * (30000 / q / G) / dt, "unit": "node-rounds/s",
            "pairs": pairs, "suspect_fraction": susp / max(pairs, 1), "dead_fraction": dead / max(pairs, 1),
            "a2a_bytes_per_tick_all_ranks": 16.0 * st["edges_remote"] / ticks, "a2a_bytes_per_tick_per_rank": 16.0 * st["edges_remote"] / ticks / world,
            "view_drops": st["view_drops"], "queue_drops": st["queue_drops"], "inbox_overflow": st["inbox_overflow"],
            "inbox_peak": max(g[0]["inbox_peak"] for g in got), "inbox_cap": kw["inbox_cap"],
            "pair_store_GB_per_rank": round((20.0 if uq else 12.0) * (nv + 8) * (n // world) / 1e9, 1),
            "queue": "memberlist's unbounded queue, implied by each shard's pair store (SWIM_F_UNBOUNDED_QUEUE; sharded since round 6, late: 2 / 4 shards on one device and on "
                     "the emulated kernels == the unsharded checker)" if uq else "16 slots with Prune() (--bounded-queue)",
            "unmeasured_on_two_devices": True}


RCCL_KIND = "rccl: one equal-split all_to_all_single of frames per tick, the counts in the frames' headers; frames sized from the load (one 64-byte-per-peer host look per tick, a tick that does not fit exchanged twice)"


def frame_note(ex) -> dict:
    """What the framed exchange puts on the wire whatever the fill (consul_amd/dist.py TorchExchange)."""
    return {"frames": "fixed" if ex.frame_records else "sized from the load (swim_frame_pack_fill: twice the largest segment of the last 8 ticks, >= 64 records; "
                                                       "a tick that does not fit is packed and exchanged again before it is delivered)",
            "frame_records": ex._F, "frame_bytes_per_tick_per_rank": ex.frame_bytes_per_tick, "ticks": ex.ticks, "ticks_exchanged_twice": ex.retries}


def run_config5(hip, args, device) -> dict:
    """BASELINE configs[4]'s shape on one GPU: N nodes (default 65 536), LAN timers, Lifeguard on (the default flags), 10 % of the
    nodes flip alive <-> dead every second (kill / revive: a node that comes back refutes with a higher incarnation), and a flood
    of serf user events (E per second from uniformly drawn live origins, Lamport-clocked, 512-slot event buffer).  Every node is
    a subject sooner or later: all views live in the dense pair store (mass_rows = N), nothing may be dropped."""
    n, secs, E = args.config5_nodes, args.config5_seconds, args.config5_events
    uq = not getattr(args, "bounded_queue", False)
    kw = dict(n_nodes=n, seed=args.seed + 5, view_cap=8, mass_rows=n, queue_cap=8 if uq else 16, event_queue_cap=16, event_ids_per_ltime=62, inbox_cap=4096, subject_cap=4, gossip_nodes=3,
              fold_interval_ms=5000, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS | (abi.F_UNBOUNDED_QUEUE if uq else 0), watch_node=abi.NONE, device=device)
    s = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    G, q = s.derived.gossip_period, s.derived.quantum_ms
    rng = np.random.default_rng(args.seed + 5)
    dead = np.zeros(n, dtype=bool)
    # four STABLE observers (never killed) whose EventCh is read like a consumer would: which share of the events do they get handed?
    stable = [int(x) for x in rng.choice(n, size=4, replace=False)]
    for w in stable:
        s.watch_events(0, w)
    churnable = np.setdiff1d(np.arange(n), stable)
    got = {w: set() for w in stable}

    def drain():
        while True:
            ev = s.poll_events()
            for e in ev:
                if e[2] == abi.EVENT_USER:
                    got[e[6]].add((e[3], e[4]))
            if len(ev) < 4096:
                break
    s.step_ms(1000); s.sync()
    s0 = s.stats()
    s.profile(True)                                  # (HIP events around the tick kernels: where the leg's time goes)
    t0 = time.perf_counter()
    fired = []
    for sec in range(secs):
        flip = rng.choice(churnable, size=n // 10, replace=False)
        kill, revive = flip[~dead[flip]], flip[dead[flip]]
        dead[flip] = ~dead[flip]
        if len(kill):
            s.kill(0, kill.tolist())
        if len(revive):
            s.revive(0, revive.tolist())
        live = np.flatnonzero(~dead)
        for tenth in range(10):                        # the events of a second arrive spread over it
            for origin in rng.choice(live, size=E // 10, replace=False):
                eid = int(rng.integers(1 << 30))
                fired.append((sec * 10 + tenth, eid, s.user_event(0, int(origin), eid)))
            s.step_ms(100)
            drain()
    s.sync()
    dt = time.perf_counter() - t0
    old = [(eid, lt) for (t, eid, lt) in fired if t < (secs - 5) * 10] or [(eid, lt) for (t, eid, lt) in fired]
    stable_cov = [sum(1 for x in old if x in got[w]) / max(len(old), 1) for w in stable]
    fired = len(fired)
    st = diff_stats(s0, s.stats())
    live = np.flatnonzero(~dead)
    clocks = [s.node_info(0, int(i)).event_clock for i in rng.choice(live, size=min(256, len(live)), replace=False)]
    rounds = secs * 1000 / q / G
    out = {"workload": f"BASELINE configs[4]'s shape on one GPU: {n} nodes, LAN timers, Lifeguard on, 10 %/s churn (kill / revive), {E} serf user events/s; "
                       f"{secs} s simulated; dense pair store for every node; memberlist's queue " + ("unbounded (implied by the pair store); serf's event queue holds 16 (serf: max(2N, 4096))" if uq else "16 slots"),
           "n_nodes": n, "simulated_s": secs, "wall_s": round(dt, 2), "rounds_per_sec": rounds / dt, "value": n * rounds / dt, "unit": "node-rounds/s",
           "events_fired": fired, "event_deliveries": st["user_events_delivered"], "event_deliveries_per_simulated_s": st["user_events_delivered"] / secs,
           "event_deliveries_per_wall_s": st["user_events_delivered"] / dt,
           "mean_coverage_of_an_event": st["user_events_delivered"] / max(fired, 1) / max(len(live), 1),
           # what a consumer sees: the share of the events fired at least 5 s before the end that each of four never-killed nodes was handed on its EventCh
           "coverage_at_stable_observers": {"observers": len(stable), "min": min(stable_cov), "mean": sum(stable_cov) / len(stable_cov), "events_counted": len(old)},
           "quoted_from_profiles": {"event_queue_depth": "profiles/r04_event_queue_depth.txt (checker, 4 096 nodes, this leg's shape): stable observers hold 0.53-0.60 of the "
                                    "events at the end of the flood whether a node's event queue holds 16, 32, 64 or 4 096 entries (serf: max(2N, 4096)), and 1.0 with any "
                                    "depth once the churn is taken away: under 10 %/s churn memberlist's own broadcasts fill the packets first (getBroadcasts before the "
                                    "delegate's), not the queue depth, bound an event's reach; depth shows in the quiet tail only (0.83 / 0.89 / 0.92 after 20 s)"},
           "why_the_coverage_is_what_it_is": ("memberlist's getBroadcasts fills a packet from its OWN queue first and hands the delegate (serf's user events) what is left; with the unbounded "
                                              "queue every node has more membership rumours queued under 10 %/s churn than a packet holds, so an event waits for room that seldom "
                                              "comes: mean coverage 0.07 where the 16-slot queue of rounds 3-5 (which pruned most membership rumours) left room for 0.35 — the faithful "
                                              "number is the low one") if uq else "bounded queue (--bounded-queue): rounds 3-5's figure",
           "dedupe_hits": st["user_events_deduped"], "stale_events": st["user_events_stale"], "event_drops": st["event_drops"],
           "lamport_clock_spread": {"sampled_live_nodes": len(clocks), "min": int(min(clocks)), "max": int(max(clocks))},
           "refutes": st["refutes"], "suspicion_timeouts": st["suspicion_timeouts"], "folds": st["folds"],
           "view_drops": st["view_drops"], "queue_drops": st["queue_drops"], "inbox_peak": s.stats()["inbox_peak"], "inbox_overflow": st["inbox_overflow"],
           "pair_store_GB": round((20.0 if uq else 12.0) * n * n / 1e9, 1),
           "kernel_ms_total": {k: round(v[1], 1) for k, v in s.profile_read().items() if v[0]}}
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed gossip rounds")
    ap.add_argument("--warmup", type=int, default=25, help="untimed gossip rounds before the failure")
    ap.add_argument("--nodes", type=int, default=65536)
    ap.add_argument("--replicas", type=int, default=64, help="cluster replicas per GPU (seeds seed..); rounds 1 and 2 up to BENCH_r02 ran 32: see batch_scaling")
    ap.add_argument("--fanout", type=int, default=3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--handles", type=int, default=3,
                    help="N = 1: spread the clusters over this many library handles, one stream each (their kernels overlap); "
                         "1 = one handle, one launch per kernel and tick for all clusters")
    ap.add_argument("--subject-cap", type=int, default=4)
    ap.add_argument("--mass-rows", type=int, default=0, help="rows of the dense pair store per cluster (swim_config.mass_rows): the victim's views in 12-byte pairs")
    ap.add_argument("--main-only", action="store_true", help="only the timed region (profiling runs): no roofline pass, no extra legs, no CPU baseline")
    ap.add_argument("--no-detection", action="store_true")
    ap.add_argument("--no-config4", action="store_true")
    ap.add_argument("--config4-nodes", type=int, default=524288, help="config4 leg: nodes on this GPU (524288 = one GPU's share of BASELINE configs[3]; ~115 s of wall time: profiles/r04_config4_524k_full.log)")
    ap.add_argument("--config4p-nodes", type=int, default=32768, help="config4_partition leg (the partition as written + heal + recovery): nodes; a row of the "
                                                                    "dense store for every node = N^2 x 20 bytes with the unbounded queue.  Default 32 768 since round 6 (the size of the "
                                                                    "checker's fixture): at 65 536 the leg is 102 s of wall time (profiles/r06_bench_driver_call9.json), and config4 at its full size takes 230 s")
    ap.add_argument("--no-config4-partition", action="store_true")
    ap.add_argument("--config4p-budget-s", type=float, default=150.0, help="config4_partition leg: stop (the curve so far is reported) after this much wall time")
    ap.add_argument("--config4-queue-cap", type=int, default=32, help="with --bounded-queue")
    ap.add_argument("--bounded-queue", action="store_true", help="config4 / config4_partition / config5 legs with queue_cap slots and Prune() (rounds 3-5) instead of memberlist's unbounded queue")
    ap.add_argument("--config4-budget-s", type=float, default=400.0, help="config4 leg: give up (detection_complete false) after this much wall time")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--config5-nodes", type=int, default=65536)
    ap.add_argument("--config5-seconds", type=int, default=20)
    ap.add_argument("--config5-events", type=int, default=20, help="config5 leg: serf user events fired per simulated second")
    ap.add_argument("--cpu-threads", type=int, default=0, help="host threads for the CPU baseline (0 = all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-convergence", action="store_true")
    ap.add_argument("--no-replica-leg", action="store_true", help="N > 1: skip the unsharded 32-clusters-per-GPU comparison")
    ap.add_argument("--no-piggyback", action="store_true", help="ablation: SWIM_F_PIGGYBACK off (not memberlist's behaviour)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="drive the split tick + torch.distributed all-to-all even at world_size 1 (plumbing check)")
    ap.add_argument("--frame-records", type=int, default=0,
                    help="--exchange rccl: records per frame of the equal-split all-to-all (header included).  0 = frames sized from the load "
                         "(twice the largest segment of the last 8 ticks; a tick that does not fit is exchanged again with frames that hold it: lossless, "
                         "one 64-byte-per-peer host look per tick); a number = fixed frames, no host look, the sticky edge-list overflow if a segment does not fit")
    ap.add_argument("--exchange", choices=("auto", "library", "rccl"), default="auto",
                    help="N > 1: `library` = the library's own device-driven exchange (peer-mapped mailboxes over xGMI, no host round trip, "
                         "no collective); `rccl` = split tick + one equal-split RCCL all_to_all_single of frames per tick, issued from Python on the simulator's stream, frames sized from the load (one small host look per tick); `auto` = library, and rccl if the "
                         "mailboxes cannot be set up or a peer's flag does not arrive")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend of the control group (gloo: several ranks on ONE device, tests)")
    args = ap.parse_args()
    if args.main_only:
        args.no_cpu_baseline = args.no_roofline = args.no_convergence = args.no_detection = args.no_config4 = args.no_config5 = True

    # Libraries (RCCL prints a version banner) must not reach stdout: the contract is ONE JSON line.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world
    local_rank %= max(torch.cuda.device_count(), 1)      # (several ranks on one device: --dist-backend gloo)
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.force_exchange
    on_dev = args.dist_backend == "nccl"
    if sharded:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if on_dev:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    from consul_amd import lib
    from consul_amd.dist import LibraryExchange, ShardedSim, TorchExchange
    from consul_amd.sim import SwimError
    hip = lib.load()
    reps = args.replicas * world                       # weak scaling: replicas grow with the ranks
    # one failure per cluster => at most a couple of rumours queued per node: 4 queue slots (LDS per
    # gossip block scales with queue_cap) and 24 inbox slots (in-degree is Poisson(k)) are ample, and
    # an overflow would raise SWIM_EOVERFLOW instead of passing silently
    cfg_kw = dict(n_nodes=args.nodes, n_replicas=reps, seed=args.seed, subject_cap=args.subject_cap,
                  gossip_nodes=args.fanout, queue_cap=4, mass_rows=args.mass_rows,
                  view_cap=4,        # one failure per cluster: an observer never holds more than a couple of explicit views

                  # records from other shards arrive unfiltered (a shard cannot see a remote receiver's view), so a
                  # sharded node's in-degree is the raw Poisson(k) of packets times the rumours in each: more room
                  inbox_cap=24 if world == 1 else 96,
                  device=local_rank, shard_rank=rank, n_shards=world,
                  flags=abi.F_DEFAULT & ~abi.F_PIGGYBACK if args.no_piggyback else abi.F_DEFAULT)
    victims = victims_for(args.seed, reps, args.nodes)

    use_library = world > 1 and args.exchange in ("auto", "library")

    def gather_handles(mine):                          # every rank's mailbox handle, by rank (plain bytes over the control group)
        out = [None] * world
        dist.all_gather_object(out, mine)
        merged = {}
        for d in out:
            merged.update(d)
        return [merged[r] for r in range(world)]

    handles = min(args.handles, reps) if (not sharded and args.handles > 1) else 1

    def fresh(multi=True):
        if multi and handles > 1:
            sizes = [reps // handles + (1 if g < reps % handles else 0) for g in range(handles)]      # 64 on 3: 22 + 21 + 21
            first = [sum(sizes[:g]) for g in range(handles)]
            return MultiSim([Sim(hip, preset(hip, abi.PRESET_LAN, **dict(cfg_kw, n_replicas=sizes[g], seed=args.seed + first[g])))
                             for g in range(handles)], first)
        sim = Sim(hip, preset(hip, abi.PRESET_LAN, **cfg_kw))
        if not sharded:
            return sim
        if use_library:
            return ShardedSim(sim, LibraryExchange(gather_handles))
        if not on_dev and world > 1:
            raise SystemExit("--dist-backend gloo drives the control group only: use --exchange library for the records")
        return ShardedSim(sim, TorchExchange(dist.group.WORLD, local_rank, args.frame_records or None))

    def allreduce_max(x: float) -> float:
        t = torch.tensor([x], device="cuda" if on_dev else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange_used = "library (peer-mapped mailboxes, device driven)" if use_library else RCCL_KIND if sharded else None
    sim = fresh()
    G = (sim.sim if sharded else sim).derived.gossip_period
    if use_library:
        # every rank must take the same decision: a short probe run, then agree
        ok = 1.0
        try:
            sim.step(G); sim.sync()
        except (SwimError, OSError) as e:
            print(f"[bench] library exchange unavailable on rank {rank}: {e}", file=sys.stderr)
            ok = 0.0
        ok = -allreduce_max(-ok)                       # min over ranks
        if ok < 1.0:
            if args.exchange == "library":
                raise SystemExit("the library's device-driven exchange failed and --exchange library leaves no alternative")
            sim.close()
            use_library = False
            exchange_used = RCCL_KIND + " (the library's mailbox exchange could not be used: see stderr)"
            sim = fresh()
            sim.step(G); sim.sync()
    else:
        sim.step(G); sim.sync()                        # first call builds the captured graphs
    tq = time.perf_counter()
    sim.step((args.warmup - 1) * G if args.warmup > 1 else 0); sim.sync()
    quiescent_ms = 1000.0 * (time.perf_counter() - tq) / max(args.warmup - 1, 1)
    for r, v in enumerate(victims):
        sim.kill(r, [v])
    sim.sync()
    barrier()
    t0 = time.perf_counter()
    sim.step(args.steps * G)
    sim.sync()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = allreduce_max(dt)
    base = sim.sim if sharded else sim
    base_quantum = base.derived.quantum_ms
    census = [base.census(r, v) for r, v in enumerate(victims)]
    # BASELINE configs[1] asks for the time-to-detect distribution: over all replicas, in ms after the failure
    t_kill = args.warmup * G * base.derived.quantum_ms

    def spread(vals):
        v = sorted(int(x) - t_kill for x in vals if x != abi.NONE)
        return {"n": len(v), "min": v[0], "median": v[len(v) // 2], "max": v[-1]} if v else {"n": 0}
    detect_after_kill = {"first_suspect": spread(c.first_suspect_ms for c in census),
                         "first_dead": spread(c.first_dead_ms for c in census),
                         "all_know_dead": spread(c.all_dead_ms for c in census)}
    sharded_digest = sharded_stats = None
    if world > 1:
        # correctness of the sharded run, in the line itself: the shards' state digests add up to the digest of the same clusters
        # (same seeds, same victims, same ticks) run unsharded on one GPU — checked below on rank 0
        mine = [None] * world
        dist.all_gather_object(mine, (int(base.digest()), base.stats()))
        sharded_digest = sum(d for d, _ in mine) & 0xFFFFFFFFFFFFFFFF
        sharded_stats = {"edges_remote": sum(st["edges_remote"] for _, st in mine), "edges": sum(st["edges"] for _, st in mine)}
    sim.close()

    value = reps * args.nodes * args.steps / dt
    line = {
        "metric": "gossip rounds/sec x simulated nodes (node-rounds/s)", "value": value, "unit": "node-rounds/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1000.0 * dt / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: 65536-node memberlist clusters, DefaultLANConfig, fanout k=3, "
                               "single failure injection per cluster, replicas batched per GPU",
                   "nodes_per_cluster": args.nodes, "replicas": reps, "fanout": args.fanout,
                   "virtual_nodes": reps * args.nodes, "ticks_per_round": G,
                   "rounds_per_sec": args.steps / dt, "quiescent_ms_per_step": quiescent_ms,
                   "handles": handles,
                   "parallelism": f"population sharded x{world}, one exchange per tick: {exchange_used}" if sharded else
                                  (f"1 GPU, the clusters on {handles} library handles (one HIP stream each; `single_handle` = all in one)"
                                   if handles > 1 else "1 GPU")},
        # what the TIMED window happened to see (it may end before any suspicion runs out): see `detection`
        "timed_window_detection_ms_after_failure": detect_after_kill,
        "parity_vs_reference": "partial (oracle unpinned): the HIP library is bit-identical to oracle/swim_oracle.c — state digests, counters, per-tick edge lists, on every "
                  "scenario of tests/ — and that checker restates memberlist v0.6.0 / serf v0.10.4 from recall: neither module is vendored in the reference "
                  "and no Go toolchain is here; from outside it is pinned only by Philox vectors, the two scaling formulas of agent/config/runtime.go, librtt's "
                  "distance table and libserf's leave-propagation claim.  The +-1 round against real memberlist is a distributional claim that could not be measured",
    }

    if world > 1:
        ticks = (args.warmup + args.steps) * G
        line["exchange"] = {"kind": exchange_used, "a2a_bytes_per_tick_all_ranks": 16.0 * sharded_stats["edges_remote"] / ticks,
                            "a2a_bytes_per_tick_per_rank": 16.0 * sharded_stats["edges_remote"] / ticks / world,
                            "remote_share_of_records": sharded_stats["edges_remote"] / max(sharded_stats["edges"], 1)}
        if rank == 0:
            kw1 = dict(cfg_kw, shard_rank=0, n_shards=1, device=local_rank)
            one = Sim(hip, preset(hip, abi.PRESET_LAN, **kw1))
            one.step(args.warmup * G)
            for r, v in enumerate(victims):
                one.kill(r, [v])
            one.step(args.steps * G); one.sync()
            ud = int(one.digest()); one.close()
            line["parity"] = {"what": "sum of the shards' state digests vs the same clusters run unsharded on one GPU (same seeds, victims, ticks)",
                              "sharded_digest": f"{sharded_digest:#018x}", "unsharded_digest": f"{ud:#018x}", "match": sharded_digest == ud}
        barrier()
        if on_dev and args.exchange == "auto":
            # the other exchange over the same window, so that both are in the line (µs per tick = ms_per_step / ticks_per_round)
            other_lib = not use_library
            try:
                o = Sim(hip, preset(hip, abi.PRESET_LAN, **cfg_kw))
                o = ShardedSim(o, LibraryExchange(gather_handles) if other_lib else TorchExchange(dist.group.WORLD, local_rank, args.frame_records or None))
                o.step(args.warmup * G)
                for r, v in enumerate(victims):
                    o.kill(r, [v])
                o.sync(); barrier()
                to = time.perf_counter()
                o.step(args.steps * G); o.sync(); barrier()
                dto = allreduce_max(time.perf_counter() - to)
                line["exchange"]["other"] = {"kind": "library (peer-mapped mailboxes)" if other_lib else RCCL_KIND,
                                             "value": reps * args.nodes * args.steps / dto, "ms_per_step": 1000.0 * dto / args.steps,
                                             "us_per_tick": 1000.0 * 1000.0 * dto / args.steps / G}
                if not other_lib:
                    line["exchange"]["other"].update(frame_note(o.exchange))
                o.close()
            except (SwimError, OSError, RuntimeError) as e:
                line["exchange"]["other"] = {"error": str(e)[:200]}
        line["exchange"]["us_per_tick"] = 1000.0 * line["ms_per_step"] / G
        if not use_library:
            line["exchange"].update(frame_note(sim.exchange))
    if world > 1 and not args.no_config4:
        if use_library:
            line["config4_sharded"] = run_config4_sharded(hip, args, rank, world, local_rank, dist, gather_handles, barrier, allreduce_max)
        else:
            line["config4_sharded"] = {"skipped": "the leg runs on the library's mailbox exchange, which is not in use in this run"}
    if sharded and not args.no_replica_leg:
        # The clusters of this workload are independent of each other, so the box can also simply run 32 whole clusters
        # per GPU with nothing on the wire (same scenario, captured-graph replay as at N=1): reported next to the
        # sharded figure so that the cost of the per-tick exchange is visible.  Every rank measures, MAX over ranks.
        kw = dict(cfg_kw, n_replicas=args.replicas, seed=args.seed + rank * args.replicas, shard_rank=0, n_shards=1)
        solo = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
        solo.step(args.warmup * G); solo.sync()
        for r, v in enumerate(victims_for(args.seed + rank, args.replicas, args.nodes)):
            solo.kill(r, [v])
        solo.sync(); barrier()
        ts = time.perf_counter()
        solo.step(args.steps * G); solo.sync(); barrier()
        dts = allreduce_max(time.perf_counter() - ts)
        solo.close()
        line["replica_parallel"] = {"value": world * args.replicas * args.nodes * args.steps / dts,
                                    "unit": "node-rounds/s", "ms_per_step": 1000.0 * dts / args.steps,
                                    "parallelism": f"{args.replicas} whole clusters per GPU x {world} GPUs, no data-path collective"}

    if rank == 0 and handles > 1 and not args.main_only:
        # the same region with every cluster in ONE handle (one launch per kernel and tick): what the roofline pass instruments
        one = fresh(multi=False)
        one.step(G); one.sync()
        one.step((args.warmup - 1) * G if args.warmup > 1 else 0)
        for r, v in enumerate(victims):
            one.kill(r, [v])
        one.sync()
        t1 = time.perf_counter()
        one.step(args.steps * G); one.sync()
        dt1 = time.perf_counter() - t1
        one.close()
        line["single_handle"] = {"value": reps * args.nodes * args.steps / dt1, "unit": "node-rounds/s", "ms_per_step": 1000.0 * dt1 / args.steps}
    if rank == 0 and not sharded and not args.main_only:
        # how the same region scales with the number of clusters batched on the GPU (same handles, same window): a tick has a
        # fixed part (five launches, each at least one chain of dependent accesses long) that more clusters amortise
        line["batch_scaling"] = {}
        for other in (32, 128):
            if other == reps:
                continue
            sizes = [other // handles + (1 if g < other % handles else 0) for g in range(handles)]
            first = [sum(sizes[:g]) for g in range(handles)]
            ms = MultiSim([Sim(hip, preset(hip, abi.PRESET_LAN, **dict(cfg_kw, n_replicas=sizes[g], seed=args.seed + first[g])))
                           for g in range(handles)], first)
            vic = victims_for(args.seed, other, args.nodes)
            ms.step(G); ms.sync()
            ms.step((args.warmup - 1) * G if args.warmup > 1 else 0)
            for r, v in enumerate(vic):
                ms.kill(r, [v])
            ms.sync()
            tb = time.perf_counter()
            ms.step(args.steps * G); ms.sync()
            dtb = time.perf_counter() - tb
            ms.close()
            line["batch_scaling"][str(other)] = {"value": other * args.nodes * args.steps / dtb, "ms_per_step": 1000.0 * dtb / args.steps}
        line["batch_scaling"][str(reps)] = {"value": value, "ms_per_step": 1000.0 * dt / args.steps}
    if rank == 0 and not sharded and not args.no_roofline:
        # instrumented pass over the same region: HIP events around every launch on the sim's stream (one handle: the launches
        # are then the ones `single_handle` times; with several handles the same kernels run at 1/handles of the size, overlapped)
        p = fresh(multi=False)
        p.step(args.warmup * G)
        for r, v in enumerate(victims):
            p.kill(r, [v])
        p.sync()
        s0 = p.stats()
        p.profile(True)
        p.step(args.steps * G)
        prof = p.profile_read()
        st = diff_stats(s0, p.stats())
        p.close()
        line["roofline"] = roofline_of(prof, st, dt, reps * args.nodes if (args.steps, args.warmup) == (20, 5) else 0)    # (the PMC passes were taken over the driver's window)          # (pipeline.algorithmic_GBps_over_wall: over the headline run's wall time)
        line["roofline"]["pipeline"]["kernel_ms_per_round"] = sum(ms for _, ms in prof.values()) / args.steps
    if rank == 0 and not sharded and not args.no_detection:
        line["detection"] = run_detection(hip, cfg_kw, victims, G, base_quantum)
    if rank == 0 and not sharded and not args.no_config4:
        line["config4"] = run_config4(hip, args, local_rank)
    if rank == 0 and not sharded and not args.no_config4 and not args.no_config4_partition:
        try:
            line["config4_partition"] = run_config4_partition(hip, args, local_rank)
        except SwimError as e:                      # (an overflowing bounded structure raises, never passes silently: reported, not fatal for the line)
            line["config4_partition"] = {"error": str(e)[:300]}
    if rank == 0 and not sharded and not args.no_config5:
        try:
            line["config5"] = run_config5(hip, args, local_rank)
        except SwimError as e:                      # (reported, not fatal for the line)
            line["config5"] = {"error": str(e)[:300]}
    if rank == 0 and not sharded and not args.no_convergence:
        # second half of the metric: rounds to full convergence at N ~ 1e6 (BASELINE configs[2]:
        # 1 048 576 nodes, DefaultWANConfig timers, one update rumour at node 0, fan-out sweep)
        conv = {}
        for k in (2, 3, 5):
            # the fan-out selects a kernel instantiation; its code object is loaded on first launch (~0.1 s): not timed
            w = Sim(hip, preset(hip, abi.PRESET_WAN, n_nodes=4096, seed=args.seed, gossip_nodes=k, device=local_rank))
            w.step(2); w.sync(); w.close()
            c3 = Sim(hip, preset(hip, abi.PRESET_WAN, n_nodes=1 << 20, seed=args.seed, gossip_nodes=k,
                                 trace_ticks=64, subject_cap=2, view_cap=2, queue_cap=4, inbox_cap=32, device=local_rank))
            c3.update(0, [0])
            c3.step(0); c3.sync()                          # builds the launch graphs, advances nothing
            tc = time.perf_counter()
            c3.step(45); c3.sync()
            dtc = time.perf_counter() - tc
            curve = [int(x) for x in c3.trace(0, 0, 0, 45)[:, 4]]
            full = (1 << 20) - 1
            conv[str(k)] = {"rounds_to_full_convergence": curve.index(full) + 1 if full in curve else None,
                            "rounds_to_half": next(i + 1 for i, v in enumerate(curve) if v >= full // 2),
                            "rounds_per_sec": 45 / dtc}
            c3.close()
        line["convergence"] = {"workload": "BASELINE configs[2]: 1048576 nodes, DefaultWANConfig, single rumour, k in {2,3,5}",
                               "n_nodes": 1 << 20, "by_fanout": conv}
    if rank == 0 and not sharded and not args.no_cpu_baseline:
        line["cpu_baseline"] = run_cpu_baseline(args, G)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
