#!/usr/bin/env python3
"""Generate tests/golden/*.json from the CPU oracle (run by hand; the output is committed).

The reference itself cannot be run here (Go modules absent, no Go toolchain), so these fixtures do
not pin the oracle to the reference — the KATs in tests/test_oracle_kat.py do that as far as it is
possible.  They freeze the oracle's behaviour so that a later edit cannot drift silently.
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from consul_amd import abi  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402

ora = abi.bind(C.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))
KEYS = ["node_rounds_active", "node_rounds_quiescent", "packets_sent", "msgs_sent", "msgs_applied", "probes",
        "probe_acks", "probe_failures", "nacks_missed", "suspicion_timeouts", "confirmations", "edges"]


def config1():
    """BASELINE config #1: 128 nodes, DefaultLANConfig, seed 1, kill node 17 at t=10 s."""
    cfg = dict(n_nodes=128, seed=1)
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **cfg))
    s.step_ms(10000); s.kill(0, [17]); s.step_ms(30000)
    c = s.census(0, 17); st = s.stats()
    return {"config": cfg, "kill_at_ms": 10000, "victim": 17, "run_ms": 30000,
            "detect_ms": [c.first_suspect_ms, c.first_dead_ms, c.all_dead_ms],
            "digest": f"{s.digest():#018x}", "stats": {k: st[k] for k in KEYS}}


def config3_small():
    """BASELINE config #3 shape at 32768 nodes: WAN timers, single update rumour, k in {2,3,5}."""
    out = {}
    for k in (2, 3, 5):
        s = Sim(ora, preset(ora, abi.PRESET_WAN, n_nodes=32768, seed=3, gossip_nodes=k, trace_ticks=64))
        s.update(0, [0]); s.step(60)
        out[str(k)] = {"infected": [int(x) for x in s.trace(0, 0, 0, 60)[:, 4]], "digest": f"{s.digest():#018x}"}
    return {"config": dict(n_nodes=32768, seed=3), "curves": out}


def config3_full():
    """BASELINE config #3 at full size: N = 1 048 576, DefaultWANConfig timers, fan-out k in {2,3,5}, one
    update rumour injected at node 0, tick 0.  Infected count per round until everyone has it."""
    out = {}
    for k in (2, 3, 5):
        s = Sim(ora, preset(ora, abi.PRESET_WAN, n_nodes=1048576, seed=1, gossip_nodes=k, trace_ticks=64, subject_cap=2))
        s.update(0, [0]); s.step(45)
        tr = [int(x) for x in s.trace(0, 0, 0, 45)[:, 4]]
        c = s.census(0, 0)
        out[str(k)] = {"infected": tr, "rounds_to_full": tr.index(1048575) + 1, "all_current_ms": c.all_current_ms,
                       "digest": f"{s.digest():#018x}"}
        s.close()
    return {"config": dict(n_nodes=1048576, seed=1, subject_cap=2), "curves": out}


def _checkpoints(res):
    return {str(sec): {"digest": f"{d:#018x}", "stats": st} for sec, (d, st) in res.items()}


def config4_mass_kill():
    """BASELINE config #4's dynamics with nothing dropped, 65 536 nodes: 3 276 nodes stop at once; to full detection."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios as sc
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **sc.MASS_KILL_64K, **sc.MASS_KILL_64K_ORACLE))
    res = sc.run_mass_kill(s, sc.MASS_KILL_64K["n_nodes"], (5, 15, 30, 60, 120, 300))
    out = {"config": sc.MASS_KILL_64K, "oracle_only": sc.MASS_KILL_64K_ORACLE, "checkpoints": {}}
    for k, v in res.items():
        if k == "done":
            out["done"] = {"second": v[0], "digest": f"{v[1]:#018x}", "stats": v[2], "detection": v[3]}
        else:
            out["checkpoints"][str(k)] = {"digest": f"{v[0]:#018x}", "stats": v[1], "detection": v[2]}
    return out


def config4_mass_kill_unbounded():
    """... with memberlist's unbounded queue (SWIM_F_UNBOUNDED_QUEUE): nothing pruned, to full detection."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios as sc
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **sc.MASS_KILL_64K_UQ, **sc.MASS_KILL_64K_ORACLE))
    res = sc.run_mass_kill(s, sc.MASS_KILL_64K_UQ["n_nodes"], (5, 15, 30, 60, 120))
    out = {"config": sc.MASS_KILL_64K_UQ, "oracle_only": sc.MASS_KILL_64K_ORACLE, "checkpoints": {}}
    for k, v in res.items():
        if k == "done":
            out["done"] = {"second": v[0], "digest": f"{v[1]:#018x}", "stats": v[2], "detection": v[3]}
        else:
            out["checkpoints"][str(k)] = {"digest": f"{v[0]:#018x}", "stats": v[1], "detection": v[2]}
    return out


def config4_mass_kill_16k_unbounded():
    """config #4's shape at 16 384 nodes / 819 stopped with memberlist's unbounded queue, to full detection (46 s; the 32-slot queue: 406 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios as sc
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **sc.MASS_KILL_16K_UQ, **sc.MASS_KILL_16K_ORACLE))
    res = sc.run_mass_kill(s, sc.MASS_KILL_16K_UQ["n_nodes"], (5, 10, 20, 30, 40))
    out = {"config": sc.MASS_KILL_16K_UQ, "oracle_only": sc.MASS_KILL_16K_ORACLE, "checkpoints": {}}
    for k, v in res.items():
        if k == "done":
            out["done"] = {"second": v[0], "digest": f"{v[1]:#018x}", "stats": v[2], "detection": v[3]}
        else:
            out["checkpoints"][str(k)] = {"digest": f"{v[0]:#018x}", "stats": v[1], "detection": v[2]}
    return out


def config4_partition_heal():
    """BASELINE config #4 as written (a partition, both directions) and its recovery phase, 32 768 nodes, nothing dropped (65 536 does
    not fit the build container's memory on the checker: tests/scenarios.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios as sc
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **sc.PARTITION_HEAL_32K, **sc.PARTITION_HEAL_32K_ORACLE))
    res = sc.run_partition_heal_mass(s, sc.PARTITION_HEAL_32K["n_nodes"])
    return {"config": sc.PARTITION_HEAL_32K, "oracle_only": sc.PARTITION_HEAL_32K_ORACLE,
            "checkpoints": {str(k): {"digest": f"{v[0]:#018x}", "stats": v[1], "detection": v[2], "not_alive_seen_by_watchers": v[3]} for k, v in res.items()}}


def config5_churn_events():
    """BASELINE config #5's shape with nothing dropped, 8 192 nodes: 10 %/s churn and 20 serf user events/s for 40 s."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios as sc
    s = Sim(ora, preset(ora, abi.PRESET_LAN, **sc.CHURN_EVENTS_8K, **sc.CHURN_EVENTS_8K_ORACLE))
    res = sc.run_churn_events(s, sc.CHURN_EVENTS_8K["n_nodes"], 40, checkpoints=(10, 20, 40))
    return {"config": sc.CHURN_EVENTS_8K, "oracle_only": sc.CHURN_EVENTS_8K_ORACLE,
            "checkpoints": {str(k): {"digest": f"{v[0]:#018x}", "stats": v[1], "ltimes_fnv": hash_list(v[2]), "watch_node_events": v[3],
                                     "watch_node_events_fnv": hash_list(v[4])} for k, v in res.items()}}


def hash_list(xs):
    h = 0xcbf29ce484222325
    for x in xs:
        h = ((h ^ (int(x) & 0xFFFFFFFF)) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return f"{h:#018x}"


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    only = set(sys.argv[1:])
    for name, fn in (("config1_kill17", config1), ("config3_infection_32k", config3_small),
                     ("config3_infection_1m", config3_full), ("config4_mass_kill_64k", config4_mass_kill), ("config4_mass_kill_64k_unbounded", config4_mass_kill_unbounded), ("config4_mass_kill_16k_unbounded", config4_mass_kill_16k_unbounded), ("config4_partition_heal_32k", config4_partition_heal),
                     ("config5_churn_events_8k", config5_churn_events)):
        if only and name not in only:
            continue
        res = fn()                                  # (computed first: an empty fixture must never sit in the tree while the checker runs for hours)
        with open(os.path.join(ROOT, "tests", "golden", name + ".json"), "w") as f:
            json.dump(res, f, indent=1)
        print("wrote", name)
