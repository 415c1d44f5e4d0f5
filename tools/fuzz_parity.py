#!/usr/bin/env python3
"""Randomised lock-step parity: random configurations and stimulus schedules, product library (optionally sharded over
several in-process shards) against the unsharded oracle; state digest and counters compared every few ticks.

usage: tools/fuzz_parity.py [--cases N] [--seed S] [--backend hip|emu|oracle]   (oracle = sharded oracle vs unsharded oracle: a CPU
self-check of the harness and of the oracle's own sharding)
"""
import argparse, ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from consul_amd import abi
from consul_amd.sim import Sim, SwimError, preset
from consul_amd.dist import LocalExchange, ShardedSim

KEYS = ["node_rounds_active", "node_rounds_quiescent", "packets_sent", "packets_dropped", "msgs_sent", "msgs_applied", "probes",
        "probe_acks", "probe_indirect_acks", "probe_tcp_acks", "probe_failures", "nacks_missed", "refutes", "suspicion_timeouts",
        "confirmations", "queue_drops", "event_drops", "user_events_delivered", "user_events_deduped", "user_events_stale",
        "piggybacks", "msgs_piggybacked", "push_pulls", "view_drops", "view_evictions", "folds", "fold_freed", "reconnects", "reconnects_reached",
        "edges", "msgs_filtered"]      # (round 4: sharded runs too — what crosses a shard boundary is judged by the receiving shard)


def draw_case(rng):
    which = int(rng.choice([abi.PRESET_LAN, abi.PRESET_LAN, abi.PRESET_WAN, abi.PRESET_LOCAL]))
    shards = int(rng.choice([1, 1, 2, 3, 4]))
    chunk = int(rng.choice([0, 0, 64, 256]))
    per = int(rng.integers(1, 9)) * (chunk or 256) if rng.random() < 0.7 else int(rng.integers(40, 1500))
    n = per * shards
    if chunk and n % (chunk * shards):
        chunk = 0
    if shards > 1 and (n // shards) % (chunk or min(256, 1 << max(0, (n // shards).bit_length() - 1))):
        per = 256 * int(rng.integers(1, 9)); n = per * shards; chunk = 0     # a shard must hold whole stagger chunks
    flags = abi.F_DEFAULT
    for f in (abi.F_BUDDY_SUSPECT, abi.F_NACK, abi.F_FILTER_NOOP, abi.F_PIGGYBACK, abi.F_TCP_FALLBACK):
        if rng.random() < 0.25:
            flags &= ~f
    if rng.random() < 0.5:
        flags |= abi.F_SERF_EVENTS
    kw = dict(n_nodes=n, n_replicas=int(rng.integers(1, 4)), seed=int(rng.integers(1, 1 << 30)), flags=flags,
              gossip_nodes=int(rng.integers(1, 6)), indirect_checks=int(rng.integers(0, 5)), subject_cap=min(n, 1024), view_cap=min(n, 1024),
              queue_cap=int(rng.choice([2, 4, 8, 16])), event_queue_cap=int(rng.choice([2, 4, 8])), inbox_cap=4096,
              loss_q32=int(float(rng.choice([0, 0, 0.05, 0.25])) * 2**32), phase_chunk=chunk,
              suspicion_mult=int(rng.integers(3, 8)), retransmit_mult=int(rng.integers(1, 5)),
              push_pull_interval_ms=int(rng.choice([0, 3000, 30000])), watch_node=int(rng.integers(0, n)))
    # bounded views and folding (drawn last, so that the draws above stay what they were).  A full table that evicts
    # makes the no-op filter visible in the state, and a shard cannot filter what goes to another shard: sharded cases
    # keep tables that never fill.
    tiny = int(rng.choice([0, 0, 4, 16]))
    if tiny and shards == 1:
        kw["view_cap"] = tiny
    kw["fold_interval_ms"] = int(rng.choice([0, 0, 1000, 5000]))
    if rng.random() < 0.5:
        kw["gossip_to_dead_ms"] = int(rng.choice([500, 2000]))      # so that settled views fold / get evicted within a case
    # round 3 (drawn after everything else, so that the cases above stay what they were): serf's reconnect(), a narrow
    # event-buffer slot, and — the product library only — the dense pair store.  The store is a representation: with tables
    # that never fill (tiny == 0) results must not depend on it, and the checker has none.
    kw["reconnect_interval_ms"] = int(rng.choice([0, 0, 1000, 3000]))
    kw["event_ids_per_ltime"] = int(rng.choice([0, 0, 2, 6]))
    rows = int(rng.choice([0, 0, 2, 16, 256]))
    HIP_ONLY.clear()
    if rows and not tiny and kw["suspicion_mult"] <= 4:
        HIP_ONLY["mass_rows"] = min(rows, n)
    return which, shards, kw


UNBOUNDED = False
LENS = False
HIP_ONLY = {}    # per case: configuration of the product library alone (the checker ignores / has no such field)


BRIDGE = {}      # per case: replica -> the node a "real" member is attached as (unsharded cases only)


def bridge_poll_equal(sims):
    """Transport.PacketCh of every attached node: both libraries must have captured the same rumours."""
    for r, att in BRIDGE.items():
        got = [s.transport_poll(r, att) for s in sims]
        if got[0] != got[1]:
            return False, (r, att, got[0][:4], got[1][:4])
    return True, None


def stimulate(rng, sims, n, reps, dead, serf, bridge=False):
    r = int(rng.integers(reps)); op = rng.random()
    ids = [int(x) for x in rng.choice(n, size=int(rng.integers(1, 4)), replace=False)]
    if r in BRIDGE:
        ids = [i for i in ids if i != BRIDGE[r]] or [(BRIDGE[r] + 1) % n]      # the attached node is driven from outside only
    if bridge and op >= 0.90:
        if r not in BRIDGE:
            live = [i for i in range(n) if not dead[r][i]]
            if not live: return
            BRIDGE[r] = int(rng.choice(live))
            [s.transport_poll(r, BRIDGE[r]) for s in sims]                      # the first call attaches
        msgs = [(int(rng.integers(n)), int(rng.integers(1, 4)), int(rng.integers(0, 3)), int(rng.integers(n))) for _ in range(int(rng.integers(1, 4)))]
        dst = int(rng.integers(n))
        if dst != BRIDGE[r]:
            [s.transport_write_to(r, BRIDGE[r], dst, msgs) for s in sims]
        return
    if op < 0.30:
        ids = [i for i in ids if not dead[r][i]]
        for i in ids: dead[r][i] = True
        if ids: [s.kill(r, ids) for s in sims]
    elif op < 0.45:
        ids = [i for i in range(n) if dead[r][i]][:3]
        for i in ids: dead[r][i] = False
        if ids: [s.revive(r, ids) for s in sims]
    elif op < 0.55:
        [s.leave(r, ids) for s in sims]
    elif op < 0.70:
        [s.update(r, ids) for s in sims]
    elif op < 0.78:
        g = (rng.random(n) < 0.02).astype(np.uint8) if rng.random() < 0.6 else np.zeros(n, dtype=np.uint8)
        [s.partition(r, g) for s in sims]
    elif op < 0.84:
        p = float(rng.choice([0.0, 0.1]))
        [s.set_loss(p) for s in sims]
    elif serf:
        live = [i for i in range(n) if not dead[r][i]]
        if live:
            o = int(rng.choice(live)); eid = int(rng.integers(1, 1 << 20))
            [s.user_event(r, o, eid) for s in sims]


def node_fields(ni):
    q = [(e.subject, e.incarnation, e.from_, e.type, e.transmits, e.seq) for e in list(ni.queue)[: ni.queue_len]]
    return dict(inc=ni.incarnation, target=ni.probe_target, deadline=ni.probe_deadline_tick, cursor=ni.probe_cursor, epoch=ni.probe_epoch,
                qlen=ni.queue_len, evqlen=ni.event_queue_len, evclock=ni.event_clock, alive=ni.alive, leaving=ni.leaving,
                awareness=ni.awareness, queue=sorted(q))


def diagnose(k, lib, ora, seed):
    """Replay case k one tick at a time and describe the first divergence."""
    rng = np.random.default_rng([seed, k])
    which, shards, kw = draw_case(rng)
    n, reps = kw["n_nodes"], kw["n_replicas"]
    if UNBOUNDED:   # --unbounded (round 6): memberlist's unbounded queue — on the product library implied by the pair store, so every case gets rows
        kw["flags"] |= abi.F_UNBOUNDED_QUEUE
        kw["gossip_nodes"] = min(kw["gossip_nodes"], 4); kw["suspicion_mult"] = min(kw["suspicion_mult"], 4); kw["queue_cap"] = 32
        kw["view_cap"] = min(n, 1024)                      # (tables that never fill)
        HIP_ONLY["mass_rows"] = min(n, int(rng.choice([16, 256, 4096])))
    if LENS:        # --lens: message lengths drawn too (queue.go orders a tier by length: one, two or three ranks), from a stream of their own — the cases' other draws stay as pinned
        lr = np.random.default_rng([seed, k, 7])
        kw["msg_len"] = [int(x) for x in lr.choice([24, 32, 48, 64, 100, 128], size=3)] + [64]
    a = (ShardedSim([Sim(lib, preset(lib, which, shard_rank=i, n_shards=shards, **kw, **HIP_ONLY)) for i in range(shards)], LocalExchange())
         if shards > 1 else Sim(lib, preset(lib, which, **kw, **HIP_ONLY)))
    b = Sim(ora, preset(ora, which, **kw))
    dead = [[False] * n for _ in range(reps)]
    serf = bool(kw["flags"] & abi.F_SERF_EVENTS)
    tick = 0
    BRIDGE.clear()
    for block in range(int(rng.integers(6, 16))):
        for _ in range(int(rng.integers(0, 3))):
            stimulate(rng, (a, b), n, reps, dead, serf, bridge=shards == 1)
        for _ in range(int(rng.integers(1, 40))):
            a.step(1); b.step(1); tick += 1; a.sync()
            sa, sb = a.stats(), b.stats()
            bad = {key: (sa[key], sb[key]) for key in KEYS if sa[key] != sb[key]}
            if a.digest() == b.digest() and not bad:
                continue
            print(f"  case {k}: first divergence in tick {tick - 1}; stats (hip, oracle): {bad}")
            if shards == 1:
                ea, eb = a.edges(), b.edges()
                sa_, sb_ = set(map(tuple, ea.tolist())), set(map(tuple, eb.tolist()))
                print(f"  edges: hip {len(ea)} oracle {len(eb)}; only hip {sorted(sa_ - sb_)[:6]}; only oracle {sorted(sb_ - sa_)[:6]}")
            shown = 0
            sims_a = a.sims if shards > 1 else [a]
            for r in range(reps):
                for i in range(n):
                    owner = sims_a[i // (n // shards)] if shards > 1 else a
                    fa, fb = node_fields(owner.node_info(r, i)), node_fields(b.node_info(r, i))
                    if fa != fb:
                        print(f"  node ({r},{i}): " + "; ".join(f"{key}: hip {fa[key]} oracle {fb[key]}" for key in fa if fa[key] != fb[key]))
                        shown += 1
                        if shown >= 4: break
                if shown >= 4: break
            if not shown:
                print("  node self-state equal everywhere: the difference is in the views")
                for r in range(reps):
                    for o in range(0, n, max(1, n // 64)):
                        owner = sims_a[o // (n // shards)] if shards > 1 else a
                        ma, mb = owner.members(r, o), b.members(r, o)
                        if not np.array_equal(ma, mb):
                            d = [(x.tolist(), y.tolist()) for x, y in zip(ma, mb) if x.tolist() != y.tolist()][:3]
                            print(f"  observer ({r},{o}) rows differ (hip, oracle): {d}"); shown += 1
                            break
                    if shown: break
            a.close(); b.close()
            return
    print(f"  case {k}: no divergence on replay (non-deterministic?)")
    a.close(); b.close()


def run_case(k, lib, ora, seed, verbose):
    rng = np.random.default_rng([seed, k])
    which, shards, kw = draw_case(rng)
    n, reps = kw["n_nodes"], kw["n_replicas"]
    if UNBOUNDED:   # --unbounded (round 6): memberlist's unbounded queue — on the product library implied by the pair store, so every case gets rows
        kw["flags"] |= abi.F_UNBOUNDED_QUEUE
        kw["gossip_nodes"] = min(kw["gossip_nodes"], 4); kw["suspicion_mult"] = min(kw["suspicion_mult"], 4); kw["queue_cap"] = 32
        kw["view_cap"] = min(n, 1024)                      # (tables that never fill)
        HIP_ONLY["mass_rows"] = min(n, int(rng.choice([16, 256, 4096])))
    if LENS:        # --lens: message lengths drawn too (queue.go orders a tier by length: one, two or three ranks), from a stream of their own — the cases' other draws stay as pinned
        lr = np.random.default_rng([seed, k, 7])
        kw["msg_len"] = [int(x) for x in lr.choice([24, 32, 48, 64, 100, 128], size=3)] + [64]
    try:
        if shards > 1:
            a = ShardedSim([Sim(lib, preset(lib, which, shard_rank=i, n_shards=shards, **kw, **HIP_ONLY)) for i in range(shards)], LocalExchange())
        else:
            a = Sim(lib, preset(lib, which, **kw, **HIP_ONLY))
        b = Sim(ora, preset(ora, which, **kw))
    except SwimError as e:
        if verbose: print(f"case {k}: config refused ({e})")
        return "refused"
    dead = [[False] * n for _ in range(reps)]
    serf = bool(kw["flags"] & abi.F_SERF_EVENTS)
    ticks = 0
    BRIDGE.clear()
    try:
        for block in range(int(rng.integers(6, 16))):
            for _ in range(int(rng.integers(0, 3))):
                stimulate(rng, (a, b), n, reps, dead, serf, bridge=shards == 1)
            step = int(rng.integers(1, 40))
            a.step(step); b.step(step); ticks += step
            a.sync()
            da, db = a.digest(), b.digest()
            sa, sb = a.stats(), b.stats()
            # (`edges` is left out once a node is attached to the transport bridge: a record for an attached node that is not a gossip
            #  rumour — a state exchange, a buddy suspect — is counted when it is sent by the product library and not at all by the
            #  checker, which captures it before it counts; an accounting difference of a debug counter, no state behind it)
            bad = [key for key in KEYS if sa[key] != sb[key] and not (key == "edges" and BRIDGE)]
            same, what = bridge_poll_equal((a, b)) if shards == 1 else (True, None)
            if not same:
                bad.append(f"transport_poll {what}")
            if UNBOUNDED and sa["queue_drops"] and not sb["queue_drops"]:
                # the flag's documented limit on the device: rumours about subjects WITHOUT a row use the queue_cap slots and can be pruned there (counted)
                if verbose: print(f"case {k}: the slots overflowed on the device ({sa['queue_drops']} drops, subjects without a row) after {ticks} ticks: not comparable from here")
                return "slots"
            if da != db or bad:
                print(f"case {k}: MISMATCH after {ticks} ticks: digest {'differs' if da != db else 'ok'}, stats {bad}\n   preset {which} shards {shards} {kw}")
                return "mismatch"
    except SwimError as e:
        # a bounded structure overflowed: legal, but both sides must agree that it did
        print(f"case {k}: {e} after ~{ticks} ticks (preset {which} shards {shards} n {n})")
        return "overflow"
    finally:
        a.close(); b.close()
    if verbose: print(f"case {k}: ok ({ticks} ticks, preset {which}, {shards} shard(s), n {n} x {reps}, rows {HIP_ONLY.get('mass_rows', 0)}, reconnect {kw['reconnect_interval_ms']})")
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--backend", default="hip"); ap.add_argument("-v", action="store_true")
    ap.add_argument("--diagnose", type=int, default=3, help="replay this many mismatching cases tick by tick")
    ap.add_argument("--only", default="", help="comma separated case numbers")
    ap.add_argument("--unbounded", action="store_true", help="every case with SWIM_F_UNBOUNDED_QUEUE (and rows of the pair store on the product library)")
    ap.add_argument("--lens", action="store_true", help="draw the three message lengths too (a stream of their own: the pinned cases keep their other draws)")
    args = ap.parse_args()
    global UNBOUNDED, LENS
    UNBOUNDED = args.unbounded; LENS = args.lens
    ora = abi.bind(C.CDLL(os.environ.get("SWIMSIM_ORACLE_SO") or os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))   # (the ASan build: tools/oracle_asan.sh)
    if args.backend == "hip":
        from consul_amd import lib as L
        lib = L.load()
    elif args.backend == "emu":                     # the kernels' source on the host (tools/emu/build.sh): the same cases where there is no GPU
        lib = abi.bind(C.CDLL(os.environ.get("SWIMSIM_EMU_SO") or os.path.join(ROOT, "tools", "emu", "_build", "libswimsim_emu.so")))
    else:
        lib = ora
    t0 = time.time(); tally = {}; diagnosed = 0
    only = [int(x) for x in args.only.split(",")] if args.only else range(args.cases)
    for k in only:
        res = run_case(k, lib, ora, args.seed, args.v)
        tally[res] = tally.get(res, 0) + 1
        if res == "mismatch" and diagnosed < args.diagnose:
            diagnose(k, lib, ora, args.seed); diagnosed += 1
    print(f"{args.cases} cases in {time.time() - t0:.1f} s: {tally}")
    sys.exit(1 if tally.get("mismatch") else 0)


if __name__ == "__main__":
    main()
