"""SWIM_F_UNBOUNDED_QUEUE: memberlist's TransmitLimitedQueue as it is upstream — unbounded (queue.go never drops a broadcast before its
retransmit limit; Consul sizes only serf's event queue: internal/gossip/libserf/serf.go:24-27).  The checker's queues simply grow.  The
HIP library keeps the rumour a node has queued about a subject IN THE PAIR of the dense store (8 more bytes: queued / transmits / type /
sequence number / accuser) and selects GetBroadcasts over the node's column, a wave per node, by (transmits asc, length desc, sequence
desc) like queue.go orders its btree.  Every integer of state must agree all the same: digests (which hash every queued
rumour with its transmit count and sequence number), counters, censuses, detection, member lists, edge lists, node_info's queue."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset
from test_parity_gpu import assert_same

pytestmark = pytest.mark.gpu

UQ = abi.F_DEFAULT | abi.F_UNBOUNDED_QUEUE


def pair(hip, oracle, hip_kw, ora_kw, which=abi.PRESET_LAN, **kw):
    kw.setdefault("flags", UQ)
    return Sim(hip, preset(hip, which, **{**kw, **hip_kw})), Sim(oracle, preset(oracle, which, **{**kw, **ora_kw}))


def queue_of(s, r, i):
    q = s.node_info(r, i)
    return int(q.queue_len), [(int(e.subject), int(e.incarnation), int(e.from_), int(e.type), int(e.transmits), int(e.seq)) for e in list(q.queue)[:min(int(q.queue_len), 32)]]


def test_single_failure_lockstep(hip, oracle):
    """config #2's shape: one victim per cluster, every tick compared (edge lists, digests): the implied queue holds one rumour per node."""
    a, b = pair(hip, oracle, dict(mass_rows=2, view_cap=4), dict(view_cap=4), n_nodes=2048, n_replicas=2, seed=11, subject_cap=4, queue_cap=4)
    victims = [17, 2000]
    for s in (a, b):
        s.step_ms(3000)
        for r, v in enumerate(victims):
            s.kill(r, [v])
    for t in range(100):
        a.step(1); b.step(1)
        assert np.array_equal(a.edges(), b.edges()), f"edge list differs at tick {a.now()[0]}"
        assert a.digest() == b.digest(), f"digest differs at tick {a.now()[0]}"
    for chunk in range(5):
        a.step_ms(5000); b.step_ms(5000)
        assert_same(a, b, list(enumerate(victims)), tag=f"chunk {chunk}")
    assert a.stats()["queue_drops"] == 0 and a.stats()["view_drops"] == 0


@pytest.mark.parametrize("n,nv,seed,secs", [(2048, 100, 44, 44), (1024, 300, 7, 40)])
def test_mass_failure_to_full_detection(hip, oracle, n, nv, seed, secs):
    """config #4's shape: nv nodes stop at once; with every victim's rumour queued at once (nothing pruned) detection is the suspicion
    timeout plus one dissemination.  300 of 1 024: a node's queue outgrows what a wave holds while it scans (the threshold path)."""
    victims = np.random.default_rng(seed).choice(n, size=nv, replace=False)
    a, b = pair(hip, oracle, dict(mass_rows=nv + 8, view_cap=8), dict(view_cap=nv + 64), n_nodes=n, seed=seed, queue_cap=8, inbox_cap=2 * nv + 256, subject_cap=8)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims.tolist())
    watched = [(0, int(v)) for v in victims[:4]]
    for sec in range(0, secs, 2):
        a.step_ms(2000); b.step_ms(2000)
        assert_same(a, b, watched, tag=f"t={sec + 3}s")
        assert a.detection(0) == b.detection(0)
        if sec % 8 == 0:
            o = int(np.setdiff1d(np.arange(n), victims)[sec % 7])
            qa, qb = queue_of(a, 0, o), queue_of(b, 0, o)
            assert qa == qb, (sec, o, qa, qb)
    pairs, by = a.detection(0)
    assert pairs == (n - nv) * nv and by[2] + by[3] == pairs, "every survivor holds every victim dead"
    st = a.stats()
    assert st["queue_drops"] == 0 and st["view_drops"] == 0 and st["inbox_peak"] == b.stats()["inbox_peak"]
    assert np.array_equal(a.members(0, 1), b.members(0, 1))


def test_loss_refutations_and_two_lengths(hip, oracle):
    """6 % packet loss: false suspicions, refutations (alive messages: the long length rank sorts first in a tier), Lifeguard, rumours about
    subjects WITHOUT a row in the queue_cap slots beside the implied ones — one order over both.  (The slots are the flag's limit on the
    device: a rumour about a subject that owns no row can still be pruned there — counted, and at 20 % loss it happens; the checker's
    queue holds them all.  Here the slots are deep enough.)"""
    n = 1024
    victims = list(range(10, 1000, 37))
    a, b = pair(hip, oracle, dict(mass_rows=len(victims) + 4, view_cap=256), dict(view_cap=512), n_nodes=n, seed=21, queue_cap=32, inbox_cap=1024, subject_cap=4,
                loss_q32=int(0.06 * 2**32), flags=UQ & ~abi.F_TCP_FALLBACK)      # (without the TCP fallback ping a lost probe is a false suspicion)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims)
    for sec in range(0, 40, 2):
        a.step_ms(2000); b.step_ms(2000)
        assert_same(a, b, [(0, victims[0])], tag=f"t={sec + 3}s")
    assert a.stats()["refutes"] > 0 and a.stats()["queue_drops"] == 0


def test_leave_update_revive_join(hip, oracle):
    a, b = pair(hip, oracle, dict(mass_rows=16, view_cap=4), dict(view_cap=64), n_nodes=1024, seed=2, subject_cap=16, n_initial=0,
                fold_interval_ms=4000, push_pull_interval_ms=3000, inbox_cap=128, queue_cap=8)
    for s in (a, b):
        s.step_ms(1000)
        s.leave(0, [5, 900])
        s.update(0, [77])
        s.step_ms(3000)
        s.kill(0, [5, 900, 33, 600])
        s.step_ms(8000)
        s.revive(0, [33])          # comes back with its old views AND its old queue, refutes
        s.step_ms(2000)
    assert_same(a, b, [(0, 33), (0, 5), (0, 77)], tag="after revive")
    for s in (a, b):
        s.step_ms(20000)
        s.join(0, [600], via=3)    # a fresh process: its column of the store — views and queued rumours — is cleared
        s.step_ms(30000)
    assert_same(a, b, [(0, 33), (0, 600)], tag="after join")
    assert np.array_equal(a.members(0, 600), b.members(0, 600))
    assert a.stats()["folds"] == b.stats()["folds"] and a.stats()["folds"] > 0


def test_churn_with_folds(hip, oracle):
    """config #5's shape, small: every second 10 % flip alive <-> dead; folds give rows back — but not while a rumour about the subject is
    still queued somewhere on the shard (both libraries) — later kills take them again."""
    import scenarios as sc
    n = 1024
    kw = dict(n_nodes=n, seed=12, queue_cap=8, inbox_cap=4096, subject_cap=4, fold_interval_ms=5000)
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=4), dict(view_cap=n), **kw)
    ra, rb = sc.run_churn(a, n, 30, checkpoints=(10, 20, 30)), sc.run_churn(b, n, 30, checkpoints=(10, 20, 30))
    assert ra == rb
    assert a.stats()["view_drops"] == 0 and a.stats()["queue_drops"] == 0


def test_churn_and_user_events(hip, oracle):
    """config #5 itself, small: churn + serf user events (the event queue rides the same packets after the memberlist queue's share)."""
    import scenarios as sc
    n = 1024
    kw = dict(n_nodes=n, seed=13, queue_cap=8, event_queue_cap=16, event_ids_per_ltime=62, inbox_cap=4096, subject_cap=4, fold_interval_ms=5000,
              flags=UQ | abi.F_SERF_EVENTS, watch_node=0)
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=4), dict(view_cap=n), **kw)
    ra, rb = sc.run_churn_events(a, n, 20, events_per_s=10, checkpoints=(5, 10, 20)), sc.run_churn_events(b, n, 20, events_per_s=10, checkpoints=(5, 10, 20))
    assert ra == rb


def test_partition_heal_and_reconnect(hip, oracle):
    """config #4 as written and its recovery, small: both directions in rows, heal, serf reconnect, push-pull, refutations, folds."""
    import scenarios as sc
    n = 1024
    kw = dict(sc.PARTITION_HEAL_64K, n_nodes=n, inbox_cap=2 * n, queue_cap=8)
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=8), dict(view_cap=n), **kw)
    cps = tuple(range(10, 161, 10))
    ra, rb = sc.run_partition_heal_mass(a, n, checkpoints=cps), sc.run_partition_heal_mass(b, n, checkpoints=cps)
    for sec in cps:
        assert ra[sec] == rb[sec], (sec, ra[sec], rb[sec])
    st = a.stats()
    assert st["view_drops"] == 0 and st["queue_drops"] == 0 and st["refutes"] > 0


def test_rows_run_out(hip, oracle):
    """More subjects than rows: the rest lives in hash tables and in the queue_cap slots (deep enough here that nothing is pruned)."""
    n, nv = 1024, 24
    victims = np.random.default_rng(3).choice(n, size=nv, replace=False)
    a, b = pair(hip, oracle, dict(mass_rows=8), dict(), n_nodes=n, seed=3, view_cap=128, queue_cap=32, inbox_cap=512)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims.tolist())
    for sec in range(10):
        a.step_ms(4000); b.step_ms(4000)
        assert_same(a, b, tag=f"t={4 * sec + 5}s")
    assert a.stats()["queue_drops"] == 0


@pytest.mark.parametrize("n_shards", [2, 4])
def test_sharded_in_process(hip, oracle, n_shards):
    """The pair store — and with it the implied queue — is per shard (its columns are the shard's observers): 2 and 4 shards on one device
    against the UNSHARDED checker; what crosses a shard boundary is judged where it arrives, piggy-back orders for another
    shard's nodes travel with the tick's records, the carried broadcasts with the next tick's."""
    import os
    from consul_amd.dist import LibraryExchange, LocalExchange, ShardedSim
    n, nv = 2048, 120
    victims = np.random.default_rng(9).choice(n, size=nv, replace=False)
    kw = dict(n_nodes=n, seed=9, queue_cap=8, inbox_cap=2048, subject_cap=4, fold_interval_ms=20000, flags=UQ)
    # (the library's own mailbox exchange too, with two shards — as tests/test_mass_gpu.py: four shards in one process would need a hardware queue
    # per stream; not on the emulated kernels, whose workgroups run one after the other)
    exchanges = (LocalExchange, LibraryExchange) if n_shards == 2 and not os.environ.get("SWIMSIM_EMU_SO") else (LocalExchange,)
    for xchg in exchanges:
        a = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, mass_rows=nv + 8, view_cap=8, **kw)) for i in range(n_shards)], xchg())
        b = Sim(oracle, preset(oracle, abi.PRESET_LAN, view_cap=nv + 64, **kw))
        for s in (a, b):
            s.step_ms(1000); s.kill(0, victims.tolist())
        for sec in range(0, 44, 4):
            a.step_ms(4000); b.step_ms(4000); a.sync()
            assert a.digest() == b.digest(), (xchg.__name__, sec)
            assert a.detection(0) == b.detection(0)
        sa, sb = a.stats(), b.stats()
        for k in ("msgs_applied", "msgs_sent", "suspicion_timeouts", "confirmations", "probe_failures", "packets_sent", "push_pulls", "folds", "edges", "msgs_filtered", "piggybacks", "msgs_piggybacked", "queue_drops"):
            assert sa[k] == sb[k], (xchg.__name__, k)
        pairs, by = a.detection(0)
        assert by[2] + by[3] == pairs == (n - nv) * nv and sa["view_drops"] == 0
        a.close(); b.close()


def test_checkpoint_with_an_implied_queue(hip, tmp_path):
    n, nv = 2048, 100
    victims = np.random.default_rng(4).choice(n, size=nv, replace=False).tolist()
    kw = dict(n_nodes=n, seed=4, mass_rows=nv + 8, view_cap=8, queue_cap=8, inbox_cap=1024, flags=UQ)
    a = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)); c = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    a.step_ms(1000); a.kill(0, victims); a.step_ms(12000)
    path = str(tmp_path / "iq.ckpt")
    a.save(path); c.load(path)
    assert a.digest() == c.digest()
    a.step_ms(20000); c.step_ms(20000)
    assert a.digest() == c.digest() and a.stats() == c.stats() and a.detection(0) == c.detection(0)


def test_refused_configurations(hip):
    import ctypes as C
    for bad in (dict(mass_rows=0), dict(mass_rows=8, gossip_nodes=5)):
        cfg = preset(hip, abi.PRESET_LAN, n_nodes=256, flags=UQ, **bad)
        h = C.c_void_p()
        assert hip.swim_create(C.byref(cfg), C.byref(h)) == abi.EINVAL


def test_transport_bridge_with_an_implied_queue(hip, oracle):
    """memberlist.Transport at rumour granularity on a handle whose queue the pair store implies: a node is attached in the MIDDLE of a mass event
    (what it had queued — in the pairs too — is void: peers keep seeing it alive, the simulator stops acting for it), the gossip and the piggy-backed
    broadcasts addressed to it are captured with their senders by k_gossip_iq / k_piggy_iq, what it writes lands in the peers' inboxes."""
    n = 1024
    victims = list(range(3, 1000, 25))
    kw = dict(n_nodes=n, seed=3, subject_cap=8, queue_cap=8, inbox_cap=512, flags=UQ)
    a, b = pair(hip, oracle, dict(mass_rows=len(victims) + 8, view_cap=8), dict(view_cap=128), **kw)
    out = []
    for s in (a, b):
        r = {}
        s.step_ms(1000); s.kill(0, victims); s.step_ms(3000)
        assert s.node_info(0, 7).queue_len > 0               # node 7 has rumours queued (implied by the pairs on the device)
        assert s.transport_poll(0, 7) == []                  # first call attaches node 7
        assert s.node_info(0, 7).queue_len == 0
        s.step_ms(6000)
        r["heard"] = sorted(s.transport_poll(0, 7))
        s.transport_write_to(0, 7, 9, [(200, 1, abi.MSG_SUSPECT, 7)])
        s.step_ms(4000)
        r["after"] = sorted(s.transport_poll(0, 7))
        r["digest"] = s.digest(); r["detection"] = s.detection(0); r["inc200"] = s.node_info(0, 200).incarnation
        out.append(r)
    assert out[0]["heard"] and out[0] == out[1]


def test_randomised_cases_with_the_unbounded_queue(hip, oracle):
    """A fixed slice of `tools/fuzz_parity.py --unbounded` (random configurations, every one with the flag and rows of the pair store on the product
    library).  The seven of seed 606 are the cases that DIFFERED before `k_fold_scan_slots` (folds / fold_freed: the fold rule must also see the
    rumours in the nodes' own slots, profiles/r06_fuzz_unbounded.txt); a case may end early as "slots" — a rumour about a subject without a row was
    pruned from the device's queue_cap slots, counted, the flag's documented limit — but none may differ."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity_uq", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    fz.UNBOUNDED = True
    tally = {}
    for seed, cases in ((606, [25, 50, 51, 61, 88, 98, 191]), (2024, list(range(24)))):
        for k in cases:
            res = fz.run_case(k, hip, oracle, seed, False)
            tally[res] = tally.get(res, 0) + 1
    assert set(tally) <= {"ok", "slots", "refused"} and tally.get("ok", 0) >= 12, tally


@pytest.mark.parametrize("lens,n,nv", [([100, 40, 24, 64], 1024, 200), ([48, 48, 48, 64], 1024, 200), ([40, 100, 100, 64], 1024, 200), ([30, 90, 60, 64], 1024, 200),
                                       ([100, 40, 24, 64], 2048, 700), ([30, 90, 60, 64], 2048, 700)])
def test_message_lengths_and_their_ranks(hip, oracle, lens, n, nv):
    """queue.go orders a tier by message length (longer first): the device keeps a length RANK per type and per rank a keep count and a threshold.  The
    presets have two ranks (alive longer than suspect = dead: the scan's one-select verdict); here three distinct lengths (three ranks, the per-type
    verdict), one length for all (one rank), suspect = dead LONGER than alive, and dead between the two — a mass failure with refutations mixed in
    (3 % loss), every queue compared with the checker's.  With 700 victims a node's queue outgrows the pool a wave holds while it scans: the selection
    runs in mid-scan, with three thresholds standing."""
    seed = 5
    victims = np.random.default_rng(seed).choice(n, size=nv, replace=False)
    a, b = pair(hip, oracle, dict(mass_rows=nv + 64, view_cap=64), dict(view_cap=nv + 128), n_nodes=n, seed=seed, queue_cap=32, inbox_cap=2 * nv + 512, subject_cap=8,
                msg_len=lens, loss_q32=int(0.03 * 2**32), flags=UQ & ~abi.F_TCP_FALLBACK)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims.tolist())
    survivors = np.setdiff1d(np.arange(n), victims)
    for sec in range(0, 30, 2):
        a.step_ms(2000); b.step_ms(2000)
        assert_same(a, b, [(0, int(victims[0])), (0, int(victims[1]))], tag=f"{lens} t={sec + 3}s")
        o = int(survivors[(7 * sec) % len(survivors)])
        qa, qb = queue_of(a, 0, o), queue_of(b, 0, o)
        assert qa == qb, (lens, sec, o, qa, qb)
    st = a.stats()
    assert st["queue_drops"] == 0 and st["view_drops"] == 0 and st["msgs_sent"] == b.stats()["msgs_sent"]
