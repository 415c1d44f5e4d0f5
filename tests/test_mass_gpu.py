"""The dense pair store (swim_config.mass_rows) is a REPRESENTATION: a subject named in a stimulus call keeps
every observer's view of it in 12-byte pairs of [row][observer] planes instead of 64-byte hash-table entries counted
against view_cap.  No result may depend on it.  So here the HIP library runs with rows and a view_cap far too small for
the scenario, the checker — which has no such store — with a view_cap that holds everything, and every integer of state
must agree: digests, counters, censuses, member lists, edge lists."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset
from test_parity_gpu import STAT_KEYS, assert_same

pytestmark = pytest.mark.gpu


def pair(hip, oracle, hip_kw, ora_kw, which=abi.PRESET_LAN, **kw):
    return Sim(hip, preset(hip, which, **{**kw, **hip_kw})), Sim(oracle, preset(oracle, which, **{**kw, **ora_kw}))


def test_single_failure_in_a_row_lockstep(hip, oracle):
    """config #2's shape: the one victim per cluster owns row 0; every tick compared (edge lists, digests)."""
    a, b = pair(hip, oracle, dict(mass_rows=2, view_cap=4), dict(view_cap=4), n_nodes=4096, n_replicas=3, seed=11, subject_cap=4)
    victims = [17, 4000, 2048]
    for s in (a, b):
        s.step_ms(3000)
        for r, v in enumerate(victims):
            s.kill(r, [v])
    for t in range(120):
        a.step(1); b.step(1)
        assert np.array_equal(a.edges(), b.edges()), f"edge list differs at tick {a.now()[0]}"
        assert a.digest() == b.digest(), f"digest differs at tick {a.now()[0]}"
    for chunk in range(6):
        a.step_ms(5000); b.step_ms(5000)
        assert_same(a, b, list(enumerate(victims)), tag=f"chunk {chunk}")
    assert a.census(0, 17).all_dead_ms != abi.NONE
    assert np.array_equal(a.members(0, 3), b.members(0, 3))
    assert a.stats()["view_drops"] == 0


@pytest.mark.parametrize("n,share,seed", [(4096, 0.05, 44), (16384, 0.02, 7)])
def test_mass_failure_matches_the_checker(hip, oracle, n, share, seed):
    """5 % of the nodes stop at once: every survivor ends up holding every victim dead.  The HIP library keeps those views in
    rows (its hash tables hold 8 entries); the checker in hash tables that hold them all."""
    nv = int(n * share)
    victims = np.random.default_rng(seed).choice(n, size=nv, replace=False)
    a, b = pair(hip, oracle, dict(mass_rows=nv + 8, view_cap=8), dict(view_cap=nv + 64),
                n_nodes=n, seed=seed, queue_cap=32, inbox_cap=2 * nv + 256, subject_cap=8)
    for s in (a, b):
        s.step_ms(1000)
        s.kill(0, victims.tolist())
    watched = [(0, int(v)) for v in victims[:8]]
    for sec in range(0, 70, 2):
        a.step_ms(2000); b.step_ms(2000)
        assert_same(a, b, watched, tag=f"t={sec + 3}s")
    st = a.stats()
    assert st["view_drops"] == 0 and st["suspicion_timeouts"] > nv and st["msgs_applied"][2] > 0
    assert st["inbox_peak"] == b.stats()["inbox_peak"]
    for _, v in watched[:3]:
        ca = a.census(0, v)
        assert ca.by_state[2] + ca.by_state[3] >= 0.99 * ca.n_observers
    assert np.array_equal(a.members(0, 1), b.members(0, 1))


def test_partition_heal_with_rows_and_folds(hip, oracle):
    """SURVEY §8(f) rank 3 with the minority in rows: 5 % cut off for 45 s, both sides declare each other dead, the cut heals,
    push-pull brings everybody back, folds free the rows."""
    import scenarios as sc
    n = 2048
    a, b = pair(hip, oracle, dict(mass_rows=160), dict(), **sc.HEAL_2K)
    ra, rb = sc.run_partition_heal(a, n), sc.run_partition_heal(b, n)
    assert ra == rb
    assert a.stats()["folds"] > 0 and a.stats()["view_drops"] == 0


def test_leave_update_revive_join_in_rows(hip, oracle):
    a, b = pair(hip, oracle, dict(mass_rows=16, view_cap=4), dict(view_cap=64), n_nodes=1024, seed=2, subject_cap=16, n_initial=0,
                fold_interval_ms=4000, push_pull_interval_ms=3000, inbox_cap=128)
    for s in (a, b):
        s.step_ms(1000)
        s.leave(0, [5, 900])
        s.update(0, [77])
        s.step_ms(3000)
        s.kill(0, [5, 900, 33, 600])
        s.step_ms(8000)
        s.revive(0, [33])          # comes back with its old views, refutes
        s.step_ms(2000)
    assert_same(a, b, [(0, 33), (0, 5), (0, 77)], tag="after revive")
    for s in (a, b):
        s.step_ms(20000)
        s.join(0, [600], via=3)    # a fresh process: clears its column of the store
        s.step_ms(30000)
    assert_same(a, b, [(0, 33), (0, 600)], tag="after join")
    assert np.array_equal(a.members(0, 600), b.members(0, 600))
    assert np.array_equal(a.members(0, 8), b.members(0, 8))


def test_churn_recycles_rows_through_folds(hip, oracle):
    """config #5's shape, small: every second 10 % flip alive <-> dead; every node becomes a subject sooner or later, folds give
    rows back, later kills take them again."""
    import scenarios as sc
    n = 2048
    kw = dict(n_nodes=n, seed=12, queue_cap=16, inbox_cap=4096, subject_cap=4, fold_interval_ms=5000)
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=4), dict(view_cap=n), **kw)
    ra, rb = sc.run_churn(a, n, 40, checkpoints=(10, 20, 30, 40)), sc.run_churn(b, n, 40, checkpoints=(10, 20, 30, 40))
    assert ra == rb
    assert a.stats()["view_drops"] == 0


def test_rows_run_out_gracefully(hip, oracle):
    """More subjects than rows: the rest lives in the hash tables, results unchanged (view_cap holds them)."""
    n, nv = 2048, 64
    victims = np.random.default_rng(3).choice(n, size=nv, replace=False)
    a, b = pair(hip, oracle, dict(mass_rows=16), dict(), n_nodes=n, seed=3, view_cap=128, queue_cap=16, inbox_cap=512)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims.tolist())
    for sec in range(10):
        a.step_ms(4000); b.step_ms(4000)
        assert_same(a, b, tag=f"t={4 * sec + 5}s")


def test_partition_both_directions_in_rows(hip, oracle):
    """BASELINE config #4 as written — a partition mask, both directions: the majority declares the minority dead AND the
    minority the majority.  With a row for every node both directions live in the dense store; compared with the checker
    (state, counters, swim_detection_get) every 5 s for 200 s."""
    n = 4096
    mask = np.zeros(n, dtype=np.uint8); mask[np.random.default_rng(5).choice(n, size=n // 20, replace=False)] = 1
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=8), dict(view_cap=n), n_nodes=n, seed=5, queue_cap=32, inbox_cap=2 * n, subject_cap=4)
    for s in (a, b):
        s.step_ms(1000); s.partition(0, mask)
    for sec in range(5, 200, 5):
        a.step_ms(5000); b.step_ms(5000)
        assert_same(a, b, tag=f"t={sec + 1}s")
        da, db = a.detection(0), b.detection(0)
        assert da == db, (sec, da, db)
    # the majority (3 892 probers) is through with the 204 after ~100 s; the 204 — each probing one node a second, ever slower as
    # their Lifeguard awareness rises — are still working on the 3 892
    nv = int(mask.sum())
    assert da[0] == 2 * nv * (n - nv) and nv * (n - nv) <= da[1][2] + da[1][3] < da[0] and a.stats()["view_drops"] == 0


def test_partition_heal_and_reconnect_with_both_directions_in_rows(hip, oracle):
    """config #4 as written AND its recovery phase, small (tests/scenarios.py run_partition_heal_mass; the 65 536-node fixture of
    test_scale_gpu.py is this scenario): 5 % cut off, both sides start declaring each other dead — a row for every node, so both
    directions live in the dense store — the cut heals after 60 s, serf's reconnect(), push-pull and refutations bring everybody back,
    folds hand the rows back.  Digest, counters, detection census and what a few observers of either side hold not-alive, beside
    the checker every 10 s."""
    import scenarios as sc
    n = 2048
    kw = dict(sc.PARTITION_HEAL_64K, n_nodes=n, inbox_cap=2 * n)
    a, b = pair(hip, oracle, dict(mass_rows=n, view_cap=8), dict(view_cap=n), **kw)
    cps = tuple(range(10, 201, 10))
    ra, rb = sc.run_partition_heal_mass(a, n, checkpoints=cps), sc.run_partition_heal_mass(b, n, checkpoints=cps)
    for sec in cps:
        assert ra[sec] == rb[sec], (sec, ra[sec], rb[sec])
    st = a.stats()
    assert st["view_drops"] == 0 and st["inbox_overflow"] == 0 and st["reconnects_reached"] > 0 and st["refutes"] > n // 20 and st["fold_freed"] > 0
    assert ra[60][2][0] == 2 * (n // 20) * (n - n // 20) and ra[200][2][0] == 0            # pairs out of reach: both directions, then none


@pytest.mark.parametrize("n_shards", [2, 4])
def test_mass_failure_sharded_in_process(hip, oracle, n_shards):
    """The dense store is per shard (its columns are the shard's observers): 2 and 4 shards on one device against the unsharded
    checker, records through the split tick (LocalExchange) and through the library's own mailboxes."""
    from consul_amd.dist import LibraryExchange, LocalExchange, ShardedSim
    n, nv = 8192, 400
    victims = np.random.default_rng(9).choice(n, size=nv, replace=False)
    kw = dict(n_nodes=n, seed=9, queue_cap=16, inbox_cap=2048, subject_cap=4, fold_interval_ms=20000)
    # (the mailbox exchange with FOUR shards in one process depends on each shard's stream getting a hardware queue of its own —
    # a wait kernel spins until its sources have signalled — which the runtime does not promise; the four-PROCESS case is
    # tests/test_scale_gpu.py's, here the mailboxes run with two)
    for xchg in ((LocalExchange, LibraryExchange) if n_shards == 2 else (LocalExchange,)):
        a = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, mass_rows=nv + 8, view_cap=8, **kw)) for i in range(n_shards)], xchg())
        b = Sim(oracle, preset(oracle, abi.PRESET_LAN, view_cap=nv + 64, **kw))
        for s in (a, b):
            s.step_ms(1000); s.kill(0, victims.tolist())
        for sec in range(0, 60, 4):
            a.step_ms(4000); b.step_ms(4000); a.sync()
            assert a.digest() == b.digest(), (xchg.__name__, sec)
            assert a.detection(0) == b.detection(0)
        sa, sb = a.stats(), b.stats()
        for k in ("msgs_applied", "suspicion_timeouts", "confirmations", "probe_failures", "packets_sent", "push_pulls", "folds"):
            assert sa[k] == sb[k], k
        assert sa["view_drops"] == 0
        a.close(); b.close()


def test_checkpoint_with_rows(hip, tmp_path):
    """The dense store is ordinary state: a run resumed from a checkpoint taken in the middle of a mass event equals the
    uninterrupted one."""
    n, nv = 4096, 200
    victims = np.random.default_rng(4).choice(n, size=nv, replace=False).tolist()
    kw = dict(n_nodes=n, seed=4, mass_rows=nv + 8, view_cap=8, queue_cap=16, inbox_cap=1024)
    a = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)); c = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    a.step_ms(1000); a.kill(0, victims); a.step_ms(12000)
    path = str(tmp_path / "mass.ckpt")
    a.save(path); c.load(path)
    assert a.digest() == c.digest()
    a.step_ms(30000); c.step_ms(30000)
    assert a.digest() == c.digest() and a.stats() == c.stats() and a.detection(0) == c.detection(0)
