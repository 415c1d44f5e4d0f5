"""The N>1 path on CPU: every replica's population block-partitioned over shards.

 * in-process: 2 and 4 shards of the oracle, records handed over by pointer (LocalExchange);
 * in-process: the framed exchange (swim_frame_pack / swim_frame_deliver) with the frames transposed by plain copies;
 * two real processes over torch.distributed/gloo: one equal-split all_to_all_single of frames per tick, the counts
   in the frames' headers (TorchExchange — the code path bench.py uses with the nccl backend on GPUs; frames sized from the load).
Both must reproduce the unsharded oracle bit for bit (digests add up; counters add up)."""
import os
import subprocess
import sys

import pytest

from consul_amd import abi
import numpy as np

from consul_amd.dist import LibraryExchange, LocalExchange, LocalFramedExchange, ShardedSim
from consul_amd.sim import Sim, preset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_shards", [2, 4])
def test_in_process_shards_match_unsharded(oracle, n_shards):
    kw = dict(n_nodes=2048, n_replicas=2, seed=5, subject_cap=128, view_cap=128, queue_cap=16, inbox_cap=128,
              loss_q32=int(0.05 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw))
                     for i in range(n_shards)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(3000)
        s.kill(0, [100, 1500]); s.kill(1, [7]); s.update(1, [1024])
        s.step_ms(20000)
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    # ... "edges" and "msgs_filtered" included: a rumour that crosses a shard boundary is judged by the no-op filter where it arrives
    # (round 4; before, a shard delivered whatever came from another shard and the two counters differed from the unsharded run's)
    for k in ("msgs_sent", "refutes", "probe_failures", "msgs_applied", "packets_sent", "confirmations", "edges", "msgs_filtered"):
        assert a[k] == b[k], k
    assert a["edges_remote"] > 0 and b["edges_remote"] == 0


def test_queries_on_a_sharded_population_answer_like_the_unsharded_one(oracle):
    """ShardedSim.census merges the shards' censuses (counts add up, first = earliest, all = latest and only when every shard is
    there); view / members / node_info go to the shard that owns the observer."""
    kw = dict(n_nodes=2048, seed=14, subject_cap=16, view_cap=64, queue_cap=16, inbox_cap=256)
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=4, **kw)) for i in range(4)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    fields = ("n_observers", "first_suspect_ms", "first_dead_ms", "all_dead_ms", "all_current_ms")
    for t in (3000, 12000, 14000, 30000):                    # before anything, suspected, verdicts spreading, settled
        for s in (sh, ref):
            s.step_ms(t - (0 if t == 3000 else {12000: 3000, 14000: 12000, 30000: 14000}[t]))
            if t == 3000:
                s.kill(0, [700]); s.update(0, [1500])
        for x in (700, 1500):
            a, b = sh.census(0, x), ref.census(0, x)
            assert list(a.by_state) == list(b.by_state) and all(getattr(a, f) == getattr(b, f) for f in fields), (t, x)
    assert ref.census(0, 700).all_dead_ms != abi.NONE and ref.census(0, 1500).all_current_ms != abi.NONE
    assert sh.census(0, 1500).n_current == ref.census(0, 1500).n_current == 2046      # (comparable once the new incarnation has reached every shard)
    for o in (3, 600, 1100, 2047):                          # one observer per shard
        va, vb = sh.view(0, o, 700), ref.view(0, o, 700)
        assert (va.state, va.incarnation, va.state_change_ms) == (vb.state, vb.incarnation, vb.state_change_ms) and va.state == abi.STATE_DEAD
        assert (sh.members(0, o) == ref.members(0, o)).all()
        ia, ib = sh.node_info(0, o), ref.node_info(0, o)
        assert (ia.incarnation, ia.awareness, ia.queue_len, ia.probe_target) == (ib.incarnation, ib.awareness, ib.queue_len, ib.probe_target)
    one = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=1, n_shards=4, **kw))], LocalExchange())
    with pytest.raises(LookupError):
        one.view(0, 3, 700)                                 # shard 0 is in another process


def test_tcp_classes_are_ground_truth_on_every_shard(oracle):
    """swim_set_tcp_class is replicated like a partition mask: a prober on one shard and its target on another compare their
    classes exactly as an unsharded run does (two datacenters under 25 % loss, TCP fallback on)."""
    kw = dict(n_nodes=1024, seed=8, subject_cap=256, view_cap=1024, queue_cap=16, inbox_cap=2048, loss_q32=int(0.25 * 2**32))
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.set_tcp_class(0, range(0, 1024, 2), 3)                     # classes interleaved: most probes cross a shard AND a class
        s.step_ms(15000)
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    assert a["inbox_overflow"] == b["inbox_overflow"] == 0     # (a shard cannot filter what another shard sends it: room for all of it)
    assert a["probe_tcp_acks"] == b["probe_tcp_acks"] > 0 and a["probe_failures"] == b["probe_failures"] > 0 and a["refutes"] == b["refutes"]


def numpy_frames(frame_records=None, adaptive=False):
    """LocalFramedExchange over host memory (the checker)."""
    def copy(dst, d0, src, s0, n):
        dst[d0:d0 + n] = src[s0:s0 + n]
    return LocalFramedExchange(alloc=lambda n: np.zeros((n, 4), dtype=np.uint32), ptr=lambda b: b.ctypes.data, copy=copy, frame_records=frame_records,
                               read=(lambda b, i: [int(x) for x in b[i]]) if adaptive else None)


@pytest.mark.parametrize("n_shards", [2, 4])
def test_framed_exchange_matches_unsharded(oracle, n_shards):
    """swim_frame_pack / swim_frame_deliver: per-destination frames {header: count, activity, tick + 1, magic; records}, transposed
    between the shards like an equal-split all-to-all does — same run as with the pointer hand-over, same as unsharded; fold ticks
    and push-pull included."""
    kw = dict(n_nodes=1024, n_replicas=2, seed=9, fold_interval_ms=3000, push_pull_interval_ms=2000, view_cap=64,
              loss_q32=int(0.05 * 2**32))
    mk = lambda ex: ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw)) for i in range(n_shards)], ex)
    fr, ptr, ref = mk(numpy_frames()), mk(LocalExchange()), Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (fr, ptr, ref):
        s.step_ms(1000); s.kill(0, [5, 700]); s.update(1, [512]); s.step_ms(30000)
    assert fr.digest() == ptr.digest() == ref.digest()
    a, b, c = fr.stats(), ptr.stats(), ref.stats()
    for k in ("msgs_sent", "refutes", "msgs_applied", "packets_sent", "edges", "msgs_filtered", "folds", "push_pulls"):
        assert a[k] == b[k] == c[k], k
    assert a["edges_remote"] == b["edges_remote"] > 0


def test_frames_sized_from_the_load_lose_nothing(oracle):
    """swim_frame_pack_fill (round 5; ADVICE r4: the bound's frames cost megabytes a tick whatever the fill): frames start at 64 records and
    follow the load; a tick whose largest segment does not fit is packed and moved again with frames that hold it BEFORE anything is
    delivered — same digest and counters as the pointer hand-over and the unsharded run, retries included, no overflow anywhere."""
    kw = dict(n_nodes=2048, n_replicas=2, seed=9, fold_interval_ms=3000, push_pull_interval_ms=2000, view_cap=64, loss_q32=int(0.05 * 2**32))
    ex = numpy_frames(adaptive=True)
    mk = lambda e: ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], e)
    fr, ref = mk(ex), Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    sizes = set()
    for s in (fr, ref):
        s.step_ms(1000); s.kill(0, [5, 700]); s.update(1, [512])
    for _ in range(30):
        fr.step_ms(1000); ref.step_ms(1000); sizes.add(ex._F)
    assert fr.digest() == ref.digest()
    a, c = fr.stats(), ref.stats()
    for k in ("msgs_sent", "refutes", "msgs_applied", "packets_sent", "edges", "msgs_filtered", "folds", "push_pulls"):
        assert a[k] == c[k], k
    assert ex.retries > 0 and len(sizes) > 1 and min(sizes) < max(sizes)       # the frames grew with the saturated phase and shrank after it
    # the header's contract, by hand: count = what the segment HAS, word 1 = activity | need << 1
    x, y = [Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, n_nodes=512, seed=3)) for i in range(2)]
    xy = ShardedSim([x, y], numpy_frames())
    xy.update(0, [1, 300]); xy.step_ms(1500)           # two rumours in full dissemination
    x.tick_begin(); y.tick_begin()
    F = 4
    sa = np.zeros((2 * F, 4), dtype=np.uint32)
    x.frame_pack_fill(sa.ctypes.data, F)
    need = int(sa[0][1]) >> 1
    assert int(sa[F][0]) == need > F - 1 and int(sa[F][1]) & 1 and tuple(sa[F][2:]) == (x.stats()["ticks"] + 1, abi.FRAME_MAGIC)
    ra = np.zeros((2 * F, 4), dtype=np.uint32); ra[F:] = sa[F:]
    with pytest.raises(Exception, match="tick|frame"):
        x.frame_deliver(ra.ctypes.data, F)               # a truncated frame is never delivered


def test_framed_exchange_refuses_what_it_cannot_carry(oracle):
    kw = dict(n_nodes=512, seed=3, push_pull_interval_ms=1000)
    sims = [Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)]
    assert sims[0].frame_records() == 0                 # the checker's lists are unbounded: the caller picks a frame size
    sh = ShardedSim(sims, numpy_frames(frame_records=4))
    with pytest.raises(Exception, match="overflow|EOVERFLOW|frame"):   # the probes' piggy-back orders alone are more than three records a tick
        sh.step_ms(5000)
    # frames of another tick (a shard that did not step) are refused
    a, b = [Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)]
    F = 64
    sa, sb = np.zeros((2 * F, 4), dtype=np.uint32), np.zeros((2 * F, 4), dtype=np.uint32)
    with pytest.raises(Exception):
        a.frame_pack(sa.ctypes.data, F)                 # no tick open
    a.tick_begin(); b.tick_begin()
    a.frame_pack(sa.ctypes.data, F); b.frame_pack(sb.ctypes.data, F)
    assert tuple(sa[F]) [2:] == (1, abi.FRAME_MAGIC) and sa[0][0] == 0      # header of the frame for shard 1: tick + 1, magic; the own frame is empty
    ra = np.zeros((2 * F, 4), dtype=np.uint32); ra[F:] = sb[:F]
    a.frame_deliver(ra.ctypes.data, F)
    a.tick_end_begin()
    with pytest.raises(Exception, match="tick"):
        a.frame_deliver(ra.ctypes.data, F)              # last tick's frame
    with pytest.raises(Exception):
        a.frame_pack(sa.ctypes.data, 1)                 # a frame holds a header and at least one record


@pytest.mark.parametrize("n_shards", [2, 4])
def test_library_exchange_calls_on_the_oracle(oracle, n_shards):
    """swim_xchg_export / connect / step (the product library's device-driven exchange) as the oracle implements them for
    shards of one process: same calls, same result as the unsharded run — including fold ticks."""
    kw = dict(n_nodes=1024, n_replicas=2, seed=9, fold_interval_ms=3000, push_pull_interval_ms=2000, view_cap=64)
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw)) for i in range(n_shards)],
                    LibraryExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(1000); s.kill(0, [5, 700]); s.update(1, [512]); s.step_ms(50000)
    assert sh.digest() == ref.digest() and sh.stats()["folds"] == ref.stats()["folds"] >= 1
    lone = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=256, n_shards=2, shard_rank=0))
    with pytest.raises(Exception):
        lone.xchg_step(1)                           # not connected
    with pytest.raises(Exception):
        sh.sims[0].xchg_connect([lone.xchg_export()] * n_shards)   # a handle of another population


def test_split_tick_rejects_out_of_phase_calls(oracle):
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=256, n_shards=2, shard_rank=0))
    with pytest.raises(Exception):
        s.step(1)                                   # swim_step is single-shard only
    with pytest.raises(Exception):
        s.tick_end()                                # no tick open
    s.tick_begin()
    with pytest.raises(Exception):
        s.tick_begin()
    with pytest.raises(Exception):
        s.kill(0, [1])                              # stimulus only between ticks
    s.tick_end()


def test_two_processes_over_gloo_match_unsharded(oracle):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line and "ok=True" in line[0], (line, out.stderr[-1000:])


def test_randomised_sharding_cases(oracle):
    """tools/fuzz_parity.py with the oracle on both sides: random configurations and stimulus, 1-4 in-process shards against
    the unsharded run (also exercises swim_tick_end_begin, the activity hint and the transport bridge of the oracle)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    tally = {}
    for k in range(10):
        res = fz.run_case(k, oracle, oracle, 7, False)
        tally[res] = tally.get(res, 0) + 1
    assert not tally.get("mismatch") and tally.get("ok", 0) >= 5, tally


def test_join_intents_are_the_same_on_a_sharded_population(oracle):
    """serf.Join's intent is handed to the joiner by the member it joined through, ready-stamped with that member's clock, in the answer to
    the join push-pull — a message, so a joiner and its `via` on DIFFERENT shards behave like on one (a clock read at injection time would
    not: round 4 had that for a few hours).  Force-leave a failed member, let it rejoin through a node of the other shard, then an
    ordinary join of a member that starts late: digests, members and counters equal to the unsharded run's at every step."""
    kw = dict(n_nodes=1024, n_initial=1000, seed=17, flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, subject_cap=64, view_cap=256, queue_cap=16,
              event_queue_cap=16, inbox_cap=1024, push_pull_interval_ms=0)
    sh = ShardedSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], LocalExchange())
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))

    def both(f):
        f(sh); f(ref)
        assert sh.digest() == ref.digest()

    both(lambda s: s.step_ms(2000))
    both(lambda s: (s.kill(0, [700]), s.step_ms(30000)))                       # node 700 (shard 1) fails and is declared Failed
    both(lambda s: (s.force_leave(0, 3, 700, False), s.step_ms(4000)))          # ... and force-left: Left everywhere
    assert int(ref.view(0, 10, 700).status) == abi.MEMBER_LEFT
    both(lambda s: (s.join(0, [700], via=5), s.step_ms(6000)))                  # it comes back through node 5 (shard 0)
    assert int(ref.view(0, 10, 700).status) == abi.MEMBER_ALIVE and int(ref.view(0, 900, 700).status) == abi.MEMBER_ALIVE
    both(lambda s: (s.join(0, [1010], via=600), s.step_ms(6000)))               # a member that starts late: shard 1, via a node of shard 1
    both(lambda s: (s.join(0, [1011], via=2), s.step_ms(6000)))                 # ... and one via a node of shard 0
    assert int(ref.view(0, 10, 1011).status) == abi.MEMBER_ALIVE
    a, b = sh.stats(), ref.stats()
    for k in ("msgs_sent", "msgs_applied", "packets_sent", "edges", "msgs_filtered", "user_events_deduped", "joins", "intents_applied"):
        assert a[k] == b[k], k
