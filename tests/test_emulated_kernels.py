"""The HIP kernels' own SOURCE, run where there is no GPU (tools/emu: consul_amd/csrc/*.hip compiled UNMODIFIED for the host against a
wave64 lock-step emulator of the slice of HIP they use — every lane a fiber, wave collectives resolved when all live lanes of a wave have
arrived, workgroups one after the other).  What this buys the CPU suite, which otherwise only sees the checker and the host logic:
the device code path itself — `k_begin` / `k_deliver` / `k_resolve` and friends, the LDS staging, the ballot / prefix-sum compaction,
the dense pair store, the fold and reap passes — is executed and held against the checker, tick by tick, on every run of
`pytest -m "not gpu"`; a logic slip in a kernel shows up here, before any GPU minute is spent.  What it is not: the product (the library
reports backend "hip-emulated" and consul_amd/lib.py refuses it), a model of the memory system, or a substitute for the
`-m gpu` tests (those run the same source as gfx950 code through the same C-ABI).

The whole `-m gpu` suite can be driven through it too: `SWIMSIM_EMU_SO=tools/emu/_build/libswimsim_emu.so python -m pytest tests -m gpu`
(tests/conftest.py; slow — config #3 at 1 048 576 nodes takes two minutes — but 22 of the 24 parity cases finish within 150 s each), and
`tools/emu/build.sh asan` puts every load and store the kernels make under AddressSanitizer + UBSan."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scenarios as sc                       # noqa: E402
from consul_amd import abi                   # noqa: E402
from consul_amd.sim import Sim, preset       # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tools", "emu")
EMU_SO = os.path.join(EMU, "_build", "libswimsim_emu.so")
SOURCES = [os.path.join(ROOT, "consul_amd", "csrc", f) for f in ("swim_host.hip", "swim_kernels.hip", "swim_device.h")] + \
          [os.path.join(ROOT, "include", "swimsim.h"), os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(EMU, "emu_engine.inc"), os.path.join(EMU, "build.sh")]

STAT_KEYS = ["node_rounds_active", "node_rounds_quiescent", "packets_sent", "packets_dropped", "msgs_sent", "msgs_applied", "probes", "probe_acks",
             "probe_indirect_acks", "probe_failures", "nacks_missed", "refutes", "suspicion_timeouts", "confirmations", "edges", "msgs_filtered",
             "push_pulls", "queue_drops", "inbox_overflow", "piggybacks", "msgs_piggybacked", "subject_overflow", "view_drops", "folds", "fold_freed"]


@pytest.fixture(scope="module")
def emu():
    if os.environ.get("SWIMSIM_EMU_SO"):
        return abi.bind(C.CDLL(os.environ["SWIMSIM_EMU_SO"]))
    if not os.path.exists(EMU_SO) or any(os.path.getmtime(EMU_SO) < os.path.getmtime(p) for p in SOURCES):
        subprocess.run(["bash", os.path.join(EMU, "build.sh")], check=True, stdout=subprocess.DEVNULL)      # ~1 min of one host core
    return abi.bind(C.CDLL(EMU_SO))


def counters(emu):
    c = (C.c_uint64 * 4)()
    emu.emu_counters(c)
    return dict(zip(("launches", "workgroups", "waves_parked_at_several_sites", "shuffles_from_a_lane_not_there"), c))


def pair(emu, oracle, which=abi.PRESET_LAN, emu_kw=None, oracle_kw=None, **kw):
    return Sim(emu, preset(emu, which, **dict(kw, **(emu_kw or {})))), Sim(oracle, preset(oracle, which, **dict(kw, **(oracle_kw or {}))))


def assert_same(a, b, tag=""):
    assert a.digest() == b.digest(), f"state digest differs {tag}"
    sa, sb = a.stats(), b.stats()
    for k in STAT_KEYS:
        assert sa[k] == sb[k], f"stat {k}: emulated kernels {sa[k]} checker {sb[k]} {tag}"


def test_the_emulated_build_is_not_the_product(emu, monkeypatch):
    from consul_amd import lib
    assert emu.swim_backend() == b"hip-emulated"
    monkeypatch.setattr(lib, "LIB_PATH", os.environ.get("SWIMSIM_EMU_SO") or EMU_SO)
    monkeypatch.setattr(lib, "_cdll", None)
    with pytest.raises(ImportError):
        lib.load()


def test_single_failure_tick_by_tick(emu, oracle):
    """BASELINE configs[0]'s shape (128 nodes, LAN timers, node 17 stops at t = 10 s): the edge list and the digest of every tick, the
    census, the trace, the event stream — what tests/test_parity_gpu.py::test_single_failure_lockstep_small asks of the device."""
    a, b = pair(emu, oracle, n_nodes=128, seed=1, trace_ticks=400)
    for s in (a, b):
        s.step_ms(10000); s.kill(0, [17])
    a.edges()
    for _ in range(260):
        a.step(1); b.step(1)
        assert np.array_equal(a.edges(), b.edges()), f"edge list differs at tick {a.now()[0]}"
        assert a.digest() == b.digest(), f"digest differs at tick {a.now()[0]}"
    assert_same(a, b)
    ca, cb = a.census(0, 17), b.census(0, 17)
    assert (ca.first_suspect_ms, ca.first_dead_ms, ca.all_dead_ms, list(ca.by_state)) == (cb.first_suspect_ms, cb.first_dead_ms, cb.all_dead_ms, list(cb.by_state))
    assert cb.all_dead_ms != abi.NONE
    assert np.array_equal(a.trace(0, 17, 100, 250), b.trace(0, 17, 100, 250)) and a.poll_events() == b.poll_events()
    assert np.array_equal(a.members(0, 3), b.members(0, 3))


def test_replicas_loss_refutation_partition_leave_and_rejoin(emu, oracle):
    """Three clusters of 1 024 nodes under 15 % packet loss (indirect probes, nacks, TCP fallback, Lifeguard's awareness, false suspicions
    and their refutations), a partition that heals, a graceful leave and a restart: compared every simulated second."""
    n = 1024
    a, b = pair(emu, oracle, n_nodes=n, n_replicas=3, seed=7, subject_cap=8, view_cap=16, queue_cap=8, inbox_cap=64)
    mask = np.zeros(n, dtype=np.uint8); mask[100:140] = 1
    for sec in range(1, 41):
        for s in (a, b):
            if sec == 2:
                s.set_loss(0.15)
            if sec == 5:
                s.kill(0, [11]); s.kill(1, [500]); s.leave(2, [77])
            if sec == 12:
                s.partition(1, mask)
            if sec == 25:
                s.partition(1, np.zeros(n, dtype=np.uint8)); s.revive(0, [11])
            s.step_ms(1000)
        assert_same(a, b, tag=f"after {sec} s")
    st = a.stats()
    assert st["refutes"] > 0 and st["probe_indirect_acks"] > 0 and st["nacks_missed"] > 0 and st["push_pulls"] > 0


def test_dense_pair_store_mass_failure_and_folds(emu, oracle):
    """config #4's shape, small: 5 % of 2 048 nodes stop at once; the emulated kernels keep every (survivor, victim) view in the dense pair
    store (rows, tile deadlines, k_expire_mass, k_send_mass, folds handing rows back), the checker in hash tables — to full detection."""
    n = 2048
    a, b = pair(emu, oracle, emu_kw=dict(view_cap=8, mass_rows=n // 20 + 8), oracle_kw=dict(view_cap=n // 20 + 64),
                n_nodes=n, seed=11, queue_cap=16, inbox_cap=1024, subject_cap=8, fold_interval_ms=5000)
    cps = (5, 10, 20, 30, 40)
    ra, rb = sc.run_mass_kill(a, n, cps, limit_s=400), sc.run_mass_kill(b, n, cps, limit_s=400)
    assert ra == rb and "done" in ra
    assert a.stats()["view_drops"] == 0 and ra["done"][3][0] == (n - n // 20) * (n // 20)


def test_unbounded_queue_implied_by_the_pair_store(emu, oracle, monkeypatch):
    """SWIM_F_UNBOUNDED_QUEUE on the device code: memberlist's queue implied by the dense pair store (QueueBroadcast = a store into the pair,
    GetBroadcasts = k_gossip_iq / k_piggy_iq selecting over a node's column), the checker's queues simply grow — config #4's shape to full
    detection with nothing pruned, then config #5's (churn, serf user events, folds that wait for queued rumours), the pooled inbox rows on
    (own rows of 256 messages, so that the state exchanges borrow big rows).  The GPU suite runs these and more: tests/test_unbounded_queue_gpu.py."""
    monkeypatch.setenv("SWIMSIM_INBOX_POOL_C1", "256")
    uq = abi.F_DEFAULT | abi.F_UNBOUNDED_QUEUE
    n, nv = 512, 60
    victims = np.random.default_rng(8).choice(n, size=nv, replace=False).tolist()
    a, b = pair(emu, oracle, emu_kw=dict(view_cap=8, mass_rows=nv + 8), oracle_kw=dict(view_cap=nv + 64), n_nodes=n, seed=8, queue_cap=8,
                inbox_cap=2 * nv + 300, subject_cap=4, flags=uq)
    for s in (a, b):
        s.step_ms(1000); s.kill(0, victims)
    for sec in range(0, 28, 4):
        a.step_ms(4000); b.step_ms(4000)
        assert_same(a, b, f"t={sec + 5}s")
        assert a.detection(0) == b.detection(0)
    pairs, by = a.detection(0)
    assert pairs == (n - nv) * nv and by[2] + by[3] == pairs and a.stats()["queue_drops"] == 0
    qa, qb = a.node_info(0, 1), b.node_info(0, 1)
    assert qa.queue_len == qb.queue_len and [(e.subject, e.seq, e.transmits) for e in qa.queue] == [(e.subject, e.seq, e.transmits) for e in qb.queue]
    n = 256
    kw = dict(n_nodes=n, seed=13, queue_cap=8, event_queue_cap=16, event_ids_per_ltime=62, inbox_cap=2048, subject_cap=4, fold_interval_ms=5000,
              flags=uq | abi.F_SERF_EVENTS, watch_node=0)
    a, b = pair(emu, oracle, emu_kw=dict(mass_rows=n, view_cap=4), oracle_kw=dict(view_cap=n), **kw)
    assert sc.run_churn_events(a, n, 10, events_per_s=6, checkpoints=(5, 10)) == sc.run_churn_events(b, n, 10, events_per_s=6, checkpoints=(5, 10))


def test_serf_events_intents_and_membership(emu, oracle):
    """Serf's layer on the emulated kernels: Lamport-clocked user events with dedupe, leave / join intents ordered against statusLTime,
    a node that joins a running cluster, the reaper."""
    n = 512
    kw = dict(n_nodes=n, n_initial=n - 16, seed=5, queue_cap=8, event_queue_cap=8, inbox_cap=256, subject_cap=8, view_cap=32,
              flags=abi.F_DEFAULT | abi.F_SERF_EVENTS, reap_interval_ms=5000, tombstone_timeout_ms=8000)
    try:
        a, b = pair(emu, oracle, **kw)
    except Exception as e:                       # (a preset key this ABI names differently: fail loudly, never silently skip the layer)
        pytest.fail(f"serf preset: {e}")
    for sec in range(1, 31):
        for s in (a, b):
            if sec in (2, 3, 9):
                for o in (5, 99, 300):
                    s.user_event(0, o, 1000 * sec + o)
            if sec == 4:
                s.join(0, [n - 16, n - 15, n - 3], 0)
            if sec == 6:
                s.leave(0, [40]); s.kill(0, [41])
            if sec == 15:
                s.revive(0, [40])
            s.step_ms(1000)
        assert_same(a, b, tag=f"after {sec} s")
        assert a.poll_events() == b.poll_events()
    st = a.stats()
    assert st["user_events_delivered"] > 0 and st["user_events_deduped"] > 0


def test_two_processes_over_gloo_with_the_emulated_kernels(emu):
    """The N > 1 path with the kernels' own code on both ranks: two processes, one shard each of a 2 x 2 048-node population, the framed
    exchange (swim_frame_pack -> ONE equal-split all_to_all_single over gloo -> swim_frame_deliver, what consul_amd/dist.py does over RCCL on
    the GPUs) — digest and counters summed over the ranks against the UNSHARDED run of the checker (tests/dist_worker.py)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", PYTHONPATH=ROOT, SWIMSIM_DIST_LIB=os.environ.get("SWIMSIM_EMU_SO") or EMU_SO)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29531",
           os.path.join(ROOT, "tests", "dist_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert line and "ok=True" in line[0], (line, out.stderr[-1000:])


def test_collectives_only_in_wave_uniform_control_flow(emu):
    """A property of the kernels the emulator can see and the device cannot report: in everything the tests above ran, no wave ever had
    lanes parked at two different collectives, and no shuffle read a lane that was not there."""
    c = counters(emu)
    assert c["launches"] > 1000 and c["workgroups"] > c["launches"]
    assert c["waves_parked_at_several_sites"] == 0 and c["shuffles_from_a_lane_not_there"] == 0, c


def test_the_references_outcome_tests_on_the_emulated_kernels(emu, tmp_path):
    """tests/host/test_serf_facade.cpp — the reference's own eventual-outcome tests re-staged against the C++ mirror of *serf.Serf
    (TestServer_LANReap agent/consul/server_test.go:666-733, TestAgent_ForceLeave[Prune] agent/agent_endpoint_test.go:2524-2677,
    TestServer_JoinLAN server_test.go:509, ...) — and the wire codec + Transport bridge end to end, linked against the emulated kernels
    (the gpu-marked twins link the product library)."""
    import test_host_facade as hf
    libdir = os.path.dirname(os.environ.get("SWIMSIM_EMU_SO") or EMU_SO)
    libname = os.path.basename(os.environ.get("SWIMSIM_EMU_SO") or EMU_SO)[3:-3]
    out = hf.build_and_run(tmp_path, libdir, libname)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout and "backend hip-emulated" in out.stdout, out.stdout + out.stderr
    out = hf.build_and_run(tmp_path, libdir, libname, hf.WIRE)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout and "backend hip-emulated" in out.stdout, out.stdout + out.stderr


def test_bench_legs_on_the_emulated_kernels(emu):
    """bench.py's config-#4 and config-#5 legs, at toy sizes, on the emulated kernels — the legs' own code paths
    (the dense pair store with a row per victim / per node, swim_detection_get, watch_events + poll_events, folds) end to end with the
    kernels' code, where tests/test_bench_handles.py can only run them on the checker (which has no dense store)."""
    import types
    import bench
    args = types.SimpleNamespace(seed=2, config4_nodes=1024, config4_queue_cap=16, config4_budget_s=300.0,
                                 config5_nodes=512, config5_seconds=3, config5_events=5)
    c4 = bench.run_config4(emu, args, 0)
    assert c4["detection_complete"] and c4["view_drops"] == 0 and c4["pairs"] == (1024 - 51) * 51 and c4["rounds_to_full_detection"] > 0
    c5 = bench.run_config5(emu, args, 0)
    assert "error" not in c5 and c5.get("view_drops", 0) == 0
