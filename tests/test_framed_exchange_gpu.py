"""The framed exchange on the product library (swimsim.h swim_frame_pack / swim_frame_deliver; consul_amd/dist.py TorchExchange):
the kernels that write and read the frames, on one device — several shards of a population in this process, their frames transposed
with device copies where the collective would move them between processes.  Nothing is read back by the host between begin and end.
Results: the unsharded checker's, bit for bit, counters included.  (The collective itself needs one process per device: the same
Python code path runs over gloo on the checker in tests/test_dist_cpu.py.)"""
import pytest

from consul_amd import abi
from consul_amd.dist import LocalFramedExchange, ShardedSim
from consul_amd.sim import Sim, preset

pytestmark = pytest.mark.gpu

STAT_KEYS = ("msgs_sent", "msgs_applied", "msgs_filtered", "edges", "refutes", "probe_failures", "packets_sent", "confirmations",
             "suspicion_timeouts", "piggybacks", "msgs_piggybacked", "push_pulls", "folds")


class HipRuntime:
    """hipMalloc / hipMemcpy / hipDeviceSynchronize of the HIP runtime the product library itself is linked against (found among this
    process's mappings once the library is loaded) — torch is kept out of this process: it brings a HIP runtime of its own, and whichever
    of the two initialises second sees no device."""

    def __init__(self):
        import ctypes as C
        path = next(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l)
        self.C, self.l = C, C.CDLL(path)
        self.bufs = []

    def _ck(self, rc, what):
        assert rc == 0, f"{what} -> hipError {rc}"

    def alloc(self, n_records):
        p = self.C.c_void_p()
        self._ck(self.l.hipMalloc(self.C.byref(p), self.C.c_size_t(n_records * 16)), "hipMalloc")
        self._ck(self.l.hipMemset(p, 0, self.C.c_size_t(n_records * 16)), "hipMemset")
        self.bufs.append(p)
        return p.value

    def copy(self, dst, d0, src, s0, n):           # device to device, record offsets
        self._ck(self.l.hipMemcpy(self.C.c_void_p(dst + 16 * d0), self.C.c_void_p(src + 16 * s0), self.C.c_size_t(16 * n), 3), "hipMemcpy")

    def sync(self):
        self._ck(self.l.hipDeviceSynchronize(), "hipDeviceSynchronize")

    def record(self, ptr, i):                      # record i of a device buffer, as four unsigned words
        out = (self.C.c_uint32 * 4)()
        self._ck(self.l.hipMemcpy(out, self.C.c_void_p(ptr + 16 * i), self.C.c_size_t(16), 2), "hipMemcpy")
        return list(out)

    def free(self):
        self.sync()
        for p in self.bufs:
            self.l.hipFree(p)
        self.bufs = []


def device_frames(rt, frame_records=None, adaptive=False):
    return LocalFramedExchange(alloc=rt.alloc, ptr=lambda b: b, copy=rt.copy, sync=rt.sync, frame_records=frame_records, read=rt.record if adaptive else None)


@pytest.mark.parametrize("n_shards", [2, 4])
def test_framed_exchange_on_hip_matches_the_unsharded_checker(hip, oracle, n_shards):
    kw = dict(n_nodes=4096, n_replicas=2, seed=5, subject_cap=256, view_cap=256, queue_cap=16, inbox_cap=1024,
              loss_q32=int(0.05 * 2**32), fold_interval_ms=5000, flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    sims = [Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **kw)) for i in range(n_shards)]
    F = sims[0].frame_records()
    assert F == 1 + max(sims[0].outbound_capacity(j) for j in range(1, n_shards)) and all(s.frame_records() == F for s in sims)
    rt = HipRuntime()
    sh = ShardedSim(sims, device_frames(rt))
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(3000)
        s.kill(0, [100, 3000]); s.kill(1, [7]); s.update(1, [2048])
        s.step_ms(30000)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in STAT_KEYS:
        assert a[k] == b[k], k
    assert a["edges_remote"] > 0 and a["folds"] >= 1
    sh.close(); rt.free()


def test_frames_sized_from_the_load_on_hip(hip, oracle):
    """swim_frame_pack_fill on the device: frames of 64 records that follow the load, a tick that does not fit packed and moved again before it
    is delivered — the unsharded checker's digest and counters, with retries on the way and no overflow."""
    kw = dict(n_nodes=4096, n_replicas=2, seed=5, subject_cap=256, view_cap=256, queue_cap=16, inbox_cap=1024,
              loss_q32=int(0.05 * 2**32), fold_interval_ms=5000, flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    sims = [Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)]
    rt = HipRuntime()
    ex = device_frames(rt, adaptive=True)
    sh = ShardedSim(sims, ex)
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    sizes = set()
    for s in (sh, ref):
        s.step_ms(3000)
        s.kill(0, [100, 3000]); s.kill(1, [7]); s.update(1, [2048])
    for _ in range(30):
        sh.step_ms(1000); ref.step_ms(1000); sizes.add(ex._F)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in STAT_KEYS:
        assert a[k] == b[k], k
    assert ex.retries > 0 and min(sizes) < max(sizes) < sims[0].frame_records()
    sh.close(); rt.free()


def test_framed_exchange_keeps_a_quiet_cluster_off_the_orders_and_wakes_up(hip, oracle):
    """The activity word travels in the frame headers and is folded into the next tick's hint ON THE DEVICE (the host never
    sees it): a quiescent population files no piggy-back orders for other shards, and a stimulus after any number of quiet
    ticks still reproduces the unsharded checker (the host raises the word again: swim_peer_activity's contract)."""
    kw = dict(n_nodes=4096, seed=8, subject_cap=64, view_cap=64, queue_cap=16, inbox_cap=1024, push_pull_interval_ms=0)
    rt = HipRuntime()
    sh = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)], device_frames(rt))
    ref = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    for s in (sh, ref):
        s.step_ms(3000)
    assert sh.stats()["edges_remote"] == 0 and sh.digest() == ref.digest()
    for s in (sh, ref):
        s.update(0, [5]); s.step(1); s.update(0, [4000]); s.step_ms(2000)
    assert sh.digest() == ref.digest()
    for s in (sh, ref):
        s.step_ms(20000)                                     # everything retires: quiet again
        s.kill(0, [2500]); s.step_ms(40000)
        s.leave(0, [77]); s.step_ms(10000)
    sh.sync()
    assert sh.digest() == ref.digest()
    a, b = sh.stats(), ref.stats()
    for k in ("piggybacks", "msgs_piggybacked", "msgs_sent", "msgs_applied", "probe_failures", "suspicion_timeouts"):
        assert a[k] == b[k], k
    assert a["piggybacks"] > 0
    sh.close(); rt.free()


def test_a_frame_too_small_raises_the_sticky_overflow_and_a_stale_frame_is_refused(hip):
    kw = dict(n_nodes=4096, seed=3)
    sims = [Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)]
    rt = HipRuntime()
    sh = ShardedSim(sims, device_frames(rt, frame_records=4))     # the probes' piggy-back orders alone need more than three records a tick
    with pytest.raises(Exception, match="edge-list"):
        sh.update(0, [1, 3000]); sh.step_ms(2000); sh.sync()
    sh.close()
    a, b = [Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=2, **kw)) for i in range(2)]
    F = 256
    sa, sb, ra = rt.alloc(2 * F), rt.alloc(2 * F), rt.alloc(2 * F)
    with pytest.raises(Exception):
        a.frame_pack(sa, F)                                  # no tick open
    a.tick_begin(); b.tick_begin()
    a.frame_pack(sa, F); b.frame_pack(sb, F); a.sync(); b.sync()
    h = rt.record(sa, F)                                     # header of shard 0's frame for shard 1: {count, activity, tick + 1, magic}
    assert h[0] <= F - 1 and h[2] == 1 and h[3] == abi.FRAME_MAGIC and rt.record(sa, 0)[0] == 0     # (the own frame is empty)
    rt.copy(ra, F, sb, 0, F); rt.sync()                      # shard 1's frame for shard 0 -> slot 1 of shard 0's receive buffer
    a.frame_deliver(ra, F); a.tick_end_begin(); a.sync()
    a.frame_deliver(ra, F)                                   # last tick's frame
    with pytest.raises(Exception, match="did not deliver|tick"):
        a.sync()
    a.close(); b.close(); rt.free()


def test_torch_exchange_over_rccl_on_the_simulators_stream(hip):
    """consul_amd/dist.py TorchExchange with backend nccl (= RCCL), as bench.py --exchange rccl drives it: pack -> all_to_all_single ->
    deliver, all on the simulator's own HIP stream, nothing read back between begin and end.  One rank per device: on a one-GPU box
    that is a world of one (its only frame is its own, empty) — the collective, the stream hand-over and the frame kernels still run,
    and the run must equal the plain unsharded one."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_worker.py")], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, PYTHONPATH=root, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1"))
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")]
    assert out.returncode == 0 and line, (out.stdout[-500:], out.stderr[-2000:])
    if "nccl-unavailable" in line[0]:
        pytest.skip(line[0])
    assert "ok=True" in line[0], line[0]
