"""Checkpoint / resume (swim_checkpoint_save / _load): a run continued from a checkpoint is the run that was never
interrupted — digests, counters, censuses, events, coordinates — and a checkpoint is refused by anything but a handle of the same
library and configuration.  The checker's implementation here; the product library's in tests/test_checkpoint_gpu.py."""
import os

import pytest

from consul_amd import abi
from consul_amd.sim import Sim, SwimError, preset

KW = dict(n_nodes=256, n_replicas=2, n_initial=200, seed=21, subject_cap=16, view_cap=24, queue_cap=8, inbox_cap=48,
          event_queue_cap=8, event_buffer=64, trace_ticks=600,
          flags=abi.F_DEFAULT | abi.F_SERF_EVENTS | abi.F_COORDINATES, rtt_jitter_us=200)


def first_half(s):
    s.step(30)
    s.kill(0, [5, 77]); s.kill(1, [9])
    s.user_event(0, 3, 1001); s.user_event(1, 8, 1002)
    s.step(120)
    s.join(0, [210, 211], via=1); s.leave(1, [40])
    s.step(55)
    s.set_loss(0.02)
    s.join(1, [220], via=0)                              # pending at the checkpoint: started, its join push-pull not run yet


def second_half(s):
    s.step(150)
    s.revive(0, [5]); s.user_event(0, 4, 1003)
    s.step(180); s.sync()


def observe(s):
    st = s.stats()
    cen = [(c.first_suspect_ms, c.first_dead_ms, c.all_dead_ms, list(c.by_state)) for c in (s.census(0, 5), s.census(0, 77), s.census(1, 9))]
    mem = s.members(0, 1).tolist()
    co = [float(x).hex() for x in s.coordinate(0, 17).vec]
    return s.digest(), s.now(), st, cen, mem, co, s.trace(0, 5, 30, 400).tolist()


def run_with_checkpoint(lib, path, kw=KW):
    a = Sim(lib, preset(lib, abi.PRESET_LAN, **kw))
    first_half(a)
    a.save(path)
    d_at_save = a.digest()
    second_half(a)
    ev_a = a.poll_events()
    want = observe(a)
    a.close()
    b = Sim(lib, preset(lib, abi.PRESET_LAN, **kw))
    b.step(7)                                            # whatever the handle did before is overwritten
    b.load(path)
    assert b.digest() == d_at_save
    second_half(b)
    ev_b = b.poll_events()
    got = observe(b)
    b.close()
    return want, got, ev_a, ev_b


def test_a_resumed_run_is_the_uninterrupted_run(oracle, tmp_path):
    want, got, ev_a, ev_b = run_with_checkpoint(oracle, str(tmp_path / "ck.bin"))
    assert got == want
    assert ev_b == ev_a and len(ev_a) > 0                # (events recorded before the checkpoint travel with it)


def test_refusals(oracle, tmp_path):
    path = str(tmp_path / "ck.bin")
    small = dict(n_nodes=64, seed=3)
    a = Sim(oracle, preset(oracle, abi.PRESET_LAN, **small))
    a.step(20); a.save(path)
    other = Sim(oracle, preset(oracle, abi.PRESET_LAN, **dict(small, seed=4)))       # another configuration
    with pytest.raises(SwimError) as e:
        other.load(path)
    assert e.value.rc == abi.EINVAL
    with pytest.raises(SwimError) as e:
        a.load(str(tmp_path / "missing.bin"))
    assert e.value.rc == abi.EIO
    with open(path, "rb") as f:
        blob = f.read()
    with open(path, "wb") as f:
        f.write(blob[: len(blob) // 2])                 # truncated
    with pytest.raises(SwimError) as e:
        a.load(path)
    assert e.value.rc == abi.EIO
    with open(path, "wb") as f:
        f.write(b"not a checkpoint at all" * 100)
    with pytest.raises(SwimError) as e:
        other.load(path)
    assert e.value.rc == abi.EINVAL
    a.tick_begin()
    with pytest.raises(SwimError) as e:
        a.save(path)                                    # inside a tick
    assert e.value.rc == abi.ESTATE
    a.close(); other.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_schedules_resume_identically(oracle, tmp_path, seed):
    """random configuration, random stimulus before and after a checkpoint taken at a random tick"""
    import numpy as np
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([64, 128, 256]))
    kw = dict(n_nodes=n, n_replicas=int(rng.integers(1, 3)), n_initial=int(n - rng.integers(0, 9)), seed=int(rng.integers(1, 1 << 30)),
              subject_cap=16, view_cap=int(rng.choice([8, 24])), queue_cap=int(rng.choice([4, 8])), inbox_cap=64, event_queue_cap=8, event_buffer=64,
              flags=abi.F_DEFAULT | abi.F_SERF_EVENTS | (abi.F_COORDINATES if rng.integers(2) else 0),
              fold_interval_ms=int(rng.choice([0, 5000])), reap_interval_ms=int(rng.choice([0, 3000])), reconnect_timeout_ms=8000, tombstone_timeout_ms=8000)
    reps = kw["n_replicas"]

    def schedule(k):
        out = []
        for _ in range(k):
            out.append((int(rng.integers(5, 60)), int(rng.integers(7)), int(rng.integers(reps)), int(rng.integers(kw["n_initial"])), int(rng.integers(1 << 20))))
        return out

    def play(s, sched):
        for ticks, op, r, x, ev in sched:
            s.step(ticks)
            try:
                if op == 0: s.kill(r, [x])
                elif op == 1: s.revive(r, [x])
                elif op == 2: s.leave(r, [x])
                elif op == 3: s.user_event(r, x, ev)
                elif op == 4: s.set_loss(0.0 if ev & 1 else 0.05)
                elif op == 5: s.join(r, [x], via=(x + 1) % kw["n_initial"])
                else: s.update(r, [x])
            except SwimError:
                pass                                     # e.g. an event from a node that is down: the same refusal on both runs
    before, after = schedule(int(rng.integers(3, 9))), schedule(int(rng.integers(3, 9)))
    path = str(tmp_path / "r.ck")
    a = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    play(a, before); a.save(path); d0 = a.digest(); play(a, after); a.step(50); a.sync()
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    b.load(path)
    assert b.digest() == d0
    play(b, after); b.step(50); b.sync()
    assert a.digest() == b.digest() and a.stats() == b.stats() and a.poll_events() == b.poll_events()
    a.close(); b.close()
