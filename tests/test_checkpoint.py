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
