"""Checkpoint / resume of the product library: every device array through a file and back.  The resumed HIP run is the
uninterrupted HIP run, both are the checker's run, and neither library accepts the other's file."""
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, SwimError, preset
from test_checkpoint import KW, first_half, observe, run_with_checkpoint, second_half

pytestmark = pytest.mark.gpu


def test_a_resumed_run_is_the_uninterrupted_run_on_hip(hip, oracle, tmp_path):
    want, got, ev_a, ev_b = run_with_checkpoint(hip, str(tmp_path / "hip.ck"))
    assert got == want and ev_b == ev_a and len(ev_a) > 0
    o = Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW))
    first_half(o); second_half(o)
    ref = observe(o)
    o.close()
    assert got[0] == ref[0] and got[1] == ref[1] and got[3:] == ref[3:]          # digest, clock, censuses, members, coordinate bits, trace
    for k in ("probes", "probe_acks", "refutes", "msgs_applied", "coord_updates", "user_events"):
        if k in ref[2]:
            assert got[2][k] == ref[2][k], k


def test_resume_at_bench_scale_and_from_a_quiet_phase(hip, tmp_path):
    """4 x 65 536 nodes: a checkpoint taken while the population is still pristine (k_quiet's fast path) and one taken in the
    saturated phase after a failure; 1.3 GB of arrays each way"""
    kw = dict(n_nodes=65536, n_replicas=4, seed=5, subject_cap=2, view_cap=4, queue_cap=4, inbox_cap=24)
    path = str(tmp_path / "big.ck")
    a = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    a.step(40); a.save(path); a.step(60)
    b = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
    b.load(path); b.step(60)
    assert a.digest() == b.digest()
    for s in (a, b):
        for r in range(4):
            s.kill(r, [1000 + r])
        s.step(30)
    a.save(path); a.step(250); a.sync()
    b.step(11)                                           # diverge, then come back
    b.load(path); b.step(250); b.sync()
    assert a.digest() == b.digest() and a.stats() == b.stats()
    ca, cb = a.census(2, 1002), b.census(2, 1002)
    assert (ca.first_suspect_ms, ca.first_dead_ms, ca.all_dead_ms) == (cb.first_suspect_ms, cb.first_dead_ms, cb.all_dead_ms) and ca.first_dead_ms != abi.NONE
    a.close(); b.close()


def test_files_do_not_cross_libraries(hip, oracle, tmp_path):
    kw = dict(n_nodes=64, seed=3)
    h, o = Sim(hip, preset(hip, abi.PRESET_LAN, **kw)), Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    h.step(10); o.step(10)
    h.save(str(tmp_path / "h.ck")); o.save(str(tmp_path / "o.ck"))
    for s, p in ((h, "o.ck"), (o, "h.ck")):
        with pytest.raises(SwimError) as e:
            s.load(str(tmp_path / p))
        assert e.value.rc == abi.EINVAL
    h.step(5); o.step(5)
    assert h.digest() == o.digest()                      # a refused load touches nothing
    h.close(); o.close()
