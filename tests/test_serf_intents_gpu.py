"""tests/test_serf_intents.py's scripts on the HIP library beside the checker: same statuses, same digest — with the members' views in the
observers' hash tables and in rows of the dense pair store (statusLTime is a word beside the view entry there, a fourth plane here)."""
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

import test_serf_intents as ti

pytestmark = pytest.mark.gpu

SCRIPTS = [ti.script_refuted_leave, ti.script_stale_leave, ti.script_rejoin_after_force_leave, ti.script_graceful_leave]


@pytest.mark.parametrize("script", SCRIPTS, ids=lambda f: f.__name__)
@pytest.mark.parametrize("rows", [0, 64], ids=["tables", "rows"])
def test_intent_scripts_on_hip_match_the_checker(hip, oracle, script, rows):
    kw = dict(ti.KW)
    a = Sim(hip, preset(hip, abi.PRESET_LAN, mass_rows=rows, **kw))
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, **kw))
    oa, ob = script(a), script(b)
    assert oa == ob
    sa, sb = a.stats(), b.stats()
    for k in ("user_events_deduped", "user_events_stale", "event_drops", "msgs_applied", "refutes", "intents_applied", "reaped", "msgs_sent", "packets_sent"):
        assert sa[k] == sb[k], k
    assert a.poll_events() == b.poll_events()


@pytest.mark.parametrize("n_shards", [1, 2])
def test_joins_with_intents_on_hip(hip, oracle, n_shards):
    """serf.Join's intent reaches the joiner in the answer to its join push-pull, stamped with the clock of the member it joined through —
    a message, so one shard or two (joiner and `via` on different ones) give the unsharded checker's digests step by step."""
    from consul_amd.dist import LocalExchange, ShardedSim
    b = Sim(oracle, preset(oracle, abi.PRESET_LAN, **ti.JOIN_KW))
    if n_shards == 1:
        a = Sim(hip, preset(hip, abi.PRESET_LAN, **ti.JOIN_KW))
    else:
        a = ShardedSim([Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=i, n_shards=n_shards, **ti.JOIN_KW)) for i in range(n_shards)], LocalExchange())
    assert ti.script_joins(a) == ti.script_joins(b)
    sa, sb = a.stats(), b.stats()
    for k in ("msgs_sent", "msgs_applied", "packets_sent", "edges", "msgs_filtered", "user_events_deduped", "joins", "intents_applied"):
        assert sa[k] == sb[k], k
