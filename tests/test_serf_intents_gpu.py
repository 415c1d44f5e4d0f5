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
