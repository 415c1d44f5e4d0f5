"""memberlist.Transport at rumour granularity: one node of the pool is driven from
outside — the shape of agent/consul/wanfed/wanfed.go's Transport (WriteToAddress / PacketCh) without the
msgpack framing.  The simulator stops acting for the attached node, peers keep seeing it alive, what they
gossip to it is captured with its sender, and what it writes lands in the peers' inboxes."""
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

KW = dict(n_nodes=512, seed=3, subject_cap=8, push_pull_interval_ms=0)


def drive(s):
    out = {}
    assert s.transport_poll(0, 7) == []                      # first call attaches node 7
    s.step_ms(1000); s.kill(0, [100]); s.step_ms(8000)
    out["heard"] = s.transport_poll(0, 7)
    # the real node tells peers 9 and 300 that node 200 looks suspect to it
    s.transport_write_to(0, 7, 9, [(200, 1, abi.MSG_SUSPECT, 7)])
    s.transport_write_to(0, 7, 300, [(200, 1, abi.MSG_SUSPECT, 7)])
    s.step_ms(4000)
    out["after"] = s.transport_poll(0, 7)
    out["inc200"] = s.node_info(0, 200).incarnation
    out["view"] = [s.view(0, o, 200).incarnation for o in (9, 300, 50)]
    out["digest"] = s.digest()
    out["idle"] = (s.node_info(0, 7).probe_cursor, s.node_info(0, 7).queue_len)
    return out


def test_bridge_semantics(oracle):
    r = drive(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    heard = r["heard"]
    assert heard and all(subj == 100 and typ == abi.MSG_SUSPECT for _, subj, _, typ, _ in heard)
    assert all(0 <= src < 512 and src != 7 for src, *_ in heard)            # Packet.From = a virtual peer
    assert r["inc200"] == 2                       # 200 heard the accusation and refuted it
    assert r["view"] == [2, 2, 2]                 # ...and the refutation spread
    assert any(subj == 200 and typ == abi.MSG_ALIVE and inc == 2 for _, subj, inc, typ, _ in r["after"])
    assert r["idle"] == (0, 0)                    # the simulator never probed or gossiped on 7's behalf


@pytest.mark.gpu
def test_bridge_hip_matches_oracle(hip, oracle):
    a = drive(Sim(hip, preset(hip, abi.PRESET_LAN, **KW)))
    b = drive(Sim(oracle, preset(oracle, abi.PRESET_LAN, **KW)))
    assert a == b
