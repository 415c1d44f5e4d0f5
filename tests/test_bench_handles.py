"""bench.py spreads the independent clusters of config #2 over several library handles (one stream each on the GPU): the
clusters must be the ones a single handle would simulate — replica r of seed s is replica 0 of seed s + r."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import MultiSim, victims_for      # noqa: E402
from consul_amd import abi                   # noqa: E402
from consul_amd.sim import Sim, preset       # noqa: E402


def test_clusters_spread_over_handles_are_the_clusters_of_one_handle(oracle):
    n, reps, seed = 512, 5, 11                        # 5 clusters on 3 handles: 2 + 2 + 1, like 32 on 3 = 11 + 11 + 10
    kw = dict(n_nodes=n, subject_cap=2, view_cap=4, queue_cap=4, inbox_cap=24)
    one = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_replicas=reps, seed=seed, **kw))
    sizes, first = [2, 2, 1], [0, 2, 4]
    two = MultiSim([Sim(oracle, preset(oracle, abi.PRESET_LAN, n_replicas=sizes[g], seed=seed + first[g], **kw)) for g in range(3)], first)
    victims = victims_for(seed, reps, n)
    for s in (one, two):
        s.step(20)
        for r, v in enumerate(victims):
            s.kill(r, [v])
        s.step(300); s.sync()
    for r, v in enumerate(victims):
        a, b = one.census(r, v), two.census(r, v)
        assert (a.first_suspect_ms, a.first_dead_ms, a.all_dead_ms, list(a.by_state)) == (b.first_suspect_ms, b.first_dead_ms, b.all_dead_ms, list(b.by_state))
        assert a.all_dead_ms != abi.NONE
    one.close(); two.close()


class _OneRank:
    """torch.distributed's part in bench.run_config4_sharded for a world of one (the leg's control flow on the checker)."""
    @staticmethod
    def all_gather_object(out, obj):
        out[0] = obj


def _leg(oracle, nodes, monkeypatch):
    import types
    import bench
    monkeypatch.setenv("SWIMSIM_BENCH_C4S_NODES", str(nodes))
    args = types.SimpleNamespace(seed=3)
    return bench.run_config4_sharded(oracle, args, 0, 1, 0, _OneRank, lambda mine: [mine[0]], lambda: None, lambda x: x)


def test_sharded_config4_leg_control_flow_on_the_checker(oracle, monkeypatch):
    """The N > 1 config-#4 leg of bench.py, world of one, on the checker (which has no dense store: its tables of 8 do drop
    views — what is checked here is the leg's bookkeeping: phases, the gathered payloads, the fields of its object)."""
    c = _leg(oracle, 4096, monkeypatch)
    assert "error" not in c, c
    assert c["n_nodes"] == 4096 and c["victims"] == 204 and c["pairs"] == (4096 - 204) * 204 and c["inbox_cap"] == 16384      # (unbounded queue since round 6: a state exchange hands over more)
    assert c["wall_s"] >= 0 and c["rounds_per_sec"] > 0 and c["inbox_overflow"] == 0 and 0 < c["suspect_fraction"] + c["dead_fraction"] <= 1


def test_sharded_config4_leg_reports_a_failure_instead_of_raising(oracle, monkeypatch):
    """A rank that cannot even create its shard (here: a population of one, which no configuration accepts) makes every rank give
    the leg up together; the bench line carries the error, the headline number is not lost."""
    c = _leg(oracle, 1, monkeypatch)
    assert c["ranks_failed"] == 1 and "create" in c["error"] and "wall_s" not in c


def test_partition_leg_control_flow_and_wall_time_budget_on_the_checker(oracle):
    """bench.run_config4_partition (config #4 as written + recovery) on the checker at 1 024 nodes (tables of 8: views are dropped —
    what is checked is the leg's bookkeeping), once to the end and once with a wall-time budget of nothing: the leg gives up after the
    first 10 s of the cut and says so instead of holding the bench line up."""
    import types
    import bench
    args = types.SimpleNamespace(seed=5, config4p_nodes=1024, config4p_budget_s=600.0)
    c = bench.run_config4_partition(oracle, args, 0)
    assert c["n_nodes"] == 1024 and c["cut_off"] == 51 and c["at_heal"]["pairs_out_of_reach"] == 2 * 51 * (1024 - 51)
    assert c["gave_up_on_wall_time_budget_s"] is None and c["curve"] and c["curve"][0]["t_s"] == 90 and c["refutes"] > 0
    assert c["simulated_s"] == c["curve"][-1]["t_s"] and (c["recovered_for_the_watchers_at_s"] in (None, c["simulated_s"]))
    args.config4p_budget_s = 0.0
    g = bench.run_config4_partition(oracle, args, 0)
    assert "gave_up" in g and "curve" not in g
