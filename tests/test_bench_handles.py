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
