"""bench.py's yardstick (`roofline.achieved` = algorithmic bytes / launch duration) against DESIGN.md §6, on the checker's
counters: the per-kernel split must add up to what the unsplit formula of SURVEY §8(d) gives, and a quiet run must price
16 bytes per node-round plus 40 per probe and nothing else."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bench import algorithmic_bytes, diff_stats          # noqa: E402
from consul_amd import abi                                # noqa: E402
from consul_amd.sim import Sim, preset                    # noqa: E402


def test_a_quiet_cluster_costs_its_headers_and_probes(oracle):
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=1024, seed=3))
    s0 = s.stats(); s.step(100); st = diff_stats(s0, s.stats())
    assert st["node_rounds_active"] == 0 and st["packets_sent"] == 0 and st["probes"] > 0
    assert algorithmic_bytes("k_begin", st) == 16.0 * st["node_rounds_quiescent"] + 40.0 * st["probes"]
    assert algorithmic_bytes("k_deliver", st) == 0.0 and algorithmic_bytes("k_resolve", st) == 0.0
    s.close()


def test_the_split_by_kernel_adds_up(oracle):
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=2048, seed=5, subject_cap=4))
    s.step(20); s.kill(0, [77]); s0 = s.stats(); s.step(260); st = diff_stats(s0, s.stats())
    msgs, applied = sum(st["msgs_sent"]), sum(st["msgs_applied"])
    assert st["node_rounds_active"] > 0 and msgs > 0 and st["msgs_filtered"] > 0 and st["piggybacks"] > 0
    total = sum(algorithmic_bytes(k, st) for k in ("k_begin", "k_deliver", "k_resolve"))
    gm = msgs - st["msgs_piggybacked"]
    unsplit = (16.0 * (st["node_rounds_active"] + st["node_rounds_quiescent"]) + 8.0 * (gm / st["packets_sent"]) * st["node_rounds_active"]
               + 4.0 * st["packets_sent"] + 4.0 * gm + 40.0 * st["probes"]                                  # emit side
               + 4.0 * (st["packets_sent"] + st["piggybacks"]) + 4.0 * msgs                                  # delivery side
               + 8.0 * msgs + 24.0 * applied + 16.0 * st["msgs_piggybacked"])                                # merge side: every rumour's view access is somebody's
    assert abs(total - unsplit) < 1e-6 * unsplit
    # a filtered rumour is charged where it is dropped, never to k_resolve
    assert algorithmic_bytes("k_resolve", st) == 8.0 * (msgs - st["msgs_filtered"]) + 24.0 * applied + 16.0 * st["msgs_piggybacked"]
    s.close()


def test_roofline_traffic_quotes_the_committed_pmc_passes_only_for_their_workload(oracle):
    """`roofline.traffic` is the HBM bytes per launch from the rocprofv3 --pmc passes committed under profiles/ (they cannot be
    collected inside the bench process) — quoted when the bench runs the window they were taken over, null otherwise."""
    from bench import roofline_of
    s = Sim(oracle, preset(oracle, abi.PRESET_LAN, n_nodes=2048, seed=5, subject_cap=4))
    s.step(20); s.kill(0, [77]); s0 = s.stats(); s.step(100); st = diff_stats(s0, s.stats())
    s.close()
    prof = {"k_begin": (40, 3.1), "k_deliver": (40, 1.7), "k_resolve": (40, 3.2), "k_census": (40, 0.2), "k_finish": (40, 0.3)}
    r = roofline_of(prof, st, 0.0075, virtual_nodes=4194304)
    assert r["kernel"] == "k_resolve" and r["traffic"] > r["traffic_as_counted"] > 5e7
    assert r["traffic_over_algorithmic"]["as_counted"] > 1.0
    assert roofline_of(prof, st, 0.0075, virtual_nodes=2048)["traffic"] is None
