"""Tile buckets (SWIMSIM_TILEBUCKETS=1): rumours, piggy-back orders and carried broadcasts travel through per-tile buckets
and the no-op filter runs at the receivers (k_deliver's per-tile drain) instead of at the senders.  The switch changes WHERE the same
question is asked, so every result must stay what it is without it: the parity cases below are the ordinary ones, re-run with the
switch on (off by default: measured slower, profiles/r04_ab_experiments.txt)."""
import numpy as np
import pytest

from consul_amd import abi
from consul_amd.sim import Sim, preset

import test_parity_gpu as tp

pytestmark = pytest.mark.gpu


@pytest.fixture
def tile_buckets(monkeypatch):
    monkeypatch.setenv("SWIMSIM_TILEBUCKETS", "1")


def test_the_switch_is_honoured(hip, tile_buckets, monkeypatch):
    a = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=4096, seed=3))
    assert a.info(abi.INFO_TILE_BUCKETS) == 1
    a.close()
    b = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=4096, seed=3, mass_rows=16, view_cap=8))     # the dense pair store keeps the sender-side filter
    assert b.info(abi.INFO_TILE_BUCKETS) == 0
    b.close()
    monkeypatch.delenv("SWIMSIM_TILEBUCKETS")
    c = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=4096, seed=3))
    assert c.info(abi.INFO_TILE_BUCKETS) == 0
    c.close()


def test_lockstep_small_edge_lists(hip, oracle, tile_buckets):
    tp.test_single_failure_lockstep_small(hip, oracle)


@pytest.mark.parametrize("n,reps,seed", [(4096, 3, 11), (65536, 2, 5)])
def test_single_failure_replicas(hip, oracle, tile_buckets, n, reps, seed):
    tp.test_single_failure_replicas(hip, oracle, n, reps, seed)


def test_many_subjects_under_loss_and_a_partition(hip, oracle, tile_buckets):
    tp.test_loss_refute_and_partition(hip, oracle)


def test_push_pull_and_heal(hip, oracle, tile_buckets):
    tp.test_push_pull_parity_and_heal(hip, oracle)


def test_user_events_and_churn(hip, oracle, tile_buckets):
    import test_serf_events as ts
    ts.test_user_events_hip_matches_oracle(hip, oracle)
    tp.test_churn_and_event_flood(hip, oracle, 5)


def test_transport_bridge(hip, oracle, tile_buckets):
    import test_transport_bridge as tb
    tb.test_bridge_hip_matches_oracle(hip, oracle)


@pytest.mark.parametrize("n_shards", [2, 4])
def test_shards_in_one_process(hip, oracle, tile_buckets, n_shards):
    tp.test_sharded_population_matches_unsharded(hip, oracle, n_shards)
    if n_shards == 2:
        tp.test_churn_and_event_flood_sharded(hip, oracle)


def test_randomised_cases(hip, oracle, tile_buckets):
    tp.test_randomised_parity_cases(hip, oracle)
