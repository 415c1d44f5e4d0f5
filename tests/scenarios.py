"""Stimulus scripts shared by the GPU tests and tools/make_golden.py (so that a golden fixture and the test that checks
it cannot drift apart).  Every function drives any object with the Sim interface (Sim, ShardedSim)."""
import numpy as np

from consul_amd import abi

# BASELINE config #4's shape on one GPU: 5 % of the nodes cut off at once (partition mask, both directions)
PARTITION_262K = dict(n_nodes=262144, seed=11, view_cap=64, queue_cap=8, inbox_cap=128, subject_cap=4)
# BASELINE config #5's shape: every second 10 % of the nodes flip alive <-> dead
CHURN_131K = dict(n_nodes=131072, seed=12, view_cap=32, queue_cap=8, inbox_cap=128, subject_cap=4, fold_interval_ms=5000)

STAT_KEYS = ("packets_sent", "msgs_sent", "msgs_applied", "probes", "probe_failures", "suspicion_timeouts", "confirmations",
             "refutes", "queue_drops", "view_drops", "view_evictions", "folds", "fold_freed", "inbox_overflow")


def partition_mask(n, share=0.05, rng_seed=44):
    victims = np.random.default_rng(rng_seed).choice(n, size=int(n * share), replace=False)
    mask = np.zeros(n, dtype=np.uint8)
    mask[victims] = 1
    return mask


def run_partition(sim, n, seconds, checkpoints=()):
    """1 s of quiet, then the partition; returns {second: (digest, stats subset)} at the checkpoints."""
    out = {}
    sim.step_ms(1000)
    sim.partition(0, partition_mask(n))
    for sec in range(1, seconds + 1):
        sim.step_ms(1000)
        if sec in checkpoints:
            sim.sync()
            st = sim.stats()
            out[sec] = (sim.digest(), {k: st[k] for k in STAT_KEYS})
    return out


def run_churn(sim, n, seconds, share=0.10, rng_seed=45, checkpoints=()):
    """Every simulated second a uniformly drawn `share` of the nodes flips alive <-> dead (kill / revive)."""
    rng = np.random.default_rng(rng_seed)
    dead = np.zeros(n, dtype=bool)
    out = {}
    for sec in range(1, seconds + 1):
        flip = rng.choice(n, size=int(n * share), replace=False)
        kill, revive = flip[~dead[flip]], flip[dead[flip]]
        dead[flip] = ~dead[flip]
        if len(kill):
            sim.kill(0, kill.tolist())
        if len(revive):
            sim.revive(0, revive.tolist())
        sim.step_ms(1000)
        if sec in checkpoints:
            sim.sync()
            st = sim.stats()
            out[sec] = (sim.digest(), {k: st[k] for k in STAT_KEYS})
    return out


HEAL_2K = dict(n_nodes=2048, seed=2, view_cap=1024, queue_cap=16, inbox_cap=2048, subject_cap=4, push_pull_interval_ms=2000,
               fold_interval_ms=5000)


def run_partition_heal(sim, n, checkpoints=(46, 76, 136)):
    """5 % cut off for 45 s (both sides start declaring each other dead), then the cut heals and
    push-pull anti-entropy brings everybody back: returns {second: (digest, stats subset)}."""
    mask = partition_mask(n, rng_seed=4)
    out = {}
    sim.step_ms(1000)
    sim.partition(0, mask)
    for sec in range(2, max(checkpoints) + 1):
        if sec == 47:
            sim.partition(0, np.zeros(n, dtype=np.uint8))
        sim.step_ms(1000)
        if sec in checkpoints:
            sim.sync()
            st = sim.stats()
            out[sec] = (sim.digest(), {k: st[k] for k in STAT_KEYS + ("push_pulls",)})
    return out
