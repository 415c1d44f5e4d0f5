"""The scenario of the device-driven exchange tests (shared by the worker processes and the reference run)."""
from consul_amd import abi

KW = dict(n_nodes=8192, n_replicas=2, seed=5, subject_cap=8, view_cap=64, queue_cap=16, inbox_cap=1024,
          loss_q32=int(0.05 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK, fold_interval_ms=5000, push_pull_interval_ms=3000)


def run(s):
    s.step_ms(3000)
    s.kill(0, [100, 6000]); s.kill(1, [7]); s.update(1, [4096])
    s.step_ms(30000)
    s.revive(0, [100])
    s.step_ms(45000)
