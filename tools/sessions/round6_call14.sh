# Round 6, call 14: the GPU suite with the 16 384 / 819 unbounded fixture and the mailbox variant of the sharded implied-queue test
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06n; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu.log 2>&1; tail -10 $O/pytest_gpu.log
