# round 4, GPU call 5: the GPU suite with the receiving-shard filter, where config #4's mass phase spends its time (per-kernel HIP events)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 120 --every 20 --profile ) > $O/config4_262k_profile.log 2>&1; tail -4 $O/config4_262k_profile.log
