# round 4, GPU call 4: the whole GPU suite on the current tree (tile buckets off by default + their own tests), config #4's leg at 524 288 nodes
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 524288 --seconds 1600 --every 100 ) > $O/config4_524k.log 2>&1; tail -4 $O/config4_524k.log
