# Round 5, call 11: the GPU suite on the round's build (one-wave k_resolve, k_census_finish, frames sized from the load, k_inbox_sort_huge,
# reconnect scan, revive by tile marks) + the recovery run to its end at 16 384 nodes; the clusters on 2 / 3 / 4 handles
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05k; mkdir -p $O
( time timeout 200 python -m pytest tests/test_scale_gpu.py::test_partition_recovery_converges_16384 -m gpu -x -q --durations=3 ) > $O/pytest_converge.log 2>&1; tail -12 $O/pytest_converge.log
( time timeout 700 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -16 $O/pytest_gpu.log
for h in 2 3 4; do
  timeout 120 python bench.py --handles $h --steps 20 --warmup 5 --main-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('handles $h driver window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/handles.txt
done
