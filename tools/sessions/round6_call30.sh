# Round 6, call 30: k_piggy_iq takes 64 consecutive nodes per workgroup (their columns' runs are neighbours) instead of the next entries of k_deliver's list: parity, per-kernel times, full leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07d; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unbounded_queue_gpu.py tests/test_scale_gpu.py -m gpu -x -q -k "unbounded or sharded_in or reaper or bridge or mass_failure_to or churn or randomised" ) > $O/pytest_uq.log 2>&1; grep "passed\|failed" $O/pytest_uq.log
for g in 768 2048; do
  ( SWIMSIM_PIGGY_GRID=$g timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/piggy_grid_$g.log 2>&1; echo "== $g"; grep "k_piggy" $O/piggy_grid_$g.log | tail -1
done
( time timeout 900 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 900 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_524k_full.log
