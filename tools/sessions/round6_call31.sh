# Round 6, call 31: k_piggy_iq without its per-node __threadfence() (a workgroup-scope fence, only before a further batch of the same node): parity, fuzz, per-kernel times, full leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07e; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unbounded_queue_gpu.py tests/test_scale_gpu.py -m gpu -x -q -k "unbounded or sharded_in or reaper or bridge or mass_failure_to or churn or randomised" ) > $O/pytest_uq.log 2>&1; grep "passed\|failed" $O/pytest_uq.log
( time timeout 300 python tools/fuzz_parity.py --unbounded --cases 200 --seed 606 ) > $O/fuzz_unbounded_606.log 2>&1; tail -4 $O/fuzz_unbounded_606.log
( timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/first60.log 2>&1; grep "k_piggy" $O/first60.log | tail -1
( time timeout 900 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 900 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_524k_full.log
