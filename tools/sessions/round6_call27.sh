# Round 6, call 27: + the order after a packet restored by a MERGE of the bumped run with the rest (binary searches in LDS), not a sort
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07a; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unbounded_queue_gpu.py tests/test_scale_gpu.py -m gpu -x -q -k "unbounded or sharded_in or reaper or bridge or mass_failure_to or churn or randomised" ) > $O/pytest_uq.log 2>&1; grep "passed\|failed" $O/pytest_uq.log
( SWIMSIM_LIB=$PWD/_diag/lib_v9.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_v9.log 2>&1; grep "iq clk\|k_gossip" $O/iqclk_v8.log
( time timeout 900 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 900 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_524k_full.log
( time timeout 300 python tools/fuzz_parity.py --unbounded --cases 200 --seed 606 ) > $O/fuzz_unbounded_606.log 2>&1; tail -4 $O/fuzz_unbounded_606.log
