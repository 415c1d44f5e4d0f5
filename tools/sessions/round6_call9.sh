# Round 6, call 9: tier pre-filter in the scan, pair-index bitonic: clock + times at 524 288 (heavy phase, then to full detection); the bench line
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06i; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unbounded_queue_gpu.py -m gpu -x -q ) > $O/pytest_uq.log 2>&1; grep passed $O/pytest_uq.log
( SWIMSIM_LIB=$PWD/_diag/libswimsim_diag.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_524k.log 2>&1; grep "iq clk\|k_gossip" $O/iqclk_524k.log
( timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 100 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_100s.log 2>&1; grep "k_gossip\|t_s\": 101" $O/config4_524k_100s.log | cut -c1-250
( time timeout 900 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 900 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_524k_full.log
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 600 $O/bench_driver.json
