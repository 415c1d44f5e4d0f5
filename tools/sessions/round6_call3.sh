# Round 6, call 3: the implied-queue kernels after the latency work (16 row blocks in flight, batched emission): config #4's shape at 524 288 for 200 s, and to full detection at 262 144
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06c; mkdir -p $O
( time timeout 300 python -m pytest tests/test_unbounded_queue_gpu.py -m gpu -x -q ) > $O/pytest_uq.log 2>&1; tail -3 $O/pytest_uq.log
( time timeout 600 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 200 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_200s.log 2>&1; tail -4 $O/config4_524k_200s.log
( time timeout 600 python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 700 --every 20 --inbox-cap 16384 --profile ) > $O/config4_262k_full.log 2>&1; tail -4 $O/config4_262k_full.log
