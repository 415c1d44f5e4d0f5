# Round 6, call 2: the GPU suite on the build with the implied queue and the pooled inbox rows; config #4's shape with nothing pruned at 65 536 / 262 144 / 524 288
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06b; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -14 $O/pytest_gpu.log
( time timeout 200 python tools/config4_run.py --nodes 65536 --unbounded --queue-cap 8 --seconds 300 --every 20 --inbox-cap 8192 --profile ) > $O/config4_65k_unbounded.log 2>&1; tail -4 $O/config4_65k_unbounded.log
( time timeout 400 python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 400 --every 20 --inbox-cap 16384 --profile ) > $O/config4_262k_unbounded.log 2>&1; tail -6 $O/config4_262k_unbounded.log
( time timeout 900 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 600 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_unbounded.log 2>&1; tail -12 $O/config4_524k_unbounded.log
