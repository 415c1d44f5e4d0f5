# Round 6, call 15: the round's last build (the reaper over rows, the reaper after the fold's census, attach clearing the implied queue): GPU suite, smoke, the bench line
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06o; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu.log 2>&1; tail -10 $O/pytest_gpu.log
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 300 $O/bench_driver.json
