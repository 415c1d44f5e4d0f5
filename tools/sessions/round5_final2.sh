# Round 5, last session: the GPU suite and the bench line on the round's last build (a 4-tick graph tier and the pair store's layout came after round5_final.sh)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05zz; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=5 ) > $O/pytest_gpu_final.log 2>&1; tail -12 $O/pytest_gpu_final.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_final.log 2>&1; tail -2 $O/smoke_final.log
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 400 $O/bench_driver.json
