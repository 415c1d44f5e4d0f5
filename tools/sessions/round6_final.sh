# Round 6, final session: the GPU suite, smoke, the driver window's kernel trace and HBM-traffic passes, the bench line with the driver's arguments
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu_final.log 2>&1; tail -14 $O/pytest_gpu_final.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_final.log 2>&1; tail -2 $O/smoke_final.log
SKIP=0 bash tools/trace_pass.sh $O/trace --steps 20 --warmup 5 > $O/trace.log 2>&1; tail -12 $O/trace.log
PMC_TIMEOUT=240 bash tools/pmc_traffic_pass.sh $O/pmc --steps 20 --warmup 5 > $O/pmc.log 2>&1; tail -5 $O/pmc.log
[ -s $O/pmc/pmc_traffic.json ] && cp $O/pmc/pmc_traffic.json profiles/r06_pmc_driver.json
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 600 $O/bench_driver.json
