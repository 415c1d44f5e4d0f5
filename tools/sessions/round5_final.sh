# Round 5, final session: the GPU suite, smoke, the driver window's kernel trace and HBM-traffic passes, the bench line with the driver's arguments and the default window
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05z; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu_final.log 2>&1; tail -14 $O/pytest_gpu_final.log
( timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_final.log 2>&1; tail -2 $O/smoke_final.log
SKIP=0 bash tools/trace_pass.sh $O/trace --steps 20 --warmup 5 > $O/trace.log 2>&1; tail -12 $O/trace.log
PMC_TIMEOUT=240 bash tools/pmc_traffic_pass.sh $O/pmc --steps 20 --warmup 5 > $O/pmc.log 2>&1; tail -5 $O/pmc.log
[ -s $O/pmc/pmc_traffic.json ] && cp $O/pmc/pmc_traffic.json profiles/r05_pmc_driver.json
( time timeout 600 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 600 $O/bench_driver.json
( time timeout 300 python bench.py --no-config4 --no-config5 --no-cpu-baseline --no-convergence ) > $O/bench_default.json 2> $O/bench_default.err; head -c 400 $O/bench_default.json
