set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f
mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 700 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | tail -4
( time python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
wc -c $O/*.json
