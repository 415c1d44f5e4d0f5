cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03g
timeout 200 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/r03g/c4_262k_bigsort.log 2>&1
tail -3 gpurun_out/r03g/c4_262k_bigsort.log
( time timeout 500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03g/pytest_gpu.log 2>&1; tail -6 gpurun_out/r03g/pytest_gpu.log
