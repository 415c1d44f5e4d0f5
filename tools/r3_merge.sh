cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03h
timeout 200 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/r03h/c4_262k_pipeline.log 2>&1
tail -3 gpurun_out/r03h/c4_262k_pipeline.log
( time timeout 400 python -m pytest tests/test_mass_gpu.py tests/test_reconnect.py tests/test_scale_gpu.py tests/test_membership.py -m gpu -x -q ) > gpurun_out/r03h/pytest_gpu.log 2>&1; tail -6 gpurun_out/r03h/pytest_gpu.log
