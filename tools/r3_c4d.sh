cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mass_gpu.py tests/test_reconnect.py "tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden" -x -q 2>&1 | tail -4
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/c4_262k_v3.log 2>&1
tail -2 gpurun_out/c4_262k_v3.log
