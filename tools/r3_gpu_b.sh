cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mass_gpu.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden --deselect tests/test_scale_gpu.py::test_churn_and_event_flood_8192_matches_golden --ignore tests/test_mass_gpu.py 2>&1 | tail -15
