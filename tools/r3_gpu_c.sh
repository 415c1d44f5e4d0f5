cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden 2>&1 | tail -15
