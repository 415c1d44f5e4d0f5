cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_mass_gpu.py -x -q 2>&1 | tail -5
bash tools/r3_ab.sh
