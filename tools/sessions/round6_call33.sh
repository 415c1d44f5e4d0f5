# Round 6, call 33: message lengths and their ranks with the unbounded queue (three ranks / one rank / suspect = dead longer than alive / dead in between): new GPU test
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07f; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unbounded_queue_gpu.py -m gpu -q -k "message_lengths" ) > $O/pytest_lens.log 2>&1; tail -30 $O/pytest_lens.log
