cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 800 python tools/config4_run.py --nodes 524288 --seconds 2500 --every 50 > gpurun_out/c4_524k_full.log 2>&1
tail -2 gpurun_out/c4_524k_full.log
