cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/config4_run.py --nodes 65536 --seconds 2000 --every 50 > gpurun_out/c4_65k_full.log 2>&1
timeout 420 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 50 > gpurun_out/c4_262k_full.log 2>&1
tail -3 gpurun_out/c4_65k_full.log; tail -3 gpurun_out/c4_262k_full.log
