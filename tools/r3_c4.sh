cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/config4_run.py --nodes 65536 --seconds 400 --every 20 --profile > gpurun_out/c4_65k.log 2>&1
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 150 --every 10 --profile > gpurun_out/c4_262k.log 2>&1
timeout 400 python tools/config4_run.py --nodes 524288 --seconds 60 --every 10 --profile > gpurun_out/c4_524k.log 2>&1
tail -4 gpurun_out/c4_65k.log; tail -3 gpurun_out/c4_262k.log; tail -3 gpurun_out/c4_524k.log
