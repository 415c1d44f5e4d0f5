# round 4, GPU call 6: serf intent ordering on the device, one view lookup per membership rumour; suite, headline check, config-4 leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
( time timeout 600 python -m pytest tests/test_serf_intents_gpu.py -m gpu -x -q ) > $O/pytest_intents.log 2>&1; tail -25 $O/pytest_intents.log
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
bash tools/ab_kernels.sh _ab/lib_0base.so consul_amd/libswimsim.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 --profile ) > $O/config4_262k.log 2>&1; tail -4 $O/config4_262k.log | cut -c1-300
