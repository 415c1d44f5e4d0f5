# Round 6, call 1: the implied queue (SWIM_F_UNBOUNDED_QUEUE) on the device for the first time — parity tests, memory, config #4's shape with nothing pruned
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06a; mkdir -p $O
python -c "import torch; f,t=torch.cuda.mem_get_info(); print('hbm free/total bytes', f, t, torch.cuda.get_device_name(0))" > $O/mem.txt 2>&1; cat $O/mem.txt
( time timeout 900 python -m pytest tests/test_unbounded_queue_gpu.py -m gpu -x -q --durations=10 ) > $O/pytest_uq.log 2>&1; tail -15 $O/pytest_uq.log
( time timeout 300 python tools/config4_run.py --nodes 65536 --unbounded --queue-cap 8 --seconds 300 --every 10 --inbox-cap 8192 --profile ) > $O/config4_65k_unbounded.log 2>&1; tail -12 $O/config4_65k_unbounded.log
( time timeout 400 python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 120 --every 10 --inbox-cap 16384 --profile ) > $O/config4_262k_unbounded.log 2>&1; tail -16 $O/config4_262k_unbounded.log
( time timeout 600 python -m pytest tests/test_mass_gpu.py tests/test_parity_gpu.py -m gpu -x -q ) > $O/pytest_regress.log 2>&1; tail -5 $O/pytest_regress.log
