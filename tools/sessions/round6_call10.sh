# Round 6, call 10: the GPU suite on the build with the sharded implied queue; the bench line with the driver's arguments; bench.py --gpus 2 with both ranks on the one device
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06j; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu.log 2>&1; tail -10 $O/pytest_gpu.log
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver.json 2> $O/bench_driver.err; tail -3 $O/bench_driver.err; head -c 300 $O/bench_driver.json
( time timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --no-config4 --no-config4-partition --no-config5 ) > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; tail -3 $O/bench_2ranks.err; head -c 300 $O/bench_2ranks_one_device.json
