# one GPU-box session of round 2: the GPU suite, the bench at the driver's and at the default arguments, kernel traces
set -x
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > gpurun_out/t4.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
SKIP=10 bash tools/trace_pass.sh gpurun_out/r02_driver_trace --steps 20 --warmup 5 > gpurun_out/r02_driver_trace.log 2>&1
bash tools/trace_pass.sh gpurun_out/r02_default_trace > gpurun_out/r02_default_trace.log 2>&1
cat gpurun_out/t4.log
