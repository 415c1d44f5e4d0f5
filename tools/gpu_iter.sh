# one quick GPU iteration: the parity tests (without the multi-process exchange ones), then a kernel trace at the driver's arguments
set -x
timeout 900 python -m pytest tests -m gpu -q -x -k "not library_exchange and not bench_two" 2>&1 | tail -15 > gpurun_out/iter_tests.log
SKIP=10 bash tools/trace_pass.sh gpurun_out/iter_trace --steps 20 --warmup 5 > gpurun_out/iter_trace.log 2>&1
cat gpurun_out/iter_tests.log; tail -12 gpurun_out/iter_trace.log
