# one GPU-box session of round 2: the GPU suite, the bench at the driver's and at the default arguments, kernel traces, PMC passes
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -40 > gpurun_out/t4.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_bench_driver.json 2> gpurun_out/r02_bench_driver.err
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
# the kernel-stats summary of EXACTLY the driver's command (all legs of the bench line: main run, roofline pass, detection, config4, convergence)
mkdir -p gpurun_out/r02_driver_fullcmd
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_driver_fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_driver_fullcmd/bench.json 2> gpurun_out/r02_driver_fullcmd/bench.err
cp $(find gpurun_out/r02_driver_fullcmd/trace -name "*kernel_stats.csv" | head -1) gpurun_out/r02_driver_fullcmd/kernel_stats.csv; rm -rf gpurun_out/r02_driver_fullcmd/trace
# ...and of its timed region alone (--main-only), with the per-tick breakdown
SKIP=10 bash tools/trace_pass.sh gpurun_out/r02_driver_trace --steps 20 --warmup 5 > gpurun_out/r02_driver_trace.log 2>&1
bash tools/trace_pass.sh gpurun_out/r02_default_trace > gpurun_out/r02_default_trace.log 2>&1
# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE, one pass each) and the SQ counters of the heaviest launches
bash tools/pmc_traffic_pass.sh gpurun_out/r02_pmc_driver --steps 20 --warmup 5 > gpurun_out/r02_pmc_driver.log 2>&1
bash tools/pmc_pass.sh gpurun_out/r02_pmc_sq --steps 20 --warmup 5 > gpurun_out/r02_pmc_sq.log 2>&1
python tools/pmc_report.py gpurun_out/r02_pmc_sq 8 > gpurun_out/r02_pmc_heavy_ticks.txt 2>&1; rm -rf gpurun_out/r02_pmc_sq
cat gpurun_out/t4.log
