# round 4, the final GPU-box session: smoke, the GPU suite, the bench at the driver's arguments and at the defaults, the kernel trace and
# tick breakdown of the timed region, HBM traffic (FETCH_SIZE / WRITE_SIZE passes), L2 / SQ counters, the kernel table of the driver's exact command
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | tail -4
( time python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
wc -c $O/*.json
SKIP=10 bash tools/trace_pass.sh $O/driver_trace --steps 20 --warmup 5 > $O/driver_trace.log 2>&1; tail -12 $O/driver_trace.log
bash tools/pmc_traffic_pass.sh $O/pmc_driver --steps 20 --warmup 5 > $O/pmc_driver.log 2>&1; tail -3 $O/pmc_driver.log | cut -c1-300
PMC_GROUPS="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/pmc_pass.sh $O/pmc_l2 --steps 20 --warmup 5 > $O/pmc_l2.log 2>&1
python tools/pmc_report.py $O/pmc_l2 8 > $O/pmc_l2_heavy_ticks.txt 2>&1
python tools/pmc_report.py $O/pmc_l2 400 > $O/pmc_l2_all_ticks.txt 2>&1
rm -rf $O/pmc_l2
mkdir -p $O/fullcmd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/fullcmd/bench.json 2> $O/fullcmd/bench.err
cp $(find $O/fullcmd/trace -name "*kernel_stats.csv" | head -1) $O/driver_fullcmd_kernel_stats.csv; rm -rf $O/fullcmd/trace
ls -la $O
