set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O/fullcmd
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/fullcmd/bench.json 2> $O/fullcmd/bench.err
cp $(find $O/fullcmd/trace -name "*kernel_stats.csv" | head -1) $O/driver_fullcmd_kernel_stats.csv; rm -rf $O/fullcmd/trace
tail -2 $O/bench_driver.err; wc -c $O/bench_driver.json $O/bench_default.json $O/fullcmd/bench.json
