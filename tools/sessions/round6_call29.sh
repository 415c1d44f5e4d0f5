# Round 6, call 29: the implied queue's kernels after the selection — rocprofv3 kernel stats and HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of
# config #4's shape at 262 144 nodes, the first 30 simulated seconds (the numbers bench.py quotes in config4.implied_queue_kernels)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07c; mkdir -p $O/pmc
CMD="python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 30 --every 30 --inbox-cap 16384"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD ) > $O/trace.out 2> $O/trace.err; tail -1 $O/trace.out
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/config4_262k_kernel_stats.csv; rm -rf $O/trace; head -8 $O/config4_262k_kernel_stats.csv
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc/pass$i; i=$((i+1))
  ( timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $CMD ) > $d.out 2> $d.err; tail -1 $d.out
done
python tools/pmc_traffic.py $O/pmc 262144 > $O/pmc_config4_262k.json 2> $O/pmc.err; cat $O/pmc_config4_262k.json | head -70; cat $O/pmc.err | tail -3
rm -rf $O/pmc/pass*/
