# Round 6, call 11: rocprofv3 evidence for the implied queue's kernels — kernel trace + stats, then HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes)
# of config #4's shape at 262 144 nodes, unbounded queue, the first 30 simulated seconds; bench.py --gpus 2 with both ranks on the one device (gloo control group)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06k; mkdir -p $O
CMD="python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 30 --every 30 --inbox-cap 16384"
( timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $CMD ) > $O/trace.out 2> $O/trace.err; tail -2 $O/trace.out
f=$(ls $O/trace/*/*_kernel_stats.csv | head -1); cp "$f" $O/config4_262k_kernel_stats.csv; head -14 $O/config4_262k_kernel_stats.csv | cut -c1-160
rm -rf $O/trace
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc/pass$i; i=$((i+1))
  ( timeout 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $CMD ) > $d.out 2> $d.err
done
python tools/pmc_traffic.py $O/pmc 262144 > $O/pmc_config4_262k.json 2> $O/pmc.err; cat $O/pmc_config4_262k.json | head -60
rm -rf $O/pmc/pass*/
( time SWIMSIM_BENCH_C4S_NODES=65536 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --no-config4 --no-config4-partition --no-config5 ) > $O/bench_2ranks_one_device.json 2> $O/bench_2ranks.err; tail -3 $O/bench_2ranks.err | cut -c1-300; head -c 400 $O/bench_2ranks_one_device.json
