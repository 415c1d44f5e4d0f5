# Round 6, call 12: HBM traffic of the implied queue's kernels (FETCH_SIZE / WRITE_SIZE in separate --pmc passes; call 11 forgot the directory), the smoke with its
# unbounded-queue case, the sharded implied queue over the library's mailboxes
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06l; mkdir -p $O/pmc
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 400 python -m pytest tests/test_unbounded_queue_gpu.py tests/test_scale_gpu.py -m gpu -x -q -k "unbounded or sharded or many_replicas or refused or checkpoint or single or mass_failure_to" ) > $O/pytest_uq.log 2>&1; tail -3 $O/pytest_uq.log
CMD="python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 30 --every 30 --inbox-cap 16384"
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=$O/pmc/pass$i; i=$((i+1))
  ( timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $CMD ) > $d.out 2> $d.err; tail -1 $d.out
done
python tools/pmc_traffic.py $O/pmc 262144 > $O/pmc_config4_262k.json 2> $O/pmc.err; cat $O/pmc_config4_262k.json | head -70; cat $O/pmc.err | tail -3
rm -rf $O/pmc/pass*/
