# Round 6, call 24: the compaction out of line (iq_build 43 KB -> 18 KB of code): v5 = register sort, v6 = stage-per-round-trip sort; then SQ counters of the
# implied queue's kernels (instructions issued against wave cycles: is the selection issue bound or fetch bound?)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06x; mkdir -p $O/pmc
for v in v5 v6; do
  ( SWIMSIM_LIB=$PWD/_diag/lib_$v.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_$v.log 2>&1; echo "== $v"; grep "iq clk\|k_gossip\|digest" $O/iqclk_$v.log | tail -8
done
CMD="python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 12 --every 12 --inbox-cap 16384"
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  d=$O/pmc/pass$i; i=$((i+1))
  ( timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $CMD ) > $d.out 2> $d.err; tail -1 $d.out
done
python tools/pmc_report.py $O/pmc 8 > $O/pmc_iq.txt 2>&1; head -60 $O/pmc_iq.txt
rm -rf $O/pmc/pass*/
