# Round 6, call 25: SQ counters of the implied queue's kernels (call 24's report left them out): instructions issued against wave cycles — issue bound or not?
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06y; mkdir -p $O/pmc
CMD="python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 12 --every 12 --inbox-cap 16384"
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH"; do
  d=$O/pmc/pass$i; i=$((i+1))
  ( timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- $CMD ) > $d.out 2> $d.err; tail -1 $d.out; tail -2 $d.err
done
python tools/pmc_report.py $O/pmc 8 > $O/pmc_iq.txt 2>&1; grep -A22 "k_gossip_iq\|k_piggy_iq" $O/pmc_iq.txt
rm -rf $O/pmc/pass*/
