# Round 6, call 35: SQ counters of the headline window's kernels (driver arguments, one handle): issue bound or waiting?
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07h; mkdir -p $O
PMC_TIMEOUT=240 PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM;SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_IFETCH" bash tools/pmc_pass.sh $O/pmc --steps 20 --warmup 5 2>&1 | tail -6
python tools/pmc_report.py $O/pmc 8 > $O/pmc_headline.txt 2>&1; cat $O/pmc_headline.txt | head -80
rm -rf $O/pmc/pass*/
