# Round 5, call 2: which unit k_resolve / k_begin / k_deliver wait on — TA / TCP / TCC stall counters on the driver window (one handle) —
# and two request-trimming variants of k_resolve (only the words of the inbox line in use; view metadata on demand)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -c . $O/counters_list.txt
bash tools/ab_kernels.sh _ab/lib_0ref.so _ab/lib_linecond.so _ab/lib_lazyvm.so _ab/lib_lc_lv.so 2>&1 | grep -v "default window" | tee $O/ab.txt
export SWIMSIM_LIB=$PWD/_ab/lib_0ref.so
PMC_GROUPS="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE;TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE;TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum GRBM_GUI_ACTIVE;TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum;TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_REQ_sum TCC_MISS_sum;SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY;SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM" \
  bash tools/pmc_pass.sh $O/pmc --steps 20 --warmup 5 2>&1 | tail -40
python tools/pmc_report.py $O/pmc 8 > $O/pmc_heavy.txt 2>&1; cat $O/pmc_heavy.txt
rm -rf $O/pmc/pass*/*/*_agent_info.csv; du -sh $O
