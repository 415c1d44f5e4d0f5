# Round 5, evidence after the final build: k_resolve's phase clock and k_begin's role clock with one wave per node block; SQ / L2 counters of the driver window
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05y; mkdir -p $O
SWIMSIM_LIB=$PWD/_ab/lib_diag.so SWIMSIM_RESOLVECLK=1 SWIMSIM_ROLECLK=$PWD/$O/roleclk.txt timeout 240 python bench.py --main-only --handles 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/diag.err
grep -a "resolve clk" $O/diag.err | tail -10 | tee $O/resolve_clk.txt
PMC_TIMEOUT=200 PMC_GROUPS="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM;TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" bash tools/pmc_pass.sh $O/pmc --steps 20 --warmup 5 2>&1 | tail -8
python tools/pmc_report.py $O/pmc 8 > $O/pmc_heavy.txt 2>&1; cat $O/pmc_heavy.txt | head -60
rm -rf $O/pmc/pass*/
