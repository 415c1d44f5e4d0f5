#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> <bench args...> — one rocprofv3 --pmc pass per counter group (groups in $PMC_GROUPS, ';'-separated, optional)
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
IFS=';' read -ra groups <<< "${PMC_GROUPS:-FETCH_SIZE;WRITE_SIZE;SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM}"
for c in "${groups[@]}"; do
  d="$out/pass$i"; i=$((i+1))
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- python bench.py --main-only --handles 1 "$@" > "$d.out" 2> "$d.err"
  tail -2 "$d.err"
done
