#!/bin/bash
# usage: tools/pmc_pass.sh <outdir> <bench args...> — one rocprofv3 --pmc pass per counter group
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM"; do
  d="$out/pass$i"; i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- python bench.py --main-only "$@" > "$d.out" 2> "$d.err"
done
find "$out" -name "*.csv" | head -20; du -sh "$out"
