#!/bin/bash
# usage: tools/pmc_groups.sh <outdir> "<counters of pass 0>" "<counters of pass 1>" ... — one rocprofv3 --pmc pass per group
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for c in "$@"; do
  d="$out/pass$i"; i=$((i+1))
  timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- python bench.py --main-only --handles 1 > "$d.out" 2> "$d.err"
  tail -2 "$d.err"
done
