#!/bin/bash
# usage: tools/pmc_traffic_pass.sh <outdir> <bench args...> — HBM traffic per kernel launch: one rocprofv3 --pmc pass for FETCH_SIZE,
# one for WRITE_SIZE (they do not fit one pass, MI355X_MICROARCH.md), of `bench.py --main-only <args>`; summary by tools/pmc_traffic.py
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  d="$out/pass$i"; i=$((i+1))
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$d" -- python bench.py --main-only --handles 1 "$@" > "$d.out" 2> "$d.err"
done
python tools/pmc_traffic.py "$out" 4194304 > "$out/pmc_traffic.json" 2> "$out/pmc_traffic.err"
cat "$out/pmc_traffic.json"
rm -rf "$out"/pass*/
