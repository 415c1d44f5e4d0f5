#!/bin/bash
# usage: tools/trace_pass.sh <outdir> <bench args...> — rocprofv3 kernel trace + stats of one bench run, tick breakdown
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
timeout ${TRACE_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python bench.py --main-only --handles 1 "$@" > "$out/bench.json" 2> "$out/bench.err"
python tools/tick_report.py "$out/trace" ${SKIP:-50} > "$out/tick_breakdown.txt" 2>&1
python tools/tick_timeline.py "$out/trace" > "$out/tick_timeline.txt" 2>&1
cp $(find "$out/trace" -name "*kernel_stats.csv" | head -1) "$out/kernel_stats.csv"
rm -rf "$out/trace"
cat "$out/tick_breakdown.txt"; python -c "import json,sys; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'])"
