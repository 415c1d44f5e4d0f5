# Round 6, call 36: config #4's shape at 65 536 and 262 144 to full detection on the final kernels (DESIGN 7.0's "smaller" row)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07i; mkdir -p $O
( time timeout 300 python tools/config4_run.py --nodes 65536 --unbounded --queue-cap 8 --seconds 400 --every 20 --inbox-cap 8192 --profile ) > $O/config4_65k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_65k_full.log
( time timeout 600 python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 900 --every 20 --inbox-cap 16384 --profile ) > $O/config4_262k_full.log 2>&1; grep "^{'k_\|full detection" $O/config4_262k_full.log
