# Round 6, call 28: k_piggy_iq's grid — 768 workgroups (what is resident at once) against 2 048 and 1 536; product build, first 60 simulated seconds of config #4's shape at 524 288
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07b; mkdir -p $O
for g in 2048 768 1024 512; do
  ( SWIMSIM_PIGGY_GRID=$g timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/piggy_grid_$g.log 2>&1; echo "== $g"; grep "k_piggy" $O/piggy_grid_$g.log | tail -1
done
