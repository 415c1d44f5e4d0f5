# Round 6, call 23: A/B of the two changes of call 22 (which together were slower than call 21), with the key built only for a row that enters the pool:
# v1 = stage-per-round-trip sort, type selects; v2 = + suspect==dead fast path; v3 = register sort, type selects; v4 = register sort + fast path
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06w; mkdir -p $O
for v in v1 v2 v3 v4; do
  ( SWIMSIM_LIB=$PWD/_diag/lib_$v.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_$v.log 2>&1; echo "== $v"; grep "iq clk\|k_gossip\|digest" $O/iqclk_$v.log | tail -8
done
