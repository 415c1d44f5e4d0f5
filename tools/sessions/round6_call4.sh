# Round 6, call 4: the phase clock of k_gossip_iq (a -DSWIMSIM_DIAG build) on config #4's shape at 262 144 and 524 288 nodes, the heavy phase
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06d; mkdir -p $O
( SWIMSIM_LIB=$PWD/_diag/libswimsim_diag.so SWIMSIM_IQCLK=1 timeout 300 python tools/config4_run.py --nodes 262144 --unbounded --queue-cap 8 --seconds 80 --every 20 --inbox-cap 16384 --profile ) > $O/iqclk_262k.log 2>&1; grep "iq clk\|k_gossip" $O/iqclk_262k.log
( SWIMSIM_LIB=$PWD/_diag/libswimsim_diag.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 80 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_524k.log 2>&1; grep "iq clk\|k_gossip" $O/iqclk_524k.log
