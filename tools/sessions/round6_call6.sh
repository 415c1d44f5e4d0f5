# Round 6, call 6: the phase clock of k_gossip_iq, tallied per workgroup (v2), with the wait for the scan's loads split out; 524 288 nodes, heavy phase
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06f; mkdir -p $O
( SWIMSIM_LIB=$PWD/_diag/libswimsim_diag.so SWIMSIM_IQCLK=1 timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/iqclk_524k.log 2>&1; grep "iq clk\|k_gossip" $O/iqclk_524k.log
