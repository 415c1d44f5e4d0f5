# Round 6, call 18: k_piggy_iq at four waves per SIMD (128 VGPRs + scratch) against three (164, no scratch): config #4's shape at 524 288, the first 60 s
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06r; mkdir -p $O
( timeout 400 python tools/config4_run.py --nodes 524288 --unbounded --queue-cap 8 --seconds 60 --every 20 --inbox-cap 32768 --profile ) > $O/config4_524k_60s.log 2>&1; grep "k_gossip\|t_s\": 61" $O/config4_524k_60s.log | cut -c1-260
