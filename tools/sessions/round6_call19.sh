# Round 6, call 19: randomised parity with the unbounded queue (tools/fuzz_parity.py --unbounded): 200 cases of seed 606, 100 of seed 17
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06s; mkdir -p $O
( time timeout 1200 python tools/fuzz_parity.py --unbounded --cases 200 --seed 606 ) > $O/fuzz_unbounded_606.log 2>&1; tail -4 $O/fuzz_unbounded_606.log
( time timeout 900 python tools/fuzz_parity.py --unbounded --cases 100 --seed 17 ) > $O/fuzz_unbounded_17.log 2>&1; tail -4 $O/fuzz_unbounded_17.log
