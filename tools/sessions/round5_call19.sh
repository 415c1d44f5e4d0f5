# Round 5, call 19 (the last GPU minutes): QueueBroadcast's two scans of the queue (invalidation, then Prune) as ONE pass — config #4's leg at 262 144 nodes — then the GPU suite on that build
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05s; mkdir -p $O
for v in twopass onepass; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 120 python tools/config4_run.py --nodes 262144 --seconds 200 2>&1 | tail -2 | sed "s/^/$v: /" | cut -c1-200 | tee -a $O/ab_c4.txt
done
( time timeout 430 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
