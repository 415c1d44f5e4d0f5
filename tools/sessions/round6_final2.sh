# Round 6, last call: the GPU suite and the smoke on the final tree (six tests more than in round6_final.sh)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06final2; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > $O/pytest_gpu_final.log 2>&1; tail -14 $O/pytest_gpu_final.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_final.log 2>&1; tail -2 $O/smoke_final.log
