# Round 5, call 3: census + epilogue in one launch; k_resolve's receiver list ordered by kind (order-only / rumours)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05c; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_state_table.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_quick.txt
timeout 600 bash tools/ab_kernels.sh _ab/lib_0ref.so _ab/lib_new_splitfin.so _ab/lib_new.so 2>&1 | tee $O/ab.txt
