# Round 5, call 5: k_resolve as one wave per node block (64 threads, no barriers that wait for anybody) against the 256-thread / four-block tile
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05e; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_state_table.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_quick.txt
timeout 500 bash tools/ab_kernels.sh _ab/lib_oldgeo.so _ab/lib_wave.so 2>&1 | tee $O/ab.txt
for v in oldgeo wave; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 300 python tools/config4_run.py --nodes 262144 2>&1 | tail -2 | tee -a $O/ab_c4.txt
done
