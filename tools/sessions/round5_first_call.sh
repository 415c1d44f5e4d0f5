# Round 5, FIRST GPU call (after tools/sessions/round5_prepare.sh): A/B of the four builds on the driver window (one handle, HIP-event
# roofline pass), then the GPU suite on each variant that is faster than the reference (SWIMSIM_LIB selects the library under test)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0ref.so _ab/lib_spec.so _ab/lib_line1_w5.so _ab/lib_spec_line1_w5.so 2>&1 | tee $O/ab.txt
for v in spec spec_line1_w5; do
  ( time SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_$v.log 2>&1; tail -4 $O/pytest_$v.log
done
