# Round 5, FIRST GPU call (after tools/sessions/round5_prepare.sh): A/B of the four builds on the driver window (one handle, HIP-event
# roofline pass), then the GPU suite on each variant that is faster than the reference (SWIMSIM_LIB selects the library under test)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0ref.so _ab/lib_spec.so _ab/lib_line1_w5.so _ab/lib_spec_line1_w5.so 2>&1 | tee $O/ab.txt
for v in spec spec_line1_w5; do
  ( time SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 600 python -m pytest tests -m gpu -x -q ) > $O/pytest_$v.log 2>&1; tail -4 $O/pytest_$v.log
done
# ... and the dense-store variant on what it is for: config #4's leg at 262 144 nodes (55 s on the reference build), then its own GPU tests
for v in 0ref hbmq; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 400 python tools/config4_run.py --nodes 262144 2>&1 | tail -3 | tee -a $O/ab_c4.txt
done
( time SWIMSIM_LIB=$PWD/_ab/lib_hbmq.so timeout 600 python -m pytest tests/test_mass_gpu.py tests/test_scale_gpu.py tests/test_serf_intents_gpu.py -m gpu -x -q ) > $O/pytest_hbmq.log 2>&1; tail -4 $O/pytest_hbmq.log
