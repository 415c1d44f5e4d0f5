# Round 5, call 1 (trimmed round5_first_call.sh): A/B of the prepared k_resolve builds on the driver window, config #4's leg on the HBM-queue build
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0ref.so _ab/lib_spec.so _ab/lib_line1_w5.so _ab/lib_spec_line1_w5.so 2>&1 | tee $O/ab.txt
for v in 0ref hbmq; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 400 python tools/config4_run.py --nodes 262144 2>&1 | tail -3 | tee -a $O/ab_c4.txt
done
