# Round 5, call 6: the one-wave k_resolve — tile order (longest job first / plain) and one or two node blocks per wave
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05f; mkdir -p $O
timeout 700 bash tools/ab_kernels.sh _ab/lib_wave.so _ab/lib_wave_plain.so _ab/lib_wave_rt2.so _ab/lib_wave_rt2_plain.so 2>&1 | tee $O/ab.txt
