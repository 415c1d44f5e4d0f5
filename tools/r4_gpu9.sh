# round 4, GPU call 9: k_resolve<MASS, SERF, DYN> against k_resolve<MASS, SERF>; the suite on it
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_serf.so _ab/lib_dyn.so > $O/ab_dyn.txt 2>&1; cat $O/ab_dyn.txt
cp _ab/lib_dyn.so consul_amd/libswimsim.so
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
