# round 4, GPU call 7: k_resolve compiled without serf's handlers for handles without an event layer: headline A/B (reference = the build before the
# intent work), the intent scripts, the suite, config #4's leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0ref.so consul_amd/libswimsim.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 ) > $O/config4_262k.log 2>&1; tail -3 $O/config4_262k.log | cut -c1-200
