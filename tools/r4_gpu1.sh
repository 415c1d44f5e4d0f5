# round 4, GPU call 1: A/B of the ISA fixes (base / +LDS-typed statistics / + deliver_span in phases + one-shard append path + medium-inbox wave sort),
# the GPU suite on the new build, the config-4 leg at 262 144 nodes with the medium-inbox sort
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0base.so _ab/lib_1stats.so _ab/lib_4all.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 50 ) > $O/config4_262k.log 2>&1; tail -5 $O/config4_262k.log
