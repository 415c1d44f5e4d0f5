# round 4, GPU call 3: tile buckets drained by k_deliver (+ the carry role), and the 64-byte node record: parity, then A/B by environment switches
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_properties_gpu.py tests/test_scale_gpu.py -m gpu -x -q ) > $O/pytest_tb.log 2>&1; tail -15 $O/pytest_tb.log
( time SWIMSIM_LIB=$PWD/_ab/lib_8nl.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q ) > $O/pytest_nl.log 2>&1; tail -15 $O/pytest_nl.log
run() {  # name, lib, env...
  name=$1; lib=$2; shift 2
  for rep in 1 2; do
    env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --handles 1 --steps 20 --warmup 5 --no-cpu-baseline --no-detection --no-config4 --no-config5 --no-convergence 2>$O/bench_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$name driver window #$rep: value %.4e ms/round %.4f |' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f us (%.4f)' % (k, v['avg_launch_us'], v.get('frac', 0)) for k, v in pk.items()))" | tee -a $O/ab.txt
  done
  env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --main-only --handles 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name default window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
  env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name driver window, 3 handles: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
}
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 ) > $O/config4_262k.log 2>&1; tail -3 $O/config4_262k.log
run base _ab/lib_7tbd.so SWIMSIM_TILEBUCKETS=0
run tb_nocarry _ab/lib_7tbd.so SWIMSIM_TB_CARRY=0
run tb _ab/lib_7tbd.so X=1
run nl_base _ab/lib_8nl.so SWIMSIM_TILEBUCKETS=0
run nl_tb _ab/lib_8nl.so X=1
