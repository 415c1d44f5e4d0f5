# A/B of library builds under _ab/ on the config-#4 leg (a push-pull hands one node a whole view table) next to the headline:
# the parity tests that exercise big inboxes on every non-base build, then per build the driver's line without the CPU legs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in _ab/lib_*.so; do
  case $f in *base*) continue;; esac
  echo "== parity with $f"
  SWIMSIM_LIB=$PWD/$f timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_scale_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q -k "not library_exchange and not bench_two" 2>&1 | tail -4
done
for f in _ab/lib_*.so; do
  SWIMSIM_LIB=$PWD/$f python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-convergence --no-detection 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config4']; r=c['roofline']['per_kernel']
print('$f value %.3e single_handle %.3e | config4 %.3e node-rounds/s, %.2f ms/round, k_resolve %.0f us, k_begin %.0f us, k_deliver %.0f us per launch' % (d['value'], d['single_handle']['value'], c['value'], c['ms_per_step'], r['k_resolve']['avg_launch_us'], r['k_begin']['avg_launch_us'], r['k_deliver']['avg_launch_us']))"
done
