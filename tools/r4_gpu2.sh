# round 4, GPU call 2: tile buckets + receiver-side filter — the GPU suite, then A/B against the same library with SWIMSIM_TILEBUCKETS=0
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
for tb in 0 1; do
  for rep in 1 2; do
    SWIMSIM_TILEBUCKETS=$tb python bench.py --handles 1 --steps 20 --warmup 5 --no-cpu-baseline --no-detection --no-config4 --no-config5 --no-convergence 2>$O/bench_tb$tb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('tile buckets $tb driver window #$rep: value %.4e ms/round %.4f |' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f us (%.4f)' % (k, v['avg_launch_us'], v.get('frac', 0)) for k, v in pk.items()))" | tee -a $O/ab.txt
  done
  SWIMSIM_TILEBUCKETS=$tb python bench.py --main-only --handles 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile buckets $tb default window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
  SWIMSIM_TILEBUCKETS=$tb python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile buckets $tb driver window, 3 handles: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
done
