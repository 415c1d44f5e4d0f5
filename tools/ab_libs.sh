# A/B of library builds under _ab/ (built with different -D tuning macros): per-kernel HIP-event averages at the driver's arguments
for f in _ab/lib_*.so; do
  SWIMSIM_LIB=$PWD/$f python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$f', 'value %.3e' % d['value'], ' '.join('%s %.1f' % (k, v['avg_launch_us']) for k,v in pk.items()))"
done
