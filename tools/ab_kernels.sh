#!/bin/bash
# A/B of library builds (default: every _ab/lib_*.so): the driver window on ONE handle with the HIP-event roofline pass — value, ms per round
# and the three tick kernels' average microseconds per launch — and the default window's value.  usage: tools/ab_kernels.sh [libs...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(_ab/lib_*.so)
for f in "${libs[@]}"; do
  for rep in 1 2; do
    SWIMSIM_LIB=$PWD/$f python bench.py --handles 1 --steps 20 --warmup 5 --no-cpu-baseline --no-detection --no-config4 --no-config5 --no-convergence 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$f driver window #$rep: value %.4e ms/round %.4f |' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f us (%.4f)' % (k, v['avg_launch_us'], v.get('frac', 0)) for k, v in pk.items()))"
  done
  SWIMSIM_LIB=$PWD/$f python bench.py --main-only --handles 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$f default window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))"
done
