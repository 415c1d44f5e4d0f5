# Round 5, call 16: a 4-tick graph tier (what is left of a call after the 16-tick graphs went out one tick at a time)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05p; mkdir -p $O
for rep in 1 2; do for v in before_mid mid; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 120 python bench.py --handles 3 --steps 20 --warmup 5 --main-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v #$rep: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
done; done
timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -2
