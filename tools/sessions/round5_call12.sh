# Round 5, call 12: the headline's own arrangement (the clusters on THREE handles) with the one-wave k_resolve against the 256-thread tile
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05l; mkdir -p $O
for rep in 1 2; do for v in oldgeo wave; do for h in 3 1; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 120 python bench.py --handles $h --steps 20 --warmup 5 --main-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v handles $h #$rep: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab3.txt
done; done; done
