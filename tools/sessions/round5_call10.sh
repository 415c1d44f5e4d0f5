# Round 5, call 10: config #4's partition + recovery leg after the reconnect fix, with room to finish (how long does the recovery take at 65 536 nodes?)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05j; mkdir -p $O
COMMON="--steps 2 --warmup 2 --handles 1 --no-detection --no-convergence --no-cpu-baseline --no-roofline --no-config5 --config4-nodes 65536"
timeout 400 python bench.py $COMMON --config4p-budget-s 240 > $O/c4p_bench.json 2> $O/c4p.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05j/c4p_bench.json').read().strip().splitlines()[-1])
c=d.get('config4_partition',{})
print({a:b for a,b in c.items() if not isinstance(b,(dict,list)) and a!='workload'})
for p in c.get('curve',[]): print(p)
PY
