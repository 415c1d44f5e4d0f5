# Round 5, call 8: config #5's leg after the revive fix; config #4's partition + recovery leg per kernel, with inboxes beyond 8 192 messages sorted by a workgroup
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05h; mkdir -p $O
COMMON="--steps 2 --warmup 2 --handles 1 --no-detection --no-config4 --no-convergence --no-cpu-baseline --no-roofline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py $COMMON --no-config4-partition > $O/c5_bench.json 2> $O/c5.err
f=$(ls $O/c5/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-160
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p -- python bench.py $COMMON --no-config5 --config4p-budget-s 60 > $O/c4p_bench.json 2> $O/c4p.err
f=$(ls $O/c4p/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-160
python - <<'PY'
import json
for n,k in (('c5','config5'),('c4p','config4_partition')):
    try:
        d=json.loads(open(f'gpurun_out/r05h/{n}_bench.json').read().strip().splitlines()[-1]); c=d[k]
        print(n, {a:b for a,b in c.items() if not isinstance(b,(dict,list)) and a!='workload'})
        if 'curve' in c: print([ (p['t_s'], p['wall_s'], p['not_alive_seen_by_watchers']) for p in c['curve']])
    except Exception as e: print(n, 'failed', e)
PY
rm -f $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
