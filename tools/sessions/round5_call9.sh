# Round 5, call 9: config #4's partition + recovery leg per kernel, with inboxes beyond 8 192 messages sorted by a workgroup (k_inbox_sort_huge)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05i; mkdir -p $O
COMMON="--steps 2 --warmup 2 --handles 1 --no-detection --no-convergence --no-cpu-baseline --no-roofline --no-config5 --config4-nodes 65536"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4p -- python bench.py $COMMON --config4p-budget-s 75 > $O/c4p_bench.json 2> $O/c4p.err
f=$(ls $O/c4p/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-160
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05i/c4p_bench.json').read().strip().splitlines()[-1])
for k in ('config4','config4_partition'):
    c=d.get(k,{})
    print(k, {a:b for a,b in c.items() if not isinstance(b,(dict,list)) and a!='workload'})
    if 'curve' in c: print([ (p['t_s'], p['wall_s'], p['not_alive_seen_by_watchers']) for p in c['curve']])
PY
rm -f $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
