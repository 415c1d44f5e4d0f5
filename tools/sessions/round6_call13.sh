# Round 6, call 13: config #5's leg with its kernel times (bench.py without the config-4 legs)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06m; mkdir -p $O
( time timeout 600 python bench.py --steps 20 --warmup 5 --no-config4 --no-config4-partition ) > $O/bench_c5.json 2> $O/bench_c5.err; tail -2 $O/bench_c5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06m/bench_c5.json')); c=d['config5']
print({k:c[k] for k in ('wall_s','rounds_per_sec','kernel_ms_total','mean_coverage_of_an_event','queue_drops','folds','refutes','inbox_peak')})
PY
