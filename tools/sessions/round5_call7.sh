# Round 5, call 7: one-line node record under the one-wave k_resolve; frames sized from the load on the device; config #5's leg per kernel
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05g; mkdir -p $O
timeout 200 python -m pytest tests/test_framed_exchange_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_framed.txt
timeout 300 bash tools/ab_kernels.sh _ab/lib_wave.so _ab/lib_wave_nodeline.so 2>&1 | grep -v "default window" | tee $O/ab.txt
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py --steps 2 --warmup 2 --handles 1 --no-detection --no-config4 --no-config4-partition --no-convergence --no-cpu-baseline --no-roofline > $O/c5_bench.json 2> $O/c5.err
tail -2 $O/c5.err
f=$(ls $O/c5/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-200
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05g/c5_bench.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('config5'))[:1500])
PY
rm -f $O/c5/*/*kernel_trace.csv $O/c5/*/*agent_info.csv
