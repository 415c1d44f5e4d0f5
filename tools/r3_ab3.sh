cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_mass_gpu.py tests/test_properties_gpu.py tests/test_checkpoint_gpu.py -x -q 2>&1 | tail -6
for w in "--steps 20 --warmup 5" ""; do
  python bench.py --gpus 1 $w --main-only --handles 1 2>/dev/null > /tmp/ab.json
  python - "$w" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"incremental-census build window=[{sys.argv[1] or 'default'}] value={d['value']:.4g} ms_per_step={d['ms_per_step']:.4f}")
PY
done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-config4 --no-config5 --no-cpu-baseline --no-convergence --no-detection 2>/dev/null > /tmp/b.json
python - <<'PY'
import json
d=json.load(open("/tmp/b.json"))
print("value", d["value"], "single", d["single_handle"]["value"])
print({k:(round(v["avg_launch_us"],1),round(v["frac"],4)) for k,v in d["roofline"]["per_kernel"].items()}, d["roofline"]["kernel_time_share"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
