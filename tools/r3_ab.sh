# A/B of the bench's timed region with and without the dense pair store (one handle: the configuration the profiles instrument)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 2; do
  for w in "--steps 20 --warmup 5" ""; do
    python bench.py --gpus 1 $w --main-only --handles 1 --mass-rows $m 2>/dev/null > gpurun_out/ab_tmp.json
    python - "$m" "$w" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"mass_rows={sys.argv[1]} window=[{sys.argv[2] or 'default'}] value={d['value']:.4g} ms_per_step={d['ms_per_step']:.4f}")
PY
  done
done
