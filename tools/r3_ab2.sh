cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for tree in "$GRAFT_REPO_ROOT/_r02tree" "$GRAFT_REPO_ROOT"; do
  cd $tree
  for w in "--steps 20 --warmup 5" ""; do
    python bench.py --gpus 1 $w --main-only --handles 1 2>/dev/null > /tmp/ab.json
    python - "$tree" "$w" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1].split('/')[-1]:10s} window=[{sys.argv[2] or 'default'}] value={d['value']:.4g} ms_per_step={d['ms_per_step']:.4f}")
PY
  done
done
done
cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
