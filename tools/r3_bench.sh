cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver.json 2> gpurun_out/r03_bench_driver.err; echo "bench wall $(( $(date +%s) - S )) s"; tail -3 gpurun_out/r03_bench_driver.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_bench_driver.json"))
print("value",d["value"],"ms/step",d["ms_per_step"])
print("roofline",{k:d["roofline"][k] for k in ("kernel","frac","avg_launch_us")}, d["roofline"]["kernel_time_share"])
print("per_kernel",{k:(round(v["avg_launch_us"],1),round(v["frac"],4)) for k,v in d["roofline"]["per_kernel"].items()})
c=d["config4"]; print("config4",{k:c[k] for k in ("n_nodes","victims","detection_complete","rounds_to_full_detection","wall_s","rounds_per_sec","view_drops","queue_drops","inbox_peak","first_60_s")})
c=d["config5"]; print("config5",{k:c[k] for k in c if k not in ("workload",)})
PY
