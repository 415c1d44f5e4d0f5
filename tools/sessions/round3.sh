# The GPU-box sessions of round 3, one after the other (each was one gpurun call; kept as the record of how profiles/r03_* were taken).

# ---- r3_ab.sh
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

# ---- r3_ab2.sh
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

# ---- r3_ab3.sh
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

# ---- r3_bench.sh
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

# ---- r3_bigsort.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03g
timeout 200 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/r03g/c4_262k_bigsort.log 2>&1
tail -3 gpurun_out/r03g/c4_262k_bigsort.log
( time timeout 500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r03g/pytest_gpu.log 2>&1; tail -6 gpurun_out/r03g/pytest_gpu.log

# ---- r3_c4.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 200 python tools/config4_run.py --nodes 65536 --seconds 400 --every 20 --profile > gpurun_out/c4_65k.log 2>&1
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 150 --every 10 --profile > gpurun_out/c4_262k.log 2>&1
timeout 400 python tools/config4_run.py --nodes 524288 --seconds 60 --every 10 --profile > gpurun_out/c4_524k.log 2>&1
tail -4 gpurun_out/c4_65k.log; tail -3 gpurun_out/c4_262k.log; tail -3 gpurun_out/c4_524k.log

# ---- r3_c4b.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/config4_run.py --nodes 65536 --seconds 2000 --every 50 > gpurun_out/c4_65k_full.log 2>&1
timeout 420 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 50 > gpurun_out/c4_262k_full.log 2>&1
tail -3 gpurun_out/c4_65k_full.log; tail -3 gpurun_out/c4_262k_full.log

# ---- r3_c4c.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mass_gpu.py tests/test_reconnect.py "tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden" tests/test_scale_gpu.py::test_partition_heal_parity -x -q 2>&1 | tail -5
timeout 420 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 50 --profile > gpurun_out/c4_262k_v2.log 2>&1
grep -E '"t_s": (51|151),' gpurun_out/c4_262k_v2.log | cut -c1-120; tail -2 gpurun_out/c4_262k_v2.log

# ---- r3_c4d.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_mass_gpu.py tests/test_reconnect.py "tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden" -x -q 2>&1 | tail -4
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/c4_262k_v3.log 2>&1
tail -2 gpurun_out/c4_262k_v3.log

# ---- r3_c4full.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 800 python tools/config4_run.py --nodes 524288 --seconds 2500 --every 50 > gpurun_out/c4_524k_full.log 2>&1
tail -2 gpurun_out/c4_524k_full.log

# ---- r3_c4q.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in 8 16; do
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 50 --queue-cap $q > gpurun_out/c4_262k_q$q.log 2>&1
echo "queue_cap $q"; grep -E '"t_s": (51|101|151),' gpurun_out/c4_262k_q$q.log | cut -c1-200; tail -1 gpurun_out/c4_262k_q$q.log
done
grep -E '"t_s": (51|101|151),' gpurun_out/c4_262k_full.log | cut -c1-200

# ---- r3_diag.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03i
for T in 20000 45000; do
SWIMSIM_LIB=consul_amd/libswimsim_diag.so SWIMSIM_RESOLVECLK=1 timeout 60 python - $T > gpurun_out/r03i/diag_$T.log 2>&1 <<'PY'
import sys, time
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset
hip = lib.load(); n = 65536; nv = 3276; T = int(sys.argv[1])
s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, seed=11, queue_cap=32, inbox_cap=6808, subject_cap=8, view_cap=8, mass_rows=nv + 8))
s.step_ms(1000); s.kill(0, list(range(0, n, 20))[:nv])
t0 = time.time(); s.step_ms(T - 500); s.sync(); t1 = time.time(); s.step_ms(500); s.sync(); t2 = time.time()
st = s.stats(); print(T, "wall", round(t1 - t0, 2), "last 5 ticks ms/tick", round((t2 - t1) * 200, 2), "inbox_peak", st["inbox_peak"], "push_pulls", st["push_pulls"])
s.close()
PY
cat gpurun_out/r03i/diag_$T.log | tail -12
done

# ---- r3_final.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f
mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 700 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | tail -4
( time python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
wc -c $O/*.json

# ---- r3_gpu_a.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 300 python -m pytest tests/test_mass_gpu.py -x -q 2>&1 | tail -5
bash tools/r3_ab.sh

# ---- r3_gpu_b.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_mass_gpu.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden --deselect tests/test_scale_gpu.py::test_churn_and_event_flood_8192_matches_golden --ignore tests/test_mass_gpu.py 2>&1 | tail -15

# ---- r3_gpu_c.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_scale_gpu.py::test_mass_failure_of_five_percent_65536_matches_golden 2>&1 | tail -15

# ---- r3_merge.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03h
timeout 200 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 100 --profile > gpurun_out/r03h/c4_262k_pipeline.log 2>&1
tail -3 gpurun_out/r03h/c4_262k_pipeline.log
( time timeout 400 python -m pytest tests/test_mass_gpu.py tests/test_reconnect.py tests/test_scale_gpu.py tests/test_membership.py -m gpu -x -q ) > gpurun_out/r03h/pytest_gpu.log 2>&1; tail -6 gpurun_out/r03h/pytest_gpu.log

# ---- r3_profiles.sh
# one GPU-box session of round 3: the bench at the driver's arguments, its kernel-trace summary, the timed region's tick breakdown,
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes), L2 request counters of the tick kernels and of the scattered-access probe
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
mkdir -p $O/fullcmd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/fullcmd/bench.json 2> $O/fullcmd/bench.err
cp $(find $O/fullcmd/trace -name "*kernel_stats.csv" | head -1) $O/driver_fullcmd_kernel_stats.csv; rm -rf $O/fullcmd/trace
SKIP=10 bash tools/trace_pass.sh $O/driver_trace --steps 20 --warmup 5 > $O/driver_trace.log 2>&1
bash tools/pmc_traffic_pass.sh $O/pmc_driver --steps 20 --warmup 5 > $O/pmc_driver.log 2>&1
PMC_GROUPS="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum;TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/pmc_pass.sh $O/pmc_l2 --steps 20 --warmup 5 > $O/pmc_l2.log 2>&1
python tools/pmc_report.py $O/pmc_l2 8 > $O/pmc_l2_heavy_ticks.txt 2>&1
python tools/pmc_report.py $O/pmc_l2 400 > $O/pmc_l2_all_ticks.txt 2>&1
rm -rf $O/pmc_l2
mkdir -p $O/scatter_pmc
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/scatter_pmc/p -- ./tools/scatter_roofline --quick > $O/scatter_quick.txt 2> $O/scatter_pmc.err
python - <<'PY' > gpurun_out/r03/scatter_l2_requests.txt 2>&1
import csv, glob, collections
rows = []
for p in glob.glob("gpurun_out/r03/scatter_pmc/p/*/*_counter_collection.csv"):
    tr = {r["Dispatch_Id"]: r for r in csv.DictReader(open(p.replace("counter_collection", "kernel_trace")))}
    acc = collections.defaultdict(dict)
    for r in csv.DictReader(open(p)):
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = acc[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d, c in sorted(acc.items(), key=lambda kv: int(kv[0])):
        t = tr.get(d)
        if not t: continue
        us = (int(t["End_Timestamp"]) - int(t["Start_Timestamp"])) / 1e3
        print(t["Kernel_Name"][:40], "grid", t.get("Grid_Size"), "us %.1f" % us, {k: int(v) for k, v in c.items()}, "Greq/s %.1f" % (c.get("TCC_REQ_sum", 0) / us / 1e3))
PY
rm -rf $O/scatter_pmc
ls -la $O

# ---- r3_profiles_b.sh
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03
mkdir -p $O/fullcmd
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/fullcmd/bench.json 2> $O/fullcmd/bench.err
cp $(find $O/fullcmd/trace -name "*kernel_stats.csv" | head -1) $O/driver_fullcmd_kernel_stats.csv; rm -rf $O/fullcmd/trace
tail -2 $O/bench_driver.err; wc -c $O/bench_driver.json $O/bench_default.json $O/fullcmd/bench.json
