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
