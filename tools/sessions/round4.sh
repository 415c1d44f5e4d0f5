# The GPU-box sessions of round 4, one after the other (each block was one gpurun call; kept as the record of how profiles/r04_* were taken).
# Calls 3 ran twice (the first build had lost two allocations to a stray #endif: memory fault at create); call 8's two-rank run was killed by its time-out.

# ---- r4_gpu1.sh
# round 4, GPU call 1: A/B of the ISA fixes (base / +LDS-typed statistics / + deliver_span in phases + one-shard append path + medium-inbox wave sort),
# the GPU suite on the new build, the config-4 leg at 262 144 nodes with the medium-inbox sort
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0base.so _ab/lib_1stats.so _ab/lib_4all.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 50 ) > $O/config4_262k.log 2>&1; tail -5 $O/config4_262k.log

# ---- r4_gpu2.sh
# round 4, GPU call 2: tile buckets + receiver-side filter — the GPU suite, then A/B against the same library with SWIMSIM_TILEBUCKETS=0
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -30 $O/pytest_gpu.log
for tb in 0 1; do
  for rep in 1 2; do
    SWIMSIM_TILEBUCKETS=$tb python bench.py --handles 1 --steps 20 --warmup 5 --no-cpu-baseline --no-detection --no-config4 --no-config5 --no-convergence 2>$O/bench_tb$tb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('tile buckets $tb driver window #$rep: value %.4e ms/round %.4f |' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f us (%.4f)' % (k, v['avg_launch_us'], v.get('frac', 0)) for k, v in pk.items()))" | tee -a $O/ab.txt
  done
  SWIMSIM_TILEBUCKETS=$tb python bench.py --main-only --handles 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile buckets $tb default window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
  SWIMSIM_TILEBUCKETS=$tb python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tile buckets $tb driver window, 3 handles: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
done

# ---- r4_gpu3.sh
# round 4, GPU call 3: tile buckets drained by k_deliver (+ the carry role), and the 64-byte node record: parity, then A/B by environment switches
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
( time timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_properties_gpu.py tests/test_scale_gpu.py -m gpu -x -q ) > $O/pytest_tb.log 2>&1; tail -15 $O/pytest_tb.log
( time SWIMSIM_LIB=$PWD/_ab/lib_8nl.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q ) > $O/pytest_nl.log 2>&1; tail -15 $O/pytest_nl.log
run() {  # name, lib, env...
  name=$1; lib=$2; shift 2
  for rep in 1 2; do
    env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --handles 1 --steps 20 --warmup 5 --no-cpu-baseline --no-detection --no-config4 --no-config5 --no-convergence 2>$O/bench_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); pk=d['roofline']['per_kernel']
print('$name driver window #$rep: value %.4e ms/round %.4f |' % (d['value'], d['ms_per_step']), ' '.join('%s %.1f us (%.4f)' % (k, v['avg_launch_us'], v.get('frac', 0)) for k, v in pk.items()))" | tee -a $O/ab.txt
  done
  env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --main-only --handles 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name default window: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
  env "$@" SWIMSIM_LIB=$PWD/$lib python bench.py --main-only --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$name driver window, 3 handles: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
}
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 ) > $O/config4_262k.log 2>&1; tail -3 $O/config4_262k.log
run base _ab/lib_7tbd.so SWIMSIM_TILEBUCKETS=0
run tb_nocarry _ab/lib_7tbd.so SWIMSIM_TB_CARRY=0
run tb _ab/lib_7tbd.so X=1
run nl_base _ab/lib_8nl.so SWIMSIM_TILEBUCKETS=0
run nl_tb _ab/lib_8nl.so X=1

# ---- r4_gpu4.sh
# round 4, GPU call 4: the whole GPU suite on the current tree (tile buckets off by default + their own tests), config #4's leg at 524 288 nodes
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 524288 --seconds 1600 --every 100 ) > $O/config4_524k.log 2>&1; tail -4 $O/config4_524k.log

# ---- r4_gpu5.sh
# round 4, GPU call 5: the GPU suite with the receiving-shard filter, where config #4's mass phase spends its time (per-kernel HIP events)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 120 --every 20 --profile ) > $O/config4_262k_profile.log 2>&1; tail -4 $O/config4_262k_profile.log

# ---- r4_gpu6.sh
# round 4, GPU call 6: serf intent ordering on the device, one view lookup per membership rumour; suite, headline check, config-4 leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
( time timeout 600 python -m pytest tests/test_serf_intents_gpu.py -m gpu -x -q ) > $O/pytest_intents.log 2>&1; tail -25 $O/pytest_intents.log
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
bash tools/ab_kernels.sh _ab/lib_0base.so consul_amd/libswimsim.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 --profile ) > $O/config4_262k.log 2>&1; tail -4 $O/config4_262k.log | cut -c1-300

# ---- r4_gpu7.sh
# round 4, GPU call 7: k_resolve compiled without serf's handlers for handles without an event layer: headline A/B (reference = the build before the
# intent work), the intent scripts, the suite, config #4's leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_0ref.so consul_amd/libswimsim.so > $O/ab.txt 2>&1; cat $O/ab.txt
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
( time python tools/config4_run.py --nodes 262144 --seconds 1300 --every 100 ) > $O/config4_262k.log 2>&1; tail -3 $O/config4_262k.log | cut -c1-200

# ---- r4_gpu8.sh
# round 4, GPU call 8: k_resolve's tile size (2 / 8 node blocks per workgroup against 4); the sharded config-4 leg at 524 288 nodes on two
# ranks (both on this one device) beside the same population unsharded
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_rtile2.so _ab/lib_rtile8.so > $O/ab_rtile.txt 2>&1; cat $O/ab_rtile.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
( time SWIMSIM_BENCH_C4S_NODES=524288 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 4 --warmup 2 --no-replica-leg --replicas 2 --dist-backend gloo --exchange library > $O/bench_gpus2.json 2> $O/bench_gpus2.err ); tail -3 $O/bench_gpus2.err
python -c "
import json; d=json.loads([l for l in open('$O/bench_gpus2.json') if l.startswith('{')][0]); print(json.dumps(d.get('config4_sharded'))); print('parity', d.get('parity'))"
( time python tools/config4_sharded_check.py --nodes 524288 ) > $O/c4_unsharded_524k.json 2>&1; cat $O/c4_unsharded_524k.json

# ---- r4_gpu9.sh
# round 4, GPU call 9: k_resolve<MASS, SERF, DYN> against k_resolve<MASS, SERF>; the suite on it
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04i; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_serf.so _ab/lib_dyn.so > $O/ab_dyn.txt 2>&1; cat $O/ab_dyn.txt
cp _ab/lib_dyn.so consul_amd/libswimsim.so
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log

# ---- r4_final.sh
# round 4, the final GPU-box session: smoke, the GPU suite, the bench at the driver's arguments and at the defaults, the kernel trace and
# tick breakdown of the timed region, HBM traffic (FETCH_SIZE / WRITE_SIZE passes), L2 / SQ counters, the kernel table of the driver's exact command
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | tail -4
( time python bench.py --no-config4 --no-config5 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -4
wc -c $O/*.json
SKIP=10 bash tools/trace_pass.sh $O/driver_trace --steps 20 --warmup 5 > $O/driver_trace.log 2>&1; tail -12 $O/driver_trace.log
bash tools/pmc_traffic_pass.sh $O/pmc_driver --steps 20 --warmup 5 > $O/pmc_driver.log 2>&1; tail -3 $O/pmc_driver.log | cut -c1-300
PMC_GROUPS="TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum;GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" bash tools/pmc_pass.sh $O/pmc_l2 --steps 20 --warmup 5 > $O/pmc_l2.log 2>&1
python tools/pmc_report.py $O/pmc_l2 8 > $O/pmc_l2_heavy_ticks.txt 2>&1
python tools/pmc_report.py $O/pmc_l2 400 > $O/pmc_l2_all_ticks.txt 2>&1
rm -rf $O/pmc_l2
mkdir -p $O/fullcmd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fullcmd/trace -- python bench.py --gpus 1 --steps 20 --warmup 5 > $O/fullcmd/bench.json 2> $O/fullcmd/bench.err
cp $(find $O/fullcmd/trace -name "*kernel_stats.csv" | head -1) $O/driver_fullcmd_kernel_stats.csv; rm -rf $O/fullcmd/trace
ls -la $O

# ---- cpu_baseline_sweep (one call)
python tools/cpu_baseline_sweep.py 16 32 64 128

# ---- r4_last.sh
# round 4, last GPU call: how many library handles (HIP streams) the clusters are best spread over with this build, then the bench line at the
# driver's arguments once more (cpu_baseline as a thread sweep, roofline.traffic from this round's --pmc passes)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O
for h in 1 2 3 4 6; do
  for w in "--steps 20 --warmup 5" ""; do
    python bench.py --main-only --handles $h $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('handles $h window [$w]: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/handles.txt
  done
done
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('$O/bench_driver.json')); print(d['value'], d['single_handle'], d['cpu_baseline'], d['roofline']['frac'], d['roofline'].get('traffic_over_algorithmic'))"

# ---- r4_gpu10.sh
# round 4, GPU call 10: the join intent handed over by the answer to the join push-pull — the intent scripts, then the whole suite
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
( time timeout 600 python -m pytest tests/test_serf_intents_gpu.py tests/test_membership.py -m gpu -x -q ) > $O/pytest_intents.log 2>&1; tail -25 $O/pytest_intents.log
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
python bench.py --main-only --handles 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('driver window, one handle: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))"

# ---- r4_gpu11.sh
# round 4, GPU call 11: the whole GPU suite on the final tree
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log

# ---- r4_gpu12.sh
# round 4, GPU call 12: the framed exchange (swim_frame_pack / swim_frame_deliver, TorchExchange over RCCL with a world of one) — its tests,
# then what the split tick + one collective per tick costs on one device beside the captured-graph replay
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m; mkdir -p $O
( time timeout 400 python -m pytest tests/test_framed_exchange_gpu.py -x -q ) > $O/pytest_framed.log 2>&1; tail -25 $O/pytest_framed.log
timeout 200 python tools/exchange_cost.py > $O/exchange_cost.txt 2>&1; head -5 $O/exchange_cost.txt
timeout 200 python bench.py --force-exchange --exchange rccl --main-only --handles 1 --steps 20 --warmup 5 > $O/bench_force_rccl.json 2> $O/bench_force_rccl.err; tail -3 $O/bench_force_rccl.err; cut -c1-400 $O/bench_force_rccl.json

# ---- r4_gpu13.sh
# round 4, GPU call 13: the framed-exchange tests again (the in-process ones now allocate through the library's own HIP runtime, not torch)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
( time timeout 400 python -m pytest tests/test_framed_exchange_gpu.py -q ) > $O/pytest_framed.log 2>&1; tail -30 $O/pytest_framed.log

# ---- r4_gpu14.sh
# round 4, GPU call 14 (the last of the round's minutes: 7.4): what has never run on the device — the MergeState / Suspicion_Timer KATs on the
# HIP library, the partition-heal-reconnect scenario beside the live checker (2 048), against the checker's fixture (32 768) and as
# size-independent properties (65 536), and the bench line's config4_partition leg (65 536) under its wall-time budget
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04p; mkdir -p $O
( time timeout 150 python -m pytest tests/test_state_table.py tests/test_mass_gpu.py::test_partition_heal_and_reconnect_with_both_directions_in_rows \
    tests/test_scale_gpu.py::test_partition_and_recovery_32768_matches_golden -m gpu -q --durations=6 ) > $O/pytest_new.log 2>&1; tail -25 $O/pytest_new.log
( time timeout 150 python -m pytest tests/test_scale_gpu.py::test_partition_and_recovery_65536_properties -m gpu -q --durations=3 ) > $O/pytest_65k.log 2>&1; tail -25 $O/pytest_65k.log
( time timeout 100 python - <<'PY'
import json, types, bench
from consul_amd import lib
hip = lib.load()
args = types.SimpleNamespace(seed=1, config4p_nodes=65536, config4p_budget_s=70.0)
print(json.dumps(bench.run_config4_partition(hip, args, 0)))
PY
) > $O/config4_partition.json 2> $O/config4_partition.err; tail -c 3000 $O/config4_partition.json; tail -5 $O/config4_partition.err

# ---- r4_gpu15.sh
# round 4, GPU call 15 (3.1 minutes left): the 65 536-node partition / recovery properties as rewritten after call 14 (a minute after the heal, not three)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
( time timeout 120 python -m pytest tests/test_scale_gpu.py::test_partition_and_recovery_65536_properties -m gpu -q --durations=3 ) > $O/pytest_65k.log 2>&1; tail -25 $O/pytest_65k.log

# ---- r4_gpu16.sh
# round 4, GPU call 16: the fold / row-recycling tests on the device after the barrier in k_fold_apply_mass (a race found by tools/emu)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04r; mkdir -p $O
( time timeout 75 python -m pytest tests/test_mass_gpu.py -m gpu -q -x --durations=5 -k "leave_update_revive_join_in_rows or partition_heal_with_rows_and_folds or checkpoint_with_rows or churn" ) > $O/pytest_fold.log 2>&1; tail -15 $O/pytest_fold.log

# ---- r4_gpu17.sh
# round 4, GPU call 17 (the last seconds): smoke() and three parity cases on the FINAL binary (the barrier in k_fold_apply_mass, piggyback() generalised for the SW_MASS_HBMQ option — off)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04s; mkdir -p $O
( time timeout 40 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( time timeout 30 python -m pytest tests/test_parity_gpu.py tests/test_mass_gpu.py -m gpu -q -x -k "lockstep_small or leave_and_revive or churn_recycles" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
