#!/bin/bash
# tools/emu/sweep.sh [schedules|asan|race|kats|all] — the sanitizer passes over the kernels' source on the host (tools/emu/README.md):
#   schedules  the emulated CPU tests under reversed and seeded-random schedules (EMU_SCHED=7, EMU_SEED=11..14) and with poisoned
#              device memory (EMU_POISON=0xA5): results must not depend on the order of waves / workgroups, nor on zero-filled hipMalloc
#   asan       the same tests on the AddressSanitizer + UBSan build: every load and store of the kernels in bounds
#   race       the same tests (+ the dense-store / fold / serf-intent gpu-marked tests) on the race-detector build; the symbolised report
#              goes to stdout (triage: profiles/r04_emu_race_report.txt)
#   kats       the CPU suite's known-answer and behaviour tests, which drive the CHECKER (fixture `oracle`), with the emulated kernels in its
#              place (SWIMSIM_ORACLE_SO): upstream's recalled tables, the state tables, queue accounting, Lifeguard, reconnect, folds,
#              serf events and intents, checkpoints, coordinates, the bridge — the same assertions on the device code
set -e
cd "$(dirname "$0")/../.."
what=${1:-all}
T="tests/test_emulated_kernels.py -q -p no:cacheprovider"
if [ $what = schedules ] || [ $what = all ]; then
  tools/emu/build.sh > /dev/null
  for s in 1 2 7; do echo "EMU_SCHED=$s: $(EMU_SCHED=$s python -m pytest $T 2>&1 | tail -1)"; done
  for s in 11 12 13 14; do echo "EMU_SEED=$s: $(EMU_SEED=$s python -m pytest $T -k 'not gloo' 2>&1 | tail -1)"; done
  echo "EMU_POISON=0xA5: $(EMU_POISON=0xA5 python -m pytest $T 2>&1 | tail -1)"
fi
if [ $what = asan ] || [ $what = all ]; then
  tools/emu/build.sh asan > /dev/null
  RT=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
  ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 LD_PRELOAD=$RT \
    SWIMSIM_EMU_SO=$PWD/tools/emu/_build/libswimsim_emu_asan.so python -m pytest $T -k 'not gloo' ${EMU_ASAN_MORE} 2>&1 | tee /tmp/emu_asan.log | tail -2
  ! grep -qi "runtime error\|AddressSanitizer" /tmp/emu_asan.log
fi
if [ $what = race ] || [ $what = all ]; then
  tools/emu/build.sh race > /dev/null
  rm -f /tmp/emu_race.txt
  EMU_RACE_OUT=/tmp/emu_race.txt SWIMSIM_EMU_SO=$PWD/tools/emu/_build/libswimsim_emu_race.so python -m pytest $T -k 'not gloo and not product' 2>&1 | tail -1
  EMU_RACE_OUT=/tmp/emu_race.txt SWIMSIM_EMU_SO=$PWD/tools/emu/_build/libswimsim_emu_race.so python -m pytest tests/test_mass_gpu.py tests/test_serf_intents_gpu.py tests/test_membership.py \
    -m gpu -q -p no:cacheprovider --timeout 900 -k 'not sharded and not checkpoint and not both_directions' 2>&1 | tail -1
  python tools/emu/race_report.py /tmp/emu_race.txt
fi
if [ $what = kats ] || [ $what = all ]; then
  tools/emu/build.sh > /dev/null
  SWIMSIM_ORACLE_SO=$PWD/tools/emu/_build/libswimsim_emu.so python -m pytest tests/test_oracle_kat.py tests/test_membership.py tests/test_reconnect.py tests/test_views_fold.py \
    tests/test_serf_events.py tests/test_detection_and_watch.py tests/test_serf_intents.py tests/test_state_table.py tests/test_checkpoint.py tests/test_coordinates.py \
    tests/test_transport_bridge.py -m "not gpu" -q -p no:cacheprovider -n 6 --timeout 900 2>&1 | tail -3
fi
