#!/bin/bash
# The checker under AddressSanitizer + UBSan: the CPU tests that drive it (stimulus, membership, reconnect, events, folds,
# checkpoints, the bridge, coordinates, sharding), with the instrumented build of oracle/swim_oracle.c.  Any report fails the run.
set -e
cd "$(dirname "$0")/.."
make -s -C oracle asan
export ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" SWIMSIM_ORACLE_SO=oracle/_build/libswim_oracle_asan.so \
  python -m pytest tests/test_oracle_kat.py tests/test_membership.py tests/test_reconnect.py tests/test_detection_and_watch.py tests/test_serf_events.py \
  tests/test_views_fold.py tests/test_checkpoint.py tests/test_state_table.py tests/test_transport_bridge.py tests/test_coordinates.py tests/test_dist_cpu.py \
  tests/test_bench_handles.py -x -q -s -m "not gpu" -p no:cacheprovider 2>&1 | tee /tmp/oracle_asan.log | tail -3
! grep -qi "runtime error\|AddressSanitizer" /tmp/oracle_asan.log
# ... and random configurations / stimulus schedules, sharded against unsharded, on the same instrumented build
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" SWIMSIM_ORACLE_SO=oracle/_build/libswim_oracle_asan.so \
  python tools/fuzz_parity.py --backend oracle --cases ${FUZZ_CASES:-40} --seed ${FUZZ_SEED:-31337} 2>&1 | tee /tmp/oracle_asan_fuzz.log | tail -2
! grep -qi "runtime error\|AddressSanitizer\|mismatch" /tmp/oracle_asan_fuzz.log
