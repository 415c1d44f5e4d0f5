# Round 6, call 20: after k_fold_scan_slots — the randomised unbounded cases again (seed 606 x200, seed 17 x100, seed 2024 x200), then the whole GPU suite
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06t; mkdir -p $O
( time timeout 1200 python tools/fuzz_parity.py --unbounded --cases 200 --seed 606 ) > $O/fuzz_unbounded_606.log 2>&1; tail -4 $O/fuzz_unbounded_606.log
( time timeout 900 python tools/fuzz_parity.py --unbounded --cases 100 --seed 17 ) > $O/fuzz_unbounded_17.log 2>&1; tail -4 $O/fuzz_unbounded_17.log
( time timeout 1200 python tools/fuzz_parity.py --unbounded --cases 200 --seed 2024 ) > $O/fuzz_unbounded_2024.log 2>&1; tail -4 $O/fuzz_unbounded_2024.log
( time timeout 900 python tools/fuzz_parity.py --cases 200 --seed 606 ) > $O/fuzz_bounded_606.log 2>&1; tail -3 $O/fuzz_bounded_606.log
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
