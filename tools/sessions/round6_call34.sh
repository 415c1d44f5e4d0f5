# Round 6, call 34: the randomised cases with the message lengths drawn too (--lens: one, two or three length ranks), with and without the unbounded queue
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r07g; mkdir -p $O
( time timeout 600 python tools/fuzz_parity.py --unbounded --lens --cases 200 --seed 606 ) > $O/fuzz_uq_lens_606.log 2>&1; tail -4 $O/fuzz_uq_lens_606.log
( time timeout 600 python tools/fuzz_parity.py --unbounded --lens --cases 200 --seed 2024 ) > $O/fuzz_uq_lens_2024.log 2>&1; tail -4 $O/fuzz_uq_lens_2024.log
( time timeout 600 python tools/fuzz_parity.py --lens --cases 150 --seed 606 ) > $O/fuzz_lens_606.log 2>&1; tail -4 $O/fuzz_lens_606.log
