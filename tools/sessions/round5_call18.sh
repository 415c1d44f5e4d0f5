# Round 5, call 18: the dense pair store laid out by groups of 64 observers ([group][row][64]) against rows outermost ([row][observer]): config #4's leg at 262 144 nodes, config #5's leg
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05r; mkdir -p $O
for v in rowmajor grouped; do
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 300 python tools/config4_run.py --nodes 262144 2>&1 | tail -2 | sed "s/^/$v: /" | cut -c1-330 | tee -a $O/ab_c4.txt
  SWIMSIM_LIB=$PWD/_ab/lib_$v.so timeout 200 python bench.py --steps 2 --warmup 2 --handles 1 --no-detection --no-config4 --no-convergence --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['config5']; print('$v config5: rounds/s %.1f wall %.2f coverage %.3f' % (d['rounds_per_sec'], d['wall_s'], d['mean_coverage_of_an_event']))" | tee -a $O/ab_c5.txt
done
timeout 250 python -m pytest tests/test_mass_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee $O/pytest_mass.txt
