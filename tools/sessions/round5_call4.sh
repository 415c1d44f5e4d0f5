# Round 5, call 4: where k_resolve's waves and k_begin's roles spend their time in the driver window (a -DSWIMSIM_DIAG build; its clocks cost ~10 %)
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05d; mkdir -p $O
SWIMSIM_LIB=$PWD/_ab/lib_diag.so SWIMSIM_RESOLVECLK=1 SWIMSIM_ROLECLK=$PWD/$O/roleclk.txt timeout 240 python bench.py --main-only --handles 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/diag.err
grep -a "resolve clk" $O/diag.err | tail -12
tail -45 $O/roleclk.txt
