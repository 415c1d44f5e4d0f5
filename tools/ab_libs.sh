# A/B of library builds under _ab/ (built with different -D tuning macros): the main leg's value, the heavy-tick kernel times
# from a kernel trace, and (diag builds) the k_resolve phase clocks
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for f in _ab/lib_*.so; do
  echo "== $f"
  out=gpurun_out/ab_$(basename $f .so); mkdir -p $out
  SWIMSIM_RESOLVECLK=1 SWIMSIM_LIB=$PWD/$f rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --main-only --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
  grep "resolve clk" $out/bench.err
  python tools/tick_report.py $out/trace 10 2>&1 | sed -n 2,6p
  python -c "import json; d=json.load(open('$out/bench.json')); print('value %.3e  ms/step %.4f' % (d['value'], d['ms_per_step']))"
  rm -rf $out/trace
done
