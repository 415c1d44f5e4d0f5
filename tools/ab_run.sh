# A/B of the library builds under _ab/: the wave/block clocks of the diagnostic build (if any), a parity subset on the candidate
# builds, then tools/ab_libs.sh (value + heavy-tick kernel times per build)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -f _ab/lib_wclk.so ]; then
  SWIMSIM_LIB=$PWD/_ab/lib_wclk.so python bench.py --main-only --handles 1 --steps 20 --warmup 5 > gpurun_out/wclk.json 2> gpurun_out/wclk.err
  grep " clk\]" gpurun_out/wclk.err
  mv _ab/lib_wclk.so _ab/wclk.so.done
fi
for f in _ab/lib_*.so; do
  case $f in *base*) continue;; esac
  echo "== parity with $f"
  SWIMSIM_LIB=$PWD/$f timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_coordinates_gpu.py tests/test_properties_gpu.py -m gpu -x -q 2>&1 | tail -4
done
bash tools/ab_libs.sh
# ...and the default window (where k_begin is the dominant kernel), untraced
for f in _ab/lib_*.so; do
  SWIMSIM_LIB=$PWD/$f python bench.py --main-only 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f default window: value %.3e  ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
# ...and the driver's window on ONE handle, untraced (what the roofline pass instruments)
for f in _ab/lib_*.so; do
  SWIMSIM_LIB=$PWD/$f python bench.py --main-only --handles 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$f driver window, one handle: value %.3e  ms/step %.4f' % (d['value'], d['ms_per_step']))"
done
