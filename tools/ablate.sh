#!/bin/bash
# timing-only ablation of the gossip role: one rocprofv3 kernel trace per SWIMSIM_ABLATE mask
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for m in 0 1 2 4 8 16 32 47; do
  SWIMSIM_ABLATE=$m rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ablate/m$m -- python bench.py --no-cpu-baseline --no-roofline --steps 120 > gpurun_out/ablate_m$m.json 2>/dev/null
  echo "mask $m"; python tools/tick_report.py gpurun_out/ablate/m$m 60 | grep -E "k_begin|span"
done
