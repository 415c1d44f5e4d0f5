# Round 5, call 17: where a wave of k_resolve<MASS> spends its life in config #4's mass phase (262 144 nodes, the first 60 s), per-kernel times of the same phase
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05q; mkdir -p $O
SWIMSIM_LIB=$PWD/_ab/lib_diag.so SWIMSIM_RESOLVECLK=1 timeout 200 python - > $O/c4_diag.out 2> $O/c4_diag.err <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from consul_amd import abi, lib as L
from consul_amd.sim import Sim, preset
lib = L.load(); n = 262144; nv = n // 20
s = Sim(lib, preset(lib, abi.PRESET_LAN, n_nodes=n, seed=11, queue_cap=32, inbox_cap=min(2 * nv + 256, 8192), subject_cap=8, view_cap=8, mass_rows=nv + 8))
victims = np.random.default_rng(44).choice(n, size=nv, replace=False)
s.step_ms(1000); s.kill(0, victims.tolist()); s.sync()
s.profile(True)
s.step_ms(40000); s.sync()
print({k: (v[0], round(v[1], 2)) for k, v in s.profile_read().items()})
s.close()
PY
cat $O/c4_diag.out; grep -a "resolve clk" $O/c4_diag.err | tail -10 | tee $O/c4_resolve_clk.txt
