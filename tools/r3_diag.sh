cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03i
for T in 20000 45000; do
SWIMSIM_LIB=consul_amd/libswimsim_diag.so SWIMSIM_RESOLVECLK=1 timeout 60 python - $T > gpurun_out/r03i/diag_$T.log 2>&1 <<'PY'
import sys, time
from consul_amd import abi, lib
from consul_amd.sim import Sim, preset
hip = lib.load(); n = 65536; nv = 3276; T = int(sys.argv[1])
s = Sim(hip, preset(hip, abi.PRESET_LAN, n_nodes=n, seed=11, queue_cap=32, inbox_cap=6808, subject_cap=8, view_cap=8, mass_rows=nv + 8))
s.step_ms(1000); s.kill(0, list(range(0, n, 20))[:nv])
t0 = time.time(); s.step_ms(T - 500); s.sync(); t1 = time.time(); s.step_ms(500); s.sync(); t2 = time.time()
st = s.stats(); print(T, "wall", round(t1 - t0, 2), "last 5 ticks ms/tick", round((t2 - t1) * 200, 2), "inbox_peak", st["inbox_peak"], "push_pulls", st["push_pulls"])
s.close()
PY
cat gpurun_out/r03i/diag_$T.log | tail -12
done
