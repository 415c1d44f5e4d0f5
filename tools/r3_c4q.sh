cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in 8 16; do
timeout 300 python tools/config4_run.py --nodes 262144 --seconds 2000 --every 50 --queue-cap $q > gpurun_out/c4_262k_q$q.log 2>&1
echo "queue_cap $q"; grep -E '"t_s": (51|101|151),' gpurun_out/c4_262k_q$q.log | cut -c1-200; tail -1 gpurun_out/c4_262k_q$q.log
done
grep -E '"t_s": (51|101|151),' gpurun_out/c4_262k_full.log | cut -c1-200
