# Round 5, call 15: more library handles than three ran 25 % SLOWER in rounds 2-5 — a hardware-queue limit?  GPU_MAX_HW_QUEUES (ROCm's default: 4) and kernel arguments in device memory
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05o; mkdir -p $O
run() { # label, handles
  timeout 120 python bench.py --handles $2 --steps 20 --warmup 5 --main-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1 handles $2: value %.4e ms/round %.4f' % (d['value'], d['ms_per_step']))" | tee -a $O/hwq.txt
}
run default 3
for q in 8 16; do for h in 3 4 6 8; do GPU_MAX_HW_QUEUES=$q run "GPU_MAX_HW_QUEUES=$q" $h; done; done
HIP_FORCE_DEV_KERNARG=1 run "HIP_FORCE_DEV_KERNARG=1" 3
GPU_MAX_HW_QUEUES=8 HIP_FORCE_DEV_KERNARG=1 run "both(8)" 4
run default 3
