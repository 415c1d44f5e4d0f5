# round 4, GPU call 8: k_resolve's tile size (2 / 8 node blocks per workgroup against 4); the sharded config-4 leg at 524 288 nodes on two
# ranks (both on this one device) beside the same population unsharded
set -x
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
bash tools/ab_kernels.sh _ab/lib_rtile2.so _ab/lib_rtile8.so > $O/ab_rtile.txt 2>&1; cat $O/ab_rtile.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
( time SWIMSIM_BENCH_C4S_NODES=524288 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 4 --warmup 2 --no-replica-leg --replicas 2 --dist-backend gloo --exchange library > $O/bench_gpus2.json 2> $O/bench_gpus2.err ); tail -3 $O/bench_gpus2.err
python -c "
import json; d=json.loads([l for l in open('$O/bench_gpus2.json') if l.startswith('{')][0]); print(json.dumps(d.get('config4_sharded'))); print('parity', d.get('parity'))"
( time python tools/config4_sharded_check.py --nodes 524288 ) > $O/c4_unsharded_524k.json 2>&1; cat $O/c4_unsharded_524k.json
