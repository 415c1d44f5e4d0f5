"""One rank of the on-device check of consul_amd/dist.py TorchExchange with the nccl (= RCCL) backend: the split tick, the frames
packed and read by the product library's kernels, the all_to_all_single issued on the simulator's own HIP stream — with as many
ranks as there are devices (one on the test box: the frame for oneself stays home, the collective still runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from consul_amd import abi, lib  # noqa: E402
from consul_amd.dist import ShardedSim, TorchExchange  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402


def scenario(s):
    s.step_ms(3000)
    s.kill(0, [100, 3000]); s.kill(1, [7]); s.update(1, [2048])
    s.step_ms(20000)
    s.sync()


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    except Exception as e:                                   # noqa: BLE001
        print(f"RESULT nccl-unavailable {type(e).__name__}: {e}")
        return
    hip = lib.load()
    kw = dict(n_nodes=4096, n_replicas=2, seed=5, subject_cap=256, view_cap=256, queue_cap=16, inbox_cap=1024,
              loss_q32=int(0.05 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK, device=dev)
    ex = TorchExchange(dist.group.WORLD, dev)
    sh = ShardedSim(Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=rank, n_shards=world, **kw)), ex)
    scenario(sh)
    d = torch.tensor([sh.digest() & 0x7FFFFFFF, (sh.digest() >> 31) & 0x7FFFFFFF, sh.digest() >> 62], dtype=torch.int64, device=f"cuda:{dev}")
    parts = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(parts, d)
    frame = (ex._F, ex.frame_bytes_per_tick)
    sh.close()
    if rank == 0:
        total = sum(int(p[0]) | (int(p[1]) << 31) | (int(p[2]) << 62) for p in parts) & 0xFFFFFFFFFFFFFFFF
        ref = Sim(hip, preset(hip, abi.PRESET_LAN, **kw))
        scenario(ref)
        print(f"RESULT ok={total == ref.digest()} world={world} digest={total:#x} ref={ref.digest():#x} frame_records={frame[0]} wire_bytes_per_tick={frame[1]}")
        ref.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
