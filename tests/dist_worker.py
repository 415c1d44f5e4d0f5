"""One rank of the world_size-2 CPU test: a shard of the oracle — or, with SWIMSIM_DIST_LIB naming it, of the HIP kernels' source emulated on the
host (tools/emu) — exchanged over gloo; the unsharded reference run is always the oracle's."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from consul_amd import abi  # noqa: E402
from consul_amd.dist import ShardedSim, TorchExchange  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ora = abi.bind(C.CDLL(os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")))
    shard_lib = abi.bind(C.CDLL(os.environ["SWIMSIM_DIST_LIB"])) if os.environ.get("SWIMSIM_DIST_LIB") else ora
    kw = dict(n_nodes=2048, n_replicas=2, seed=5, subject_cap=128, view_cap=128, queue_cap=16, inbox_cap=128,
              loss_q32=int(0.05 * 2**32), flags=abi.F_DEFAULT & ~abi.F_TCP_FALLBACK)
    sh = ShardedSim(Sim(shard_lib, preset(shard_lib, abi.PRESET_LAN, shard_rank=rank, n_shards=world, **kw)),
                    TorchExchange(dist.group.WORLD, None))
    sh.step_ms(3000)
    sh.kill(0, [100, 1500]); sh.kill(1, [7]); sh.update(1, [1024])
    sh.step_ms(25000)
    import torch
    d = torch.tensor([sh.digest() & 0x7FFFFFFF, (sh.digest() >> 31) & 0x7FFFFFFF, sh.digest() >> 62], dtype=torch.int64)
    parts = [torch.zeros_like(d) for _ in range(world)]
    dist.all_gather(parts, d)
    st = sh.stats()
    cnt = torch.tensor([st["packets_sent"], st["edges_remote"], st["refutes"], st["probe_failures"]], dtype=torch.int64)
    dist.all_reduce(cnt)
    if rank == 0:
        total = sum(int(p[0]) | (int(p[1]) << 31) | (int(p[2]) << 62) for p in parts) & 0xFFFFFFFFFFFFFFFF
        ref = Sim(ora, preset(ora, abi.PRESET_LAN, **kw))
        ref.step_ms(3000)
        ref.kill(0, [100, 1500]); ref.kill(1, [7]); ref.update(1, [1024])
        ref.step_ms(25000)
        rs = ref.stats()
        ok = total == ref.digest() and int(cnt[0]) == rs["packets_sent"] and int(cnt[2]) == rs["refutes"] and int(cnt[1]) > 0
        print(f"RESULT ok={ok} digest={total:#x} ref={ref.digest():#x} edges={int(cnt[0])}/{rs['edges']} remote={int(cnt[1])}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
