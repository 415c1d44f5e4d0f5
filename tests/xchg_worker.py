"""One rank of the two-process test of the library's device-driven exchange (swim_xchg_*): both ranks on device 0, the
mailbox handles passed through files (hipIpc between processes of one device).  usage: xchg_worker.py <rank> <world> <dir>"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from consul_amd import abi, lib  # noqa: E402
from consul_amd.dist import LibraryExchange, ShardedSim  # noqa: E402
from consul_amd.sim import Sim, preset  # noqa: E402
import xchg_scenario as xs  # noqa: E402


def main():
    rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    hip = lib.load()

    def gather(mine):
        for r, h in mine.items():
            tmp = os.path.join(d, f"h{r}.tmp")
            open(tmp, "wb").write(h); os.rename(tmp, os.path.join(d, f"h{r}"))
        out, t0 = [], time.time()
        for r in range(world):
            p = os.path.join(d, f"h{r}")
            while not os.path.exists(p):
                if time.time() - t0 > 120:
                    raise TimeoutError(f"no handle from rank {r}")
                time.sleep(0.01)
            out.append(open(p, "rb").read())
        return out

    sim = Sim(hip, preset(hip, abi.PRESET_LAN, shard_rank=rank, n_shards=world, device=0, **xs.KW))
    sh = ShardedSim(sim, LibraryExchange(gather))
    xs.run(sh)
    sh.sync()
    st = sim.stats()
    open(os.path.join(d, f"r{rank}.tmp"), "w").write(f"{sim.digest()} {st['edges_remote']} {st['folds']} {st['refutes']} {sum(st['msgs_applied'])}")
    os.rename(os.path.join(d, f"r{rank}.tmp"), os.path.join(d, f"r{rank}"))
    # keep the mailbox mapped until everybody is done with it
    t0 = time.time()
    while not all(os.path.exists(os.path.join(d, f"r{r}")) for r in range(world)) and time.time() - t0 < 120:
        time.sleep(0.01)
    sh.close()


if __name__ == "__main__":
    main()
