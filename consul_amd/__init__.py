"""consul_amd — MI355X-native simulator of Consul's Serf/memberlist SWIM gossip hot path.

Product path: consul_amd.lib (loader of libswimsim.so, the HIP library; no CPU fallback) -> consul_amd.sim.Sim (ctypes
wrapper over the C-ABI of include/swimsim.h) -> consul_amd.dist (a population sharded over several devices: the
library's own mailbox exchange, or RCCL through torch.distributed).  consul_amd.abi mirrors the header's structs.
The host-side mirror of the reference's *serf.Serf / memberlist.Config interface is C++ (include/swimsim_serf.hpp,
include/swimsim_wire.hpp), the language-neutral boundary is the C-ABI.
"""
from . import abi  # noqa: F401
from .sim import Sim, SwimError, derive, preset  # noqa: F401


def open_sim(which: int = abi.PRESET_LAN, **overrides) -> Sim:
    """Create a simulator on the HIP library from a memberlist preset plus field overrides."""
    from . import lib
    cdll = lib.load()
    return Sim(cdll, preset(cdll, which, **overrides))
