"""consul_amd — MI355X-native simulator of Consul's Serf/memberlist SWIM gossip hot path.

Product path: consul_amd.lib (HIP library loader) -> consul_amd.sim.Sim (C-ABI wrapper) ->
consul_amd.memberlist / consul_amd.serf (host-side mirror of the reference interface).
"""
from . import abi  # noqa: F401
from .sim import Sim, SwimError, derive, preset  # noqa: F401


def open_sim(which: int = abi.PRESET_LAN, **overrides) -> Sim:
    """Create a simulator on the HIP library from a memberlist preset plus field overrides."""
    from . import lib
    cdll = lib.load()
    return Sim(cdll, preset(cdll, which, **overrides))
