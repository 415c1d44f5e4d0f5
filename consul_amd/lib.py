"""Loader for the product library (consul_amd/libswimsim.so = hand-written HIP for gfx950).

There is deliberately no fallback: if the shared library is missing, or it is not the HIP build,
importing the product path raises.  The CPU oracle under oracle/ is test infrastructure and is
never loaded from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SWIMSIM_LIB") or os.path.join(_HERE, "libswimsim.so")   # (SWIMSIM_LIB: A/B builds of the same HIP library)
SOURCES = [os.path.join(_HERE, "csrc", f) for f in ("swim_host.hip", "swim_kernels.hip", "swim_device.h")]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "swimsim.h")

_cdll = None


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 the kernels + C-ABI into consul_amd/libswimsim.so (in-tree)."""
    srcs = SOURCES + [HEADER]
    if not force and os.path.exists(LIB_PATH) and all(
            os.path.getmtime(LIB_PATH) >= os.path.getmtime(p) for p in srcs if os.path.exists(p)):
        return LIB_PATH
    if not all(os.path.exists(p) for p in srcs):
        if os.path.exists(LIB_PATH):
            return LIB_PATH
        raise FileNotFoundError("libswimsim.so sources are missing")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value", "-shared", "-fPIC",
           "-o", LIB_PATH, SOURCES[0]]
    subprocess.run(cmd, check=True)
    return LIB_PATH


def load() -> C.CDLL:
    global _cdll
    if _cdll is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the SWIM hot path has no CPU fallback; build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'`")
        cdll = abi.bind(C.CDLL(LIB_PATH))
        backend = cdll.swim_backend().decode()
        if backend != "hip-gfx950":
            raise ImportError(f"{LIB_PATH} reports backend {backend!r}, expected 'hip-gfx950'")
        _cdll = cdll
    return _cdll
