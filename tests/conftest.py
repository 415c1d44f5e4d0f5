import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from consul_amd import abi  # noqa: E402

ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of failing them with SWIM_ENODEV."""
    if os.path.exists("/dev/kfd") or os.environ.get("SWIMSIM_EMU_SO"):
        return
    skip = pytest.mark.skip(reason="no AMD GPU here (/dev/kfd missing): run on the MI355X box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The plain-C checker (oracle/), built on demand."""
    src = os.path.join(ROOT, "oracle", "swim_oracle.c")
    if os.environ.get("SWIMSIM_ORACLE_SO"):            # e.g. oracle/_build/libswim_oracle_asan.so (`make -C oracle asan`; LD_PRELOAD libasan)
        return abi.bind(C.CDLL(os.environ["SWIMSIM_ORACLE_SO"]))
    if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return abi.bind(C.CDLL(ORACLE_SO))


@pytest.fixture(scope="session")
def hip():
    """The product library; loading it needs no GPU, creating a sim does."""
    if os.environ.get("SWIMSIM_EMU_SO"):      # tools/emu: the kernel source compiled for the host against a wave64 lock-step emulator, so that
        return abi.bind(C.CDLL(os.environ["SWIMSIM_EMU_SO"]))   # the gpu-marked tests can be run (slowly) where there is no GPU; never the product path
    from consul_amd import lib
    return lib.load()
