"""The C-ABI boundary: both shared libraries export every symbol include/swimsim.h declares, and
the ctypes mirror (consul_amd/abi.py) has the same struct layouts as the C header.  No GPU needed:
nothing here computes."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from consul_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "swimsim.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(swim_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    assert declared_symbols() == sorted(abi.PROTOTYPES)


@pytest.mark.parametrize("which", ["hip", "oracle"])
def test_library_exports_every_declared_symbol(which, hip, oracle):
    lib = hip if which == "hip" else oracle
    for name in declared_symbols():
        assert hasattr(lib, name), f"{which} library lacks {name}"
    assert lib.swim_backend() == (b"hip-gfx950" if which == "hip" else b"oracle-c")


def test_product_loader_refuses_anything_but_the_hip_library(tmp_path, monkeypatch):
    from consul_amd import lib
    monkeypatch.setattr(lib, "_cdll", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "libswimsim.so"))
    with pytest.raises(ImportError):
        lib.load()                                   # missing: no CPU fallback
    oracle_so = os.path.join(ROOT, "oracle", "_build", "libswim_oracle.so")
    os.symlink(oracle_so, tmp_path / "libswimsim.so")
    with pytest.raises(ImportError):
        lib.load()                                   # wrong backend: refused


def test_create_without_a_gpu_fails_loudly(hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = abi.Config()
    assert hip.swim_config_preset(C.byref(cfg), abi.PRESET_LAN) == 0
    h = abi.SimP()
    assert hip.swim_create(C.byref(cfg), C.byref(h)) == abi.ENODEV


def test_struct_layouts_match_the_c_header(tmp_path):
    structs = {"swim_config": abi.Config, "swim_derived": abi.Derived, "swim_member": abi.Member,
               "swim_event": abi.Event, "swim_rumour": abi.Rumour, "swim_node_info": abi.NodeInfo,
               "swim_census": abi.Census, "swim_edge": abi.Edge, "swim_stats_t": abi.Stats,
               "swim_kernel_time": abi.KernelTime}
    probes = {"swim_config": ["seed", "msg_len", "flags", "shard_rank"], "swim_node_info": ["alive", "queue"],
              "swim_census": ["all_current_ms"], "swim_stats_t": ["user_events_stale", "msgs_applied"],
              "swim_derived": ["packet_budget"], "swim_kernel_time": ["total_ms"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for n in structs:
        lines.append(f'printf("{n} %zu\\n", sizeof({n}));')
        for f in probes.get(n, []):
            lines.append(f'printf("{n}.{f} %zu\\n", offsetof({n}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, cls in structs.items():
        assert int(out[n]) == C.sizeof(cls), n
        for f in probes.get(n, []):
            assert int(out[f"{n}.{f}"]) == getattr(cls, f).offset, f"{n}.{f}"
