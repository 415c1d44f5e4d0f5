"""The compiled host side (include/swimsim_serf.hpp) under the reference-shaped acceptance tests in
tests/host/test_serf_facade.cpp: on CPU against the oracle library (checks the facade logic), and on
the GPU box against libswimsim.so (the product path end to end)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_serf_facade.cpp")
WIRE = os.path.join(ROOT, "tests", "host", "test_wire.cpp")


def build_and_run(tmp_path, libdir, libname, src=SRC):
    exe = str(tmp_path / "facade")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, src, f"-L{libdir}", f"-l{libname}",
                    f"-Wl,-rpath,{libdir}"], check=True)
    return subprocess.run([exe], capture_output=True, text=True, timeout=600)


def test_serf_facade_logic_on_oracle(tmp_path, oracle):
    out = build_and_run(tmp_path, os.path.join(ROOT, "oracle", "_build"), "swim_oracle")
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout + out.stderr
    assert "backend oracle-c" in out.stdout


@pytest.mark.gpu
def test_serf_facade_on_hip(tmp_path, hip):
    out = build_and_run(tmp_path, os.path.join(ROOT, "consul_amd"), "swimsim")
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout + out.stderr
    assert "backend hip-gfx950" in out.stdout


def test_wire_codec_and_bridge_on_oracle(tmp_path, oracle):
    """include/swimsim_wire.hpp: memberlist's packet format (msgpack structs, compound, label, CRC) against hand-derived
    byte vectors, and an end-to-end pass through swim_transport_poll / swim_transport_write_to."""
    out = build_and_run(tmp_path, os.path.join(ROOT, "oracle", "_build"), "swim_oracle", WIRE)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
def test_wire_codec_and_bridge_on_hip(tmp_path, hip):
    out = build_and_run(tmp_path, os.path.join(ROOT, "consul_amd"), "swimsim", WIRE)
    assert out.returncode == 0 and "ALL PASSED" in out.stdout and "backend hip-gfx950" in out.stdout, out.stdout + out.stderr
