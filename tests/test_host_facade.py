"""The compiled host side (include/swimsim_serf.hpp) under the reference-shaped acceptance tests in
tests/host/test_serf_facade.cpp: on CPU against the oracle library (checks the facade logic), and on
the GPU box against libswimsim.so (the product path end to end)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host", "test_serf_facade.cpp")


def build_and_run(tmp_path, libdir, libname):
    exe = str(tmp_path / "facade")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-o", exe, SRC, f"-L{libdir}", f"-l{libname}",
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
