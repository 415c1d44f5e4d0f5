"""The wire codec (include/swimsim_wire.hpp) under AddressSanitizer + UBSan against random and mutated packets / streams: a
decoder fed by a real, possibly remote memberlist node may refuse (DecodeError) but never read out of bounds, overflow, recurse
without limit or throw anything else."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_decoders_survive_hostile_bytes(tmp_path):
    exe = tmp_path / "fuzz_wire"
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-o", str(exe),
                    os.path.join(ROOT, "tests", "host", "fuzz_wire.cpp")], check=True)
    out = subprocess.run([str(exe), "60000"], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0 and "fuzz ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
