"""Builds and runs tests/cpp/test_host_api.cpp (the C++ mirror of the reference interface, include/stark_rings.hpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "test_host_api")


def _build():
    import oracle_lib

    oracle_lib.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_host_api.cpp")
    cmd = ["g++", "-std=c++17", "-O2", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", BIN, src,
           "-L" + os.path.join(ROOT, "stark_rings_amd"), "-lstarkrings_hip",
           "-L" + os.path.join(ROOT, "oracle"), "-lsr_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "stark_rings_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-lpthread"]
    subprocess.check_call(cmd, cwd=ROOT)


def test_cpp_host_mirror_compiles():
    """CPU: the header-only mirror and its test compile and link against the C ABI."""
    _build()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_host_mirror_parity():
    _build()
    r = subprocess.run([BIN], cwd=ROOT, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all ok" in r.stdout
