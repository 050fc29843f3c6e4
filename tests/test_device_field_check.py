"""Device against host build of the SAME arithmetic source (fields.hpp, ntt_goldilocks.hpp): tools/ubench/field_check.hip runs
Goldilocks add / sub / mul / folds, the Fq3 slot product, the lazy decimation-in-time legs (any 64-bit `a`, canonical shift product)
and the 16-point networks built from them on the GPU and compares every result with the host compilation -- device-only code
generation problems (inline asm, EXEC masks, wait states) show up here before they show up as a wrong transform."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "field_check.hip")
EXE = os.path.join(ROOT, "tools", "ubench", "field_check")
DEPS = [SRC, os.path.join(ROOT, "stark_rings_amd", "csrc", "fields.hpp"), os.path.join(ROOT, "stark_rings_amd", "csrc", "ntt_goldilocks.hpp")]


def _build():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = shutil.which("hipcc")
    if not hipcc:
        pytest.fail("hipcc not found: cannot build tools/ubench/field_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-o", EXE, SRC], check=True, cwd=ROOT,
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)


@pytest.mark.gpu
def test_device_arithmetic_equals_host_build_of_the_same_source():
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in DEPS):
        _build()
    r = subprocess.run([EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "field_check: 0 mismatches" in out, out[-2000:]
