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


SELF_SRC = os.path.join(ROOT, "tools", "ubench", "repcheck_selftest.hip")
SELF_EXE = os.path.join(ROOT, "tools", "ubench", "repcheck_selftest")


@pytest.mark.gpu
def test_invariant_counters_fire_when_their_precondition_is_violated():
    """Test of the test behind tests/test_rep_invariants.py: compiled with -DSR_GL_CHECK_REPS, each counter of csrc/fields.hpp: repcheck
    fires for the violation it guards -- a canonical add handed p, a lazy sum that would wrap twice, a lazy difference that would
    borrow twice, a word >= p through st_result, a twiddle-1 butterfly of a phased stage fed p + 1, a nine-limb Stark sum that leaves
    int32, a Stark product whose operands could overflow its column accumulator -- and all stay at zero on legal operands
    (tools/ubench/repcheck_selftest.hip)."""
    if not os.path.exists(SELF_EXE) or any(os.path.getmtime(d) > os.path.getmtime(SELF_EXE) for d in [SELF_SRC, os.path.join(ROOT, "stark_rings_amd", "csrc", "stark_lazy.hpp")] + DEPS[1:]):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        if not os.path.exists(hipcc):
            hipcc = shutil.which("hipcc")
        if not hipcc:
            pytest.fail("hipcc not found: cannot build tools/ubench/repcheck_selftest")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-DSR_GL_CHECK_REPS", "-Wno-unused-value", "-o", SELF_EXE, SELF_SRC],
                       check=True, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    r = subprocess.run([SELF_EXE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, cwd=ROOT)
    out = r.stdout.decode()
    assert r.returncode == 0 and "repcheck_selftest: 0 wrong" in out, out[-2000:]
