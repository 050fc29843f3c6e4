"""Static budget of the headline kernels (no GPU): hipcc -S of tools/ubench/gl_isa.hip (the Goldilocks D = 2^16 kernels alone) and
a count of what the listing holds.  The step is bound by VALU issue / energy, so an instruction that creeps back in,
a spill or a scratch allocation is a performance regression the parity tests cannot see."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "ubench", "gl_isa.hip")
OUT = os.path.join(ROOT, "build_tmp", "gl_isa_budget.s")
DEPS = [SRC] + [os.path.join(ROOT, "stark_rings_amd", "csrc", f) for f in ("fields.hpp", "ntt_goldilocks.hpp")]

# kernel (mangled-name fragment) -> (max VALU instructions in the listing, max VGPRs)
BUDGET = {
    "rows256_kernelILi2E": (2830, 128),        # fused product: two forward 256-point transforms, slot product, inverse
    "rows256_kernelILi3E": (2080, 128),        # the same with the right operand already in NTT form
    "cols256_kernelILi0ELi4E": (1200, 128),    # forward column pass
    "cols256_kernelILi1ELi4E": (1235, 128),    # inverse column pass
    # the lane plans' column passes (twist factors kept in registers): compiled for three workgroups per CU (up to 168 VGPRs), the
    # allocator needs 120 / 126 -- within 128, so four still share a CU; compiled for four it spills
    "cols256_keep_kernelILi0E": (1300, 128),
    "cols256_keep_kernelILi1E": (1330, 128),
}


def _listing():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = shutil.which("hipcc")
    if not hipcc:
        pytest.skip("hipcc not found")
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in DEPS):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", OUT, SRC], check=True,
                       cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    return open(OUT).read()


def test_headline_kernels_stay_within_their_instruction_and_register_budget():
    s = _listing()
    seen = set()
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", s, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        for frag, (max_valu, max_vgpr) in BUDGET.items():
            if frag not in name:
                continue
            seen.add(frag)
            valu = sum(1 for line in body.split("\n") if re.match(r"\s+v_[a-z0-9_]+\s", line))
            assert valu <= max_valu, "%s: %d VALU instructions (budget %d)" % (name, valu, max_valu)
            meta = s[s.index(".amdhsa_kernel " + name):]
            meta = meta[:meta.index(".end_amdhsa_kernel")]
            vgpr = int(re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", meta).group(1))
            scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", meta).group(1))
            assert vgpr <= max_vgpr, "%s: %d VGPRs (budget %d: four waves per SIMD)" % (name, vgpr, max_vgpr)
            assert scratch == 0, "%s: %d bytes of scratch (spills)" % (name, scratch)
    assert seen == set(BUDGET), "kernels not found in the listing: %s" % (set(BUDGET) - seen)
