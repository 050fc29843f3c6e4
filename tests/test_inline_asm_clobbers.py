"""Source-level guard for the hand-written inline asm in csrc/: a statement that contains an SCC-writing SALU instruction
must declare the "scc" clobber, and one that names vcc explicitly must declare "vcc".  (A missing "scc" once let the
scheduler drop Goldilocks::mad_eps_fix between an s_add_u32 / s_addc_u32 address pair: a +2^32 address and a GPU memory
fault in gl::strided_kernel<4>.)  CPU only."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCC_WRITERS = re.compile(r"\bs_(or|and|xor|nor|nand|xnor|andn2|orn2|not|add|addc|sub|subb|lshl|lshr|ashr|cmp|bfe|abs|min|max|mul_i32x)_")


def asm_statements(text):
    i = 0
    while True:
        m = re.search(r"\basm\s*(volatile)?\s*\(", text[i:])
        if not m:
            return
        start = i + m.end()
        depth, j = 1, start
        while depth and j < len(text):
            depth += text[j] == "("
            depth -= text[j] == ")"
            j += 1
        yield text[start:j - 1]
        i = j


def _violations(text, path="<text>"):
    out, seen = [], 0
    if True:
        for stmt in asm_statements(text):
            body = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', stmt.split(":")[0])).replace("\\n", "\n").replace("\\t", " ")
            if not body.strip():
                continue  # empty optimisation barrier
            seen += 1
            clobbers = stmt.split(":")[3] if stmt.count(":") >= 3 else ""
            if SCC_WRITERS.search(body) and '"scc"' not in clobbers:
                out.append("%s: asm with an SCC-writing SALU op lacks the scc clobber:\n%s" % (path, stmt))
            if re.search(r"\bvcc\b", body) and '"vcc"' not in clobbers:
                out.append("%s: asm naming vcc lacks the vcc clobber:\n%s" % (path, stmt))
    return out, seen


def test_inline_asm_declares_scc_and_vcc_clobbers():
    total = 0
    for path in glob.glob(os.path.join(ROOT, "stark_rings_amd", "csrc", "*")):
        bad, seen = _violations(open(path).read(), path)
        assert not bad, "\n".join(bad)
        total += seen
    assert total >= 4  # mac2, mac1, the two mad_eps_fix statements
    # negative control: the guard must notice the clobber going missing
    fields = open(os.path.join(ROOT, "stark_rings_amd", "csrc", "fields.hpp")).read()
    assert ': "scc");' in fields
    assert _violations(fields.replace(': "scc");', ');'))[0]


def _sgpr_outputs_without_early_clobber(text, path="<text>"):
    """A multi-instruction statement whose SGPR output is written before its last "s" input is read must mark that output
    early-clobber ("=&s"): otherwise the register allocator may hand the output the registers of such an input (ADVICE r3:
    the generated Stark column statements write their carry pair in the first instruction and read 2^24 / 2^27 from SGPRs in
    later ones).  Conservative rule: more than one instruction + any "s" input => every "=s" output needs the '&'."""
    out = []
    for stmt in asm_statements(text):
        parts = stmt.split(":")
        if len(parts) < 3:
            continue
        body = "".join(re.findall(r'"((?:[^"\\]|\\.)*)"', parts[0])).replace("\\n", "\n")
        n_instr = len([ln for ln in body.split("\n") if ln.strip()])
        if n_instr < 2:
            continue
        if re.search(r'"s"\s*\(', parts[2]) and re.search(r'"=s"\s*\(', parts[1]):
            out.append("%s: SGPR output without early-clobber in a multi-instruction asm that reads SGPR inputs:\n%s" % (path, stmt))
    return out


def test_sgpr_outputs_of_multi_instruction_statements_are_early_clobber():
    checked = 0
    for path in glob.glob(os.path.join(ROOT, "stark_rings_amd", "csrc", "*")):
        text = open(path).read()
        bad = _sgpr_outputs_without_early_clobber(text, path)
        assert not bad, "\n".join(bad)
        checked += 1
    assert checked >= 5
    inc = open(os.path.join(ROOT, "stark_rings_amd", "csrc", "stark_mul_cols.inc")).read()
    assert '"=&s"(cy)' in inc
    assert _sgpr_outputs_without_early_clobber(inc.replace('"=&s"(cy)', '"=s"(cy)'))  # negative control


def test_sgpr_carry_reaches_its_reader_after_two_wait_states(tmp_path):
    """Goldilocks::mul hands the middle carry from a v_mad_u64_u32 (SGPR-pair carry-out) to a v_subb_co_u32 (borrow-in) in ANOTHER
    asm statement; gfx950 wants 2 wait states between a VALU write of an SGPR and a VALU read of it, and the compiler's hazard
    recogniser does not look inside inline asm.  Compile a kernel with sixteen independent products (room for the scheduler to
    reorder), and measure the distance in the generated ISA: every SGPR pair written as a carry-out by v_mad_u64_u32 and read as a
    borrow-in by v_subb_co_u32 must have two wait states (instructions, or s_nop N = N + 1) in between."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        import pytest
        pytest.skip("hipcc not available")
    src = tmp_path / "mul16.hip"
    src.write_text('#include "%s"\n'
                   "__global__ void mul16(uint64_t *x, const uint64_t *w) {\n"
                   "    uint64_t v[16];\n"
                   "#pragma unroll\n"
                   "    for (int i = 0; i < 16; i++) v[i] = x[threadIdx.x + 256 * i];\n"
                   "#pragma unroll\n"
                   "    for (int i = 0; i < 16; i++) v[i] = sr::Goldilocks::mul(v[i], w[16 * threadIdx.x + i]);\n"
                   "#pragma unroll\n"
                   "    for (int i = 0; i < 16; i++) x[threadIdx.x + 256 * i] = sr::Goldilocks::mul(v[i], v[(i + 1) & 15]);\n"
                   "}\n" % os.path.join(ROOT, "stark_rings_amd", "csrc", "fields.hpp"))
    out = tmp_path / "mul16.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out), str(src)])
    lines = []
    for ln in out.read_text().split("\n"):
        m = re.match(r"\s+([sv]_[a-z0-9_]+)\s*(.*?)(\s*;.*)?$", ln)
        if m:
            lines.append((m.group(1), m.group(2)))
    pairs = 0
    for i, (op, args) in enumerate(lines):
        if op != "v_mad_u64_u32":
            continue
        m = re.match(r"v\[\d+:\d+\],\s*(s\[\d+:\d+\])", args)
        if not m:
            continue  # carry-out to vcc: compiler-generated, the compiler's own hazard handling applies
        sg = m.group(1)
        waits = 0
        for op2, args2 in lines[i + 1:]:
            if op2.startswith("v_subb_co_u32") and args2.rstrip().endswith(sg):
                assert waits >= 2, "only %d wait state(s) between the write of %s and its use as a borrow-in" % (waits, sg)
                pairs += 1
                break
            if re.search(r"\b%s\b" % re.escape(sg), args2) and not op2.startswith("s_nop"):
                break  # the pair is re-used for something else first: not the pattern under test
            waits += int(args2.strip() or 0) + 1 if op2 == "s_nop" else 1
    assert pairs >= 32, "expected the 32 products' carries in the listing, found %d" % pairs
