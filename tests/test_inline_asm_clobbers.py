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
