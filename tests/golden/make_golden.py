#!/usr/bin/env python3
"""Extract the reference's literal known-answer vectors into JSON fixtures.

Run in the build container only (``/root/reference`` does not exist on the GPU
box).  The output ``tests/golden/reference_kats.json`` is DATA: the decimal
literals the reference's own unit tests compare against (inputs and expected
outputs) and the constant tables those tests check.  No reference source text is
kept.  Every entry records the file:line range it was read from.

Sources:
  goldilocks/ntt.rs  test_crt / test_crt2 / test_icrt / test_icrt_2, ROOTS_OF_UNITY_24, KAPPA, EIGHT_INV, FOUR_INV
  stark_prime/ntt.rs test_crt / test_crt2 / test_icrt / test_icrt_2, ROOTS_OF_UNITY_32, SIXTEEN_INV*
  babybear/ntt.rs    test_babybear_icrt_hardcoded, ROOTS_OF_UNITY_24, KAPPA, EIGHT_INV, FOUR_INV
  stark_prime/decomposition.rs  test_stark_prime_decomposition (balanced digits of one Fq, basis 2^16)
  balanced_decomposition/mod.rs test_gadget_decompose / test_gadget_recompose / test_sparse_matrix_gadget_decompose and the
                                parameters of the property tests (Q, D, BASIS_TEST_RANGE)
  monomial.rs        test_monomial_range_check (which scalars pass psi_range_check on the frog ring) and test_monomial_ops
"""
import json
import os
import re
import sys

REF = os.environ.get("SR_REFERENCE", "/root/reference")
MODELS = os.path.join(REF, "crates/ring/src/cyclotomic_ring/models")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")

LIT = re.compile(
    r'MontFp!\("(\d+)"\)|Fq::from\((\d+)\)|Fq::new\(BigInt\(\[(\d+)(?:u64)?\]\)\)'
    r"|Fq::(zero)\(\)|Fq::(one)\(\)|Fq::(ZERO)|Fq::(ONE)"
)


def lit_values(text):
    vals = []
    for m in LIT.finditer(text):
        a, b, c, z, o, zz, oo = m.groups()
        if a is not None:
            vals.append(int(a))
        elif b is not None:
            vals.append(int(b))
        elif c is not None:
            vals.append(int(c))
        elif z or zz:
            vals.append(0)
        else:
            vals.append(1)
    return vals


def read(path):
    with open(path) as f:
        return f.read().split("\n")


def fn_span(lines, name):
    """Return (first, last) 1-based line numbers of ``fn name`` body."""
    start = None
    for i, l in enumerate(lines):
        if re.search(r"\bfn %s\(" % re.escape(name), l):
            start = i
            break
    if start is None:
        raise KeyError(name)
    depth = 0
    seen = False
    for j in range(start, len(lines)):
        depth += lines[j].count("{") - lines[j].count("}")
        if "{" in lines[j]:
            seen = True
        if seen and depth == 0:
            return start + 1, j + 1
    raise ValueError("unterminated fn " + name)


def vec_blocks(lines, first, last):
    """All ``vec![ ... ];`` literal blocks inside the span, in order."""
    text = "\n".join(lines[first - 1 : last])
    blocks = []
    for m in re.finditer(r"vec!\[(.*?)\];", text, flags=re.S):
        blocks.append(lit_values(m.group(1)))
    return blocks


def const_table(lines, name):
    for i, l in enumerate(lines):
        if re.search(r"const %s\b" % name, l):
            depth = 0
            buf = []
            for j in range(i, len(lines)):
                buf.append(lines[j].split("//")[0])
                depth += lines[j].count("[") - lines[j].count("]")
                if ";" in lines[j] and depth <= 0:
                    body = "\n".join(buf).split("=", 1)[1]
                    return lit_values(body), (i + 1, j + 1)
    raise KeyError(name)


def pad(v, n):
    return v + [0] * (n - len(v))


def S(vals):
    return [str(v) for v in vals]


def main():
    if not os.path.isdir(MODELS):
        sys.exit("reference not present at %s (fixtures are pre-generated; nothing to do)" % REF)
    out = {"_about": "literal KAT vectors of NethermindEth/stark-rings unit tests; standard-form decimal integers",
           "rings": {}}

    # ---------------- goldilocks (X^24 - X^12 + 1) ----------------
    rel = "goldilocks/ntt.rs"
    L = read(os.path.join(MODELS, rel))
    g = {"modulus": str(2**64 - 2**32 + 1), "degree": 24, "source": rel, "kats": []}
    roots, span = const_table(L, "ROOTS_OF_UNITY_24")
    g["roots_of_unity_24"] = {"values": S(roots), "lines": span}
    for cname in ("KAPPA", "EIGHT_INV", "FOUR_INV"):
        v, span = const_table(L, cname)
        g[cname.lower()] = {"value": str(v[0]), "lines": span}
    # test_crt / test_crt2: input poly -> expected residues BEFORE homogenize
    # (the tests call dehomogenize_fq3 on the crt output before comparing)
    for name in ("test_crt", "test_crt2"):
        a, b = fn_span(L, name)
        blocks = vec_blocks(L, a, b)
        g["kats"].append({"name": name, "kind": "crt_then_dehomogenize", "lines": [a, b],
                          "coeffs": S(pad(blocks[0], 24)), "residues": S(blocks[1])})
    # test_icrt / test_icrt_2: residues --homogenize--> icrt -> coefficients
    for name in ("test_icrt", "test_icrt_2"):
        a, b = fn_span(L, name)
        blocks = vec_blocks(L, a, b)
        g["kats"].append({"name": name, "kind": "homogenize_then_icrt", "lines": [a, b],
                          "coeffs": S(pad(blocks[0], 24)), "residues": S(blocks[1])})
    out["rings"]["goldilocks24"] = g

    # ---------------- stark prime (X^16 + 1) ----------------
    rel = "stark_prime/ntt.rs"
    L = read(os.path.join(MODELS, rel))
    s = {"modulus": str(2**251 + 17 * 2**192 + 1), "degree": 16, "source": rel, "kats": []}
    roots, span = const_table(L, "ROOTS_OF_UNITY_32")
    s["roots_of_unity_32"] = {"values": S(roots), "lines": span}
    for cname in ("SIXTEEN_INV", "SIXTEEN_INV_TIMES_ROOT_OF_UNITY_32_24"):
        v, span = const_table(L, cname)
        s[cname.lower()] = {"value": str(v[0]), "lines": span}
    for name in ("test_crt", "test_crt2"):
        a, b = fn_span(L, name)
        blocks = vec_blocks(L, a, b)
        s["kats"].append({"name": name, "kind": "crt", "lines": [a, b],
                          "coeffs": S(pad(blocks[0], 16)), "evals": S(blocks[1])})
    for name in ("test_icrt", "test_icrt_2"):
        a, b = fn_span(L, name)
        blocks = vec_blocks(L, a, b)
        s["kats"].append({"name": name, "kind": "icrt", "lines": [a, b],
                          "coeffs": S(pad(blocks[0], 16)), "evals": S(blocks[1])})
    out["rings"]["stark16"] = s

    # ---------------- babybear (X^72 - X^36 + 1) ----------------
    rel = "babybear/ntt.rs"
    L = read(os.path.join(MODELS, rel))
    bb = {"modulus": "2013265921", "degree": 72, "source": rel, "kats": []}
    roots, span = const_table(L, "ROOTS_OF_UNITY_24")
    bb["roots_of_unity_24"] = {"values": S(roots), "lines": span}
    for cname in ("KAPPA", "EIGHT_INV", "FOUR_INV"):
        v, span = const_table(L, cname)
        bb[cname.lower()] = {"value": str(v[0]), "lines": span}
    a, b = fn_span(L, "test_babybear_icrt_hardcoded")
    text = "\n".join(L[a - 1 : b])
    ntt = [0] * 72
    exp = [0] * 72
    for m in re.finditer(r'initial_ntt\[(\d+)\] = MontFp!\("(\d+)"\)', text):
        ntt[int(m.group(1))] = int(m.group(2))
    for m in re.finditer(r'expected\[(\d+)\] = MontFp!\("(\d+)"\)', text):
        exp[int(m.group(1))] = int(m.group(2))
    bb["kats"].append({"name": "test_babybear_icrt_hardcoded", "kind": "homogenize_then_icrt",
                       "lines": [a, b], "residues": S(ntt), "coeffs": S(exp)})
    out["rings"]["babybear72"] = bb

    # ---------------- frog (X^16 + 1 over a 64-bit prime; "next" row) ----------------
    rel = "frog_ring/ntt.rs"
    L = read(os.path.join(MODELS, rel))
    fr = {"modulus": "15912092521325583641", "degree": 16, "source": rel, "kats": []}
    roots, span = const_table(L, "ROOTS_OF_UNITY_8")
    fr["roots_of_unity_8"] = {"values": S(roots), "lines": span}
    for name, kind in (("test_crt", "crt_then_dehomogenize"), ("test_crt2", "crt_then_dehomogenize"),
                       ("test_icrt", "homogenize_then_icrt"), ("test_icrt_2", "homogenize_then_icrt")):
        a, b = fn_span(L, name)
        blocks = vec_blocks(L, a, b)
        fr["kats"].append({"name": name, "kind": kind, "lines": [a, b],
                           "coeffs": S(pad(blocks[0], 16)), "residues": S(blocks[1])})
    out["rings"]["frog16"] = fr

    # ---------------- balanced decomposition ("next" row 2) ----------------
    SIGNED = re.compile(r"(-?)\s*Fq::(?:from\((-?\d+)\)|(zero)\(\))")

    def signed_values(text):
        vals = []
        for m in SIGNED.finditer(text):
            neg, num, zero = m.groups()
            v = 0 if zero else int(num)
            vals.append(-v if neg else v)
        return vals

    dec = {}
    rel = "stark_prime/decomposition.rs"
    L = read(os.path.join(MODELS, rel))
    a, b = fn_span(L, "test_stark_prime_decomposition")
    text = "\n".join(L[a - 1:b])
    x = re.search(r'MontFp!\("(\d+)"\)', text).group(1)
    call = re.search(r"decompose\((.+?),\s*(\d+)\)", text)
    basis = eval(call.group(1), {"__builtins__": {}})  # "1 << 16"
    digits = signed_values(text[text.index("vec!["):])
    dec["stark_prime_fq"] = {"source": rel, "lines": [a, b], "x": x, "basis": basis, "padding": int(call.group(2)),
                             "digits": [str(v) for v in digits]}
    rel = "balanced_decomposition/mod.rs"
    L = read(os.path.join(REF, "crates/ring/src", rel))
    a, b = fn_span(L, "test_gadget_decompose")
    text = "\n".join(L[a - 1:b])
    call = re.search(r"gadget_decompose\((\d+),\s*(\d+)\)", text)
    halves = text.split("let decomposed")
    ring_deg = int(re.search(r"\(0\.\.(\d+)\)", text).group(1))
    dec["goldilocks24_gadget"] = {"source": rel, "lines": [a, b], "degree": ring_deg, "basis": int(call.group(1)),
                                  "padding": int(call.group(2)),
                                  "input_coefficient_values": signed_values(halves[0]),       # every coefficient of element e
                                  "expected_coefficient_values": signed_values(halves[1])}    # ... of output element e * k + j
    a, b = fn_span(L, "test_sparse_matrix_gadget_decompose")
    text = "\n".join(L[a - 1:b])
    halves = text.split("let decomposed")
    call = re.search(r"gadget_decompose\((\d+),\s*(\d+)\)", text)
    pair = re.compile(r"Fq::from\((-?\d+)\)\)\.collect::<Vec<Fq>>\(\)\),\s*(\d+),", re.S)
    dec["goldilocks24_sparse_gadget"] = {"source": rel, "lines": [a, b], "basis": int(call.group(1)), "padding": int(call.group(2)),
                                         "input_entries": [[int(v), int(c)] for v, c in pair.findall(halves[0])],
                                         "expected_entries": [[int(v), int(c)] for v, c in pair.findall(halves[1])]}
    consts = "\n".join(L)
    dec["property_test_parameters"] = {
        "source": rel, "D": int(re.search(r"const D: usize = (\d+);", consts).group(1)),
        "Q": int(re.search(r"const Q: u64 = (\d+);", consts).group(1)),
        "bases": [int(v) for v in re.search(r"BASIS_TEST_RANGE: \[u128; \d+\] = \[([^\]]+)\]", consts).group(1).split(",")],
        "scalar_padding": 32, "vector_padding": 16}
    out["decomposition"] = dec

    # ---------------- monomial helpers ("next" row 4): crates/ring/src/monomial.rs tests ----------------
    rel = "monomial.rs"
    L = read(os.path.join(REF, "crates/ring/src", rel))
    a, b = fn_span(L, "test_monomial_range_check")
    text = "\n".join(L[a - 1:b])
    names = {}
    for m in re.finditer(r"let (\w+) = <RqPoly as PolyRing>::BaseRing::from\((\d+)u128\)( - (\w+))?;", text):
        names[m.group(1)] = int(m.group(2)) - (names[m.group(4)] if m.group(4) else 0)
    checks = [[names[m.group(1)], m.group(2) == "ok"]
              for m in re.finditer(r"psi_range_check::<RqPoly>\((\w+)\)\.is_(ok|err)\(\)", text)]
    a2, b2 = fn_span(L, "test_monomial_ops")
    ops = "\n".join(L[a2 - 1:b2])
    mono = {m.group(1): int(m.group(2)) for m in re.finditer(r"let (x\d+) = monomial::<RqPoly>\((\d+),", ops)}
    out["monomial"] = {"source": rel, "ring": "frog16", "range_check": {"lines": [a, b], "cases": checks},
                       "ops": {"lines": [a2, b2], "monomial_degrees": mono,
                               "facts": ["0 + 1 = 1", "X^2 + X^2 has coefficient 2 at index 2", "X^2 * X^15 has coefficient -1 at index 1"]}}

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    n = sum(len(r["kats"]) for r in out["rings"].values())
    print("wrote %s (%d KATs)" % (OUT, n))


if __name__ == "__main__":
    main()
