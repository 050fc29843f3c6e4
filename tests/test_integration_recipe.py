"""INTEGRATION.md section 3 is a patch against the reference crate that must not add a second impl of a trait a type already has
(error E0119; stable Rust has no specialisation).  No Rust toolchain exists in this image, so the check is by reading -- and this test
makes the reading reproducible: every reference line the recipe says it replaces or edits is where the recipe says it is, and the
traits it hooks have exactly the impls the recipe lists.  Skipped where /root/reference is absent (the GPU box)."""
import os
import re

import pytest

REF = "/root/reference/crates/ring/src"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not present on this machine")


def _line(rel, n):
    return open(os.path.join(REF, rel)).read().split("\n")[n - 1]


def _grep(pattern):
    hits = []
    for d, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".rs"):
                p = os.path.join(d, f)
                for i, ln in enumerate(open(p).read().split("\n"), 1):
                    if re.search(pattern, ln):
                        hits.append((os.path.relpath(p, REF), i, ln.strip()))
    return hits


def test_the_lines_the_recipe_replaces_are_where_it_says():
    # the four macro invocations the patch puts behind #[cfg(not(feature = "hip"))]
    for rel, n, cfg in (("cyclotomic_ring/models/goldilocks/mod.rs", 151, "GoldilocksRingConfig"),
                        ("cyclotomic_ring/models/babybear/mod.rs", 163, "BabyBearRingConfig"),
                        ("cyclotomic_ring/models/stark_prime/mod.rs", 97, "StarkRingConfig"),
                        ("cyclotomic_ring/models/frog_ring/mod.rs", 135, "FrogRingConfig")):
        assert _line(rel, n).strip() == "impl_crt_icrt_for_a_ring!(RqNTT, RqPoly, %s);" % cfg, (rel, n)
    # the impl blocks it edits in place
    assert _line("cyclotomic_ring/models/goldilocks/mod.rs", 69).startswith("impl CyclotomicConfig<1> for GoldilocksRingConfig")
    assert _line("cyclotomic_ring/models/babybear/mod.rs", 81).startswith("impl CyclotomicConfig<1> for BabyBearRingConfig")
    assert _line("cyclotomic_ring/models/frog_ring/mod.rs", 72).startswith("impl CyclotomicConfig<1> for FrogRingConfig")
    assert _line("cyclotomic_ring/models/stark_prime/mod.rs", 34).startswith("impl CyclotomicConfig<4> for StarkRingConfig")
    assert "Decompose" in _line("cyclotomic_ring/coeff_form.rs", 588) and "for CyclotomicPolyRingGeneral" in _line("cyclotomic_ring/coeff_form.rs", 589)
    assert _line("balanced_decomposition/mod.rs", 21).startswith("pub trait Decompose: Ring")
    assert _line("balanced_decomposition/mod.rs", 163).startswith("impl<R: Decompose> GadgetDecompose for &[R]")
    assert _line("cyclotomic_ring/ring_config.rs", 11).startswith("pub trait CyclotomicConfig<const N: usize>")
    assert "fn elementwise_crt" in _line("cyclotomic_ring/crt.rs", 10) and "fn elementwise_icrt" in _line("cyclotomic_ring/crt.rs", 34)


def test_no_trait_the_recipe_touches_has_an_impl_it_would_collide_with():
    # CRT / ICRT: implemented ONLY inside the macro (crt.rs) -- replacing the macro's expansion leaves exactly one impl per type
    crt = _grep(r"impl\b.*\bI?CRT for\b")
    assert crt and all(rel == "cyclotomic_ring/crt.rs" for rel, _, _ in crt), crt
    # Decompose: the blanket impl for ConvertibleRing and the generic one for the coefficient-form ring; nothing per model
    dec = [(rel, n) for rel, n, ln in _grep(r"impl\b.*[^a-zA-Z]Decompose\s*(for\b|$)") if "Gadget" not in ln]
    assert sorted(dec) == [("balanced_decomposition/convertible_ring.rs", 43), ("cyclotomic_ring/coeff_form.rs", 588)], dec
    # GadgetDecompose: blanket impls only (the recipe adds a call inside the one for &[R], no new impl)
    gd = _grep(r"impl<R: Decompose> GadgetDecompose for")
    assert sorted(n for rel, n, _ in gd if rel == "balanced_decomposition/mod.rs") == [163, 192, 208, 276, 311]
    # CyclotomicConfig: four model impls, all edited in place
    cc = [rel for rel, _, _ in _grep(r"^impl CyclotomicConfig<")]
    assert sorted(cc) == sorted("cyclotomic_ring/models/%s/mod.rs" % m for m in ("goldilocks", "babybear", "stark_prime", "frog_ring"))


def test_the_recipe_itself_declares_no_impl_of_an_existing_trait_for_an_existing_type():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 3. The patch"):text.index("## 4. Linking")]
    added = [ln[1:].strip() for ln in sec.split("\n") if ln.startswith("+") and not ln.startswith("+++")]
    heads = [ln for ln in added if re.match(r"impl\b", ln)]
    # the only impl headers on '+' lines are the two inside the replacement macro (generic over $icrt / $crt)
    assert heads == ["impl CRT for $icrt {", "impl ICRT for $crt {"], heads
    # outside the diff, the one new impl is for a NEW type
    rust_blocks = re.findall(r"```rust\n(.*?)```", sec, flags=re.S)   # (context lines of the ```diff blocks are the reference's own text)
    free = [ln.strip() for blk in rust_blocks for ln in blk.split("\n") if re.match(r"\s*impl\b", ln)]
    assert free and all("GoldilocksPow2Config" in ln for ln in free), free
