"""CPU tests of the product's host-visible logic (no GPU):
 * the C-ABI library loads and exports every symbol include/stark_rings_hip.h declares;
 * fields.hpp arithmetic (the same source the kernels compile) through sr_selftest_field_op
   against the pure-Python model;
 * compute entry points fail loudly without a device (no CPU fallback).
"""
import ctypes
import os
import random
import re

import numpy as np
import pytest

import pyref as P
from stark_rings_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = ["goldilocks", "babybear", "stark", "frog"]
KAPPA_BITS = {"goldilocks": 0, "babybear": 32, "stark": 256, "frog": 64}


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "stark_rings_hip.h")).read()
    declared = set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), "library does not export %s" % name
    assert declared == set(_lib.SYMBOLS), "ctypes table and header disagree: %s" % (declared ^ set(_lib.SYMBOLS))


def _op(field, op, a, b, limbs):
    lib = _lib.load()
    mask = (1 << 64) - 1
    A = (ctypes.c_uint64 * 4)(*[(a >> (64 * i)) & mask for i in range(4)])
    B = (ctypes.c_uint64 * 4)(*[(b >> (64 * i)) & mask for i in range(4)])
    O = (ctypes.c_uint64 * 4)()
    assert lib.sr_selftest_field_op(field, op, A, B, O) == 0, _lib.last_error()
    return sum(int(O[i]) << (64 * i) for i in range(limbs))


@pytest.mark.parametrize("fid,name", list(enumerate(NAMES)))
def test_field_ops_match_python_model(fid, name):
    p, _, limbs = P.PRIMES[name]
    rng = random.Random(11 + fid)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 2**32 % p, (2**32 - 1) % p, (2**63) % p]
    vals = edge + [rng.randrange(p) for _ in range(300)]
    Rb_inv = pow(pow(2, 64 * limbs, p), -1, p)
    kap_inv = pow(pow(2, KAPPA_BITS[name], p), -1, p)
    for i in range(0, len(vals) - 1):
        a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
        assert _op(fid, 0, a, b, limbs) == (a + b) % p
        assert _op(fid, 1, a, b, limbs) == (a - b) % p
        assert _op(fid, 2, a, b, limbs) == a * b * Rb_inv % p       # reference Fp product on in-memory images
        assert _op(fid, 3, a, b, limbs) == a * b * kap_inv % p      # twiddle product
    for x in (0, 1, 7, 2**40, 2**64 - 1):
        assert _op(fid, 4, x, 0, limbs) == (x % p) * pow(2, KAPPA_BITS[name], p) % p


def test_goldilocks_mul_worst_cases():
    # products whose 128-bit image stresses every carry/borrow branch of reduce128
    p = P.GOLDILOCKS_P
    specials = [p - 1, p - 2, 2**32, 2**32 - 1, 2**32 + 1, 2**63, 2**64 - 2**33, 0xFFFFFFFF00000000, 0xFFFFFFFE, 1]
    for a in specials:
        for b in specials:
            a_, b_ = a % p, b % p
            assert _op(0, 3, a_, b_, 1) == a_ * b_ % p
            assert _op(0, 2, a_, b_, 1) == a_ * b_ * pow(2**64, -1, p) % p


def test_no_device_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from stark_rings_amd import CyclotomicRing, RingError

    with pytest.raises(RingError, match="no HIP device|no CPU fallback"):
        CyclotomicRing("goldilocks", 10)


def test_stark_lazy_limb_arithmetic_matches_python_model():
    """stark_lazy.hpp (nine signed 28-bit limbs, lazy carries, R = 2^280) through the selftest hook, field id 4: sums,
    differences, the Montgomery product, the table form, a chain of six uncarried additions / subtractions feeding both
    product forms, and repeated doubling with weak reduction -- every result leaves through the canonicalising store."""
    p = P.STARK_P
    rng = random.Random(41)
    r_inv = pow(pow(2, 280, p), -1, p)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 2**28 - 1, 2**28, 2**224, 2**251, 2**251 - 1, 17 * 2**192, (1 << 250) + 12345,
            sum(((1 << 28) - 1) << (28 * i) for i in range(9)) % p]
    vals = edge + [rng.randrange(p) for _ in range(400)]
    for i in range(len(vals)):
        a, b = vals[i], vals[(i * 5 + 2) % len(vals)]
        assert _op(4, 0, a, b, 4) == (a + b) % p
        assert _op(4, 1, a, b, 4) == (a - b) % p
        assert _op(4, 3, a, b, 4) == a * b * r_inv % p
        s, d = (a + 6 * b), (a - 6 * b)
        assert _op(4, 5, a, b, 4) == (s * b * r_inv + d * s * r_inv) % p
        assert _op(4, 6, a, b, 4) == 256 * a % p
    for x in (0, 1, 7, 2**40, 2**64 - 1):
        assert _op(4, 4, x, 0, 4) == x * pow(2, 280, p) % p
    # the slow path of canonical(): a borrow out of limb 0 after the fold (value still >= 0), and values that fold to < 0
    h = 2**250 + 17 * 2**192
    for a, b in ((h, h), (2**251, 0), (2**251 + 2**28, 0), (0, 1), (1, 2**251), (2**251 + 2**56, 2**251 + 2**84), (p - 1, p - 1)):
        assert _op(4, 0, a, b, 4) == (a + b) % p
        assert _op(4, 1, a, b, 4) == (a - b) % p
        assert _op(4, 1, b, a, 4) == (b - a) % p
    # canonical() on signed lazy states around every boundary it distinguishes (multiples of 2^251, of p, limb borders)
    bnd = sorted({v % p for base in (0, 2**251, p, 2**28, 2**196, 17 * 2**192, 2**224, (p - 1) // 2, (p + 1) // 3, (2 * p) // 5,
                                     (p + 2**251) // 7, 2**251 // 3, 2**251 // 5)
                  for v in (base - 2, base - 1, base, base + 1, base + 2, base + 2**28, base - 2**28)})
    for a in bnd:
        for b in bnd:
            assert _op(4, 7, a, b, 4) == (3 * a - 5 * b) % p
            assert _op(4, 8, a, b, 4) == (7 * a - 2 * b) % p


def test_goldilocks_shift_products_all_exponents():
    """gl::mul_pow2<E> (ntt_goldilocks.hpp), every E in [1, 95]: the three forms (left shift + fold for E < 64, right shift for
    64 <= E < 96) against Python integers on boundary values and random canonical inputs."""
    import random

    lib = _lib.load()
    p = 2**64 - 2**32 + 1
    rnd = random.Random(7)
    xs = [0, 1, 2, p - 1, p - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**63, 2**63 - 1, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF,
          0xFFFFFFFE00000001, 0x8000000000000001, (p - 1) // 2, (p + 1) // 2] + [rnd.randrange(p) for _ in range(48)]
    A = (ctypes.c_uint64 * 4)()
    B = (ctypes.c_uint64 * 4)()
    out = (ctypes.c_uint64 * 4)()
    for e in range(1, 96):
        for x in xs:
            A[0], B[0] = x, e
            assert lib.sr_selftest_field_op(0, 5, A, B, out) == 0
            assert out[0] == (x << e) % p, (hex(x), e)


def test_goldilocks_products_take_any_64_bit_representative():
    """The lazy butterflies of the tuned path hand Goldilocks::mul and gl::mul_pow2<E> arbitrary 64-bit representatives (p, p + 1,
    2^64 - 1 ...), not field elements: both must return the CANONICAL value of the product for every such operand (VERDICT r3 #1b).
    Host build of the same source against Python integers; tools/ubench/field_check.hip repeats it device against host."""
    import random

    lib = _lib.load()
    p = 2**64 - 2**32 + 1
    rnd = random.Random(9)
    raw = [0, 1, p - 1, p, p + 1, p + 2**32 - 2, 2**64 - 1, 2**64 - 2, 2**64 - 2**32, 2**64 - 2**32 - 1, 2**32 - 1, 2**32, 2**63, 2**63 - 1,
           0x8000000080000000, 0xFFFFFFFF80000000] + [rnd.randrange(2**64) for _ in range(40)]
    A = (ctypes.c_uint64 * 4)()
    B = (ctypes.c_uint64 * 4)()
    out = (ctypes.c_uint64 * 4)()
    for x in raw:
        for y in raw:
            A[0], B[0] = x, y
            assert lib.sr_selftest_field_op(0, 3, A, B, out) == 0          # Goldilocks::mul (mul_tw)
            assert out[0] == x * y % p, (hex(x), hex(y))
        for e in range(1, 96):
            A[0], B[0] = x, e
            assert lib.sr_selftest_field_op(0, 5, A, B, out) == 0
            assert out[0] == (x << e) % p, (hex(x), e)
