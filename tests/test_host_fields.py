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
