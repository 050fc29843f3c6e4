"""Monomial helpers of the reference (crates/ring/src/monomial.rs:17-93) over a CyclotomicRing, in coefficient form.

  monomial / zero_monomial / unit_monomial   monomial.rs:17-31
  psi          sum_{i in [1, d/2)} i (X^i - X^(d-i))               monomial.rs:36-49
  exp          sign(a) X^a as a UNIT monomial (X^(d - |a|) for a < 0)  monomial.rs:56-66
  exp_signed   monomial(|a|, 1) * sign(a)                            monomial.rs:72-78
  psi_range_check   ct(exp(a) * psi) == a                            monomial.rs:84-93
  Zq::center / sign                                                  crates/ring/src/ring.rs:160-181

Scalars are Python integers in standard form (the reference's `BaseRing::from(u128)` side); ring elements are numpy uint64
words in the reference's memory image.  Integers enter and leave that image through the device codec (rings.serialize /
deserialize); the product inside the range check is the ring product of the C ABI (`RqPoly * RqPoly` -> sr_ring_mul_batch).
Where the reference panics (monomial index past the dimension) or returns an Err, RingError is raised.
"""
import numpy as np

from .rings import RingError


def from_ints(ring, values):
    """standard-form integers -> coefficients in the memory image (len(values) must be a multiple of D)"""
    w, p = ring.wire_coeff_bytes, ring.modulus
    raw = b"".join((int(v) % p).to_bytes(w, "little") for v in values)
    return ring.deserialize(np.frombuffer(raw, dtype=np.uint8))


def to_ints(ring, words):
    w = ring.wire_coeff_bytes
    raw = ring.serialize(words).tobytes()
    return [int.from_bytes(raw[i:i + w], "little") for i in range(0, len(raw), w)]


def center(ring, a):
    p = ring.modulus
    a %= p
    return p - a if a > (p - 1) // 2 else a


def sign(ring, a):
    p = ring.modulus
    return p - 1 if a % p > (p - 1) // 2 else 1


def _monomial_ints(ring, i, coeff):
    if not 0 <= i < ring.degree:
        raise RingError("monomial: index %d is outside the ring's dimension %d" % (i, ring.degree))
    m = [0] * ring.degree
    m[i] = coeff
    return m


def monomial(ring, i, coeff):
    return from_ints(ring, _monomial_ints(ring, i, coeff))


def zero_monomial(ring):
    return from_ints(ring, [0] * ring.degree)


def unit_monomial(ring, i):
    return monomial(ring, i, 1)


def _psi_ints(ring):
    d, p = ring.degree, ring.modulus
    out = [0] * d
    for i in range(1, d // 2):
        out[i] = (out[i] + i) % p
        out[d - i] = (out[d - i] - i) % p
    return out


def psi(ring):
    return from_ints(ring, _psi_ints(ring))


def _exp_index(ring, a):
    c = center(ring, a)
    if c >= 1 << 64:
        raise RingError("exp: ConversionError::ToInteger")
    return c if sign(ring, a) == 1 else ring.degree - c


def exp(ring, a):
    return unit_monomial(ring, _exp_index(ring, a))


def exp_signed(ring, a):
    c = center(ring, a)
    if c >= 1 << 64:
        raise RingError("exp_signed: ConversionError::ToInteger")
    return monomial(ring, c, sign(ring, a))


def psi_range_check_batch(ring, values):
    """psi_range_check for many scalars with one batched ring product: -> list of bool (True = Ok(()), False = the
    RangeCheck error)."""
    values = [int(v) % ring.modulus for v in values]
    if not values:
        return []
    flat = []
    for a in values:
        flat += _monomial_ints(ring, _exp_index(ring, a), 1)
    b = from_ints(ring, flat)
    ps = np.tile(psi(ring), len(values))
    prod = to_ints(ring, ring.mul(ps, b))
    return [prod[e * ring.degree] == a for e, a in enumerate(values)]


def psi_range_check(ring, a):
    if not psi_range_check_batch(ring, [a])[0]:
        raise RingError("Range check failed")
