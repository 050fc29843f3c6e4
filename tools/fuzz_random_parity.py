#!/usr/bin/env python3
"""Randomised differential soak (not part of pytest): random ring, degree, batch and operation against the oracle for a fixed
wall-clock budget.  usage: fuzz_random_parity.py [seconds] [seed]   (SR_FUZZ_RINGS=stark restricts the rings drawn)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

import oracle_lib as O
import pyref as P
from stark_rings_amd import CyclotomicRing

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rings = {}


def ring(name, k):
    if (name, k) not in rings:
        rings[(name, k)] = CyclotomicRing(name, k)
    return rings[(name, k)]


t_end = time.time() + budget
n_checks, by_op = 0, {}
while time.time() < t_end:
    names = os.environ.get("SR_FUZZ_RINGS", "goldilocks,babybear,stark").split(",")
    name = names[int(rng.integers(0, len(names)))]
    kmax = {"goldilocks": 18, "babybear": 17, "stark": 14}[name]
    k = int(rng.integers(0, kmax + 1))
    F = O.FIELD_ID[name]
    p = P.PRIMES[name][0]
    d = 1 << k
    cap = max(1, min(40, (1 << 19) // d))
    batch = int(rng.integers(1, cap + 1)) if rng.random() < 0.8 else int(rng.integers(0, 3))
    r = ring(name, k)
    seed = int(rng.integers(0, 1 << 30))
    a = O.fill_uniform(F, seed, 0, batch * d)
    b = O.fill_uniform(F, seed + 1, 0, batch * d)
    if batch and rng.random() < 0.3:   # adversarial coefficients
        special = [0, 1, p - 1, p - 2, (p - 1) // 2, (1 << 32) % p, ((1 << 32) - 1) % p, ((1 << 64) - 1) % p]
        vals = [special[int(rng.integers(0, len(special)))] for _ in range(d)]
        a[:d * O.LIMBS[F]] = O.to_mont(F, vals)
    op = ["crt", "icrt", "mul", "ntt_mul", "add", "sub", "reduce", "decompose", "rot", "matvec", "wire", "neg", "scale", "add_scalar",
          "mul_ntt_rhs", "mul_elem", "sum", "product"][int(rng.integers(0, 18))]
    ok = True
    if op == "crt":
        ok = np.array_equal(r.elementwise_crt(a.copy()), O.pow2_fwd(F, a, k, batch, 4)) if batch else True
    elif op == "icrt":
        ok = np.array_equal(r.elementwise_icrt(O.pow2_fwd(F, a, k, batch, 4)), a) if batch else True
    elif op == "mul":
        ok = np.array_equal(r.mul(a, b), O.pow2_ring_mul(F, a, b, k, batch, 4)) if batch else r.mul(a, b).size == 0
    elif op == "ntt_mul":
        ok = np.array_equal(r.ntt_mul(a.copy(), b), O.pow2_pointwise(F, a, b)) if batch else True
    elif op in ("add", "sub"):
        if batch:
            sa, sb = O.from_mont(F, a), O.from_mont(F, b)
            got = O.from_mont(F, (r.add if op == "add" else r.sub)(a.copy(), b))
            ok = got == [((x + y) if op == "add" else (x - y)) % p for x, y in zip(sa, sb)]
    elif op == "reduce":
        in_len = int(rng.integers(0, 2 * d + 1))
        src = O.fill_uniform(F, seed + 2, 0, max(batch, 1) * in_len) if in_len else np.zeros(0, dtype=np.uint64)
        got = r.reduce(src, in_len, max(batch, 1))
        L = O.LIMBS[F]
        for e in range(max(batch, 1)):
            want = O.pow2_reduce(F, src[e * in_len * L:(e + 1) * in_len * L], in_len, k) if in_len else np.zeros(d * L, dtype=np.uint64)
            ok = ok and np.array_equal(got[e * d * L:(e + 1) * d * L], want)
    elif op == "decompose" and batch:
        basis = [2, 4, 6, 16, 1 << 10, 1 << 16, 1 << 32, 10, 1000][int(rng.integers(0, 9))]
        pad = 1
        while (basis // 2) * (basis ** pad - 1) // (basis - 1) < (p - 1) // 2:
            pad += 1
        pad += int(rng.integers(1, 3))
        want, over = O.decompose_balanced(F, a, d, batch, basis, pad)
        got = r.gadget_decompose(a, basis, pad)
        ok = (not over) and np.array_equal(got, want) and np.array_equal(r.gadget_recompose(got, basis, pad), a)
    elif op == "rot" and batch and d > 1:
        ok = np.array_equal(r.rot(a.copy()), O.rot(F, a, d, False))
    elif op == "matvec" and batch:
        ncols = int(rng.integers(1, 12))
        nrows = max(1, batch // ncols)
        m = O.fill_uniform(F, seed + 3, 0, nrows * ncols * d)
        v = O.fill_uniform(F, seed + 4, 0, ncols * d)
        L = O.LIMBS[F]
        got = r.matvec_ntt(m, v, nrows, ncols)
        ms, vs = O.from_mont(F, m), O.from_mont(F, v)
        want = [sum(ms[(rr * ncols + c) * d + s_] * vs[c * d + s_] for c in range(ncols)) % p for rr in range(nrows) for s_ in range(d)] \
            if nrows * ncols * d <= 4096 else None
        if want is not None:
            ok = O.from_mont(F, got) == want
        else:   # larger cases: row 0 against slot products from the oracle
            acc = [0] * d
            for c in range(ncols):
                prod = O.from_mont(F, O.pow2_pointwise(F, m[c * d * L:(c + 1) * d * L], v[c * d * L:(c + 1) * d * L]))
                acc = [(x + y) % p for x, y in zip(acc, prod)]
            ok = O.from_mont(F, got[:d * L]) == acc
    elif op in ("neg", "scale", "add_scalar") and batch:
        sa = O.from_mont(F, a)
        sc = int(rng.integers(0, 1 << 62)) % p if rng.random() < 0.7 else [0, 1, p - 1][int(rng.integers(0, 3))]
        img = O.to_mont(F, [sc])
        if op == "neg":
            ok = O.from_mont(F, r.neg(a.copy())) == [(p - x) % p for x in sa]
        elif op == "scale":
            ok = O.from_mont(F, r.scale(a.copy(), img)) == [x * sc % p for x in sa]
        else:
            ntt = bool(rng.integers(0, 2))
            want = [(x + sc) % p if (ntt or i % d == 0) else x for i, x in enumerate(sa)]   # power-of-two rings: one word per slot
            ok = O.from_mont(F, r.add_scalar(a.copy(), img, ntt)) == want
    elif op == "mul_ntt_rhs" and batch:
        ok = np.array_equal(r.mul_ntt_rhs(a, O.pow2_fwd(F, b, k, batch, 4)), O.pow2_ring_mul(F, a, b, k, batch, 4))
    elif op == "mul_elem" and batch:   # Matrix<R> *= &R: every element times the first element of b
        L = O.LIMBS[F]
        one = b[:d * L].copy()
        got = r.mul_elem(a.copy(), one)
        ok = all(np.array_equal(got[e * d * L:(e + 1) * d * L], O.pow2_pointwise(F, a[e * d * L:(e + 1) * d * L], one)) for e in range(batch))
    elif op == "sum":       # impl Sum: fold(zero(), +) over the slice, word-wise (batch 0: zero())
        sa = O.from_mont(F, a) if batch else []
        ok = O.from_mont(F, r.sum(a)) == [sum(sa[e * d + i] for e in range(batch)) % p for i in range(d)]
    elif op == "product":   # impl Product for RqNTT: fold(one(), *) slot-wise, left to right like the reference (batch 0: one())
        L = O.LIMBS[F]
        acc = O.to_mont(F, [1] * d)
        for e in range(batch):
            acc = O.pow2_pointwise(F, acc, a[e * d * L:(e + 1) * d * L])
        ok = np.array_equal(r.product(a), acc)
    elif op == "wire" and batch:
        wire = r.serialize(a)
        ok = np.array_equal(wire, O.serialize(F, a)) and np.array_equal(r.deserialize(wire), a)
    n_checks += 1
    by_op[op] = by_op.get(op, 0) + 1
    if not ok:
        print("MISMATCH: %s k=%d batch=%d op=%s seed=%d" % (name, k, batch, op, seed))
        sys.exit(1)
if os.environ.get("SR_LIB_PATH", "").endswith("_check.so"):   # the invariant-checking build: its counters must have stayed at zero
    import ctypes

    from stark_rings_amd import _lib
    buf = (ctypes.c_uint64 * 8)()
    assert _lib.load().sr_selftest_rep_counters(buf, 0) == 0
    c = [int(v) for v in buf]
    print("checking build: invariant counters %s, largest Stark limb %d" % (c[:7], c[7]))
    if any(c[:7]):
        sys.exit(1)
print("random differential soak: %d checks in %.0f s, 0 mismatches; per op %s" % (n_checks, budget, by_op))
