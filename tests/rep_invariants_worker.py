"""Worker of tests/test_rep_invariants.py: runs in its own process with SR_LIB_PATH pointing at the CHECKING build of the library
(libstarkrings_hip_check.so = the product sources compiled with -DSR_GL_CHECK_REPS, csrc/fields.hpp: repcheck) and drives the real
kernels of every tuned Goldilocks plan over uniform, edge, structured and crafted operands.  After every group of calls the device
counters must read zero: no canonical butterfly saw a representative >= p, no lazy sum wrapped twice, no lazy difference borrowed
twice, no word >= p left the library, no "canonical out" routine returned one.  Results are compared with the oracle as well (the
checking build computes the same values).  Prints one line per plan and `rep_invariants: OK` at the end."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import ctypes  # noqa: E402

import oracle_lib as O  # noqa: E402
import model_fast_goldilocks as M  # noqa: E402
from stark_rings_amd import CyclotomicRing, _lib  # noqa: E402
from test_gpu_parity import _structured_operands, edge_and_random  # noqa: E402

F = O.GOLDILOCKS
P = 0xFFFFFFFF00000001
NAMES = ["canonical routine handed a representative >= p", "lazy sum wrapped twice", "lazy difference borrowed twice",
         "word >= p left the library", "'canonical out' routine returned >= p"]


def counters(reset=True):
    buf = (ctypes.c_uint64 * 8)()
    rc = _lib.load().sr_selftest_rep_counters(buf, 1 if reset else 0)
    assert rc == 0, "not the checking build: %s" % _lib.last_error()
    return [int(v) for v in buf]


def expect_clean(what):
    c = counters()
    bad = ["%s: %d" % (NAMES[i], c[i]) for i in range(5) if c[i]]
    assert not bad, "%s: %s" % (what, "; ".join(bad))


def canonical(a):
    return bool((a < np.uint64(P)).all())


def crafted(k, batch, seed):
    """NTT-domain operands whose inverse networks leave the representative p in the slot that skips the table product
    (tests/test_gpu_parity.py::test_goldilocks_inverse_with_lazy_sums_landing_on_p)"""
    d = 1 << k
    rng = random.Random(seed)
    raw = O.fill_uniform(F, 0x51 + k, 0, batch * d).copy()
    for blk in range(batch * d // 256):
        raw[blk * 256:blk * 256 + 16] = np.array(M.crafted_inverse_block(rng, 1 + blk % 15), dtype=np.uint64)
    if k in (13, 20, 21):
        T = M.tables(13)
        for tile in range(batch * d // 4096):
            slots = [1 + (i0 * 7 + tile) % 15 for i0 in range(16)]
            raw[tile * 4096:tile * 4096 + 256] = np.array(M.crafted_inverse_block256(rng, T, slots), dtype=np.uint64)
    return raw


def run_plan(k, env=None):
    for key, val in (env or {}).items():
        os.environ[key] = val
    try:
        ring = CyclotomicRing("goldilocks", k, device=0)
    finally:
        for key in (env or {}):
            del os.environ[key]
    tag = "D=2^%d%s" % (k, (" " + ",".join("%s=%s" % kv for kv in env.items())) if env else "")
    counters()  # table construction uses the same arithmetic: start from zero
    d = 1 << k
    batch = 6 if k <= 17 else 3
    # uniform + edge elements
    a = edge_and_random(F, k, batch, 0xA0 + k)
    b = np.roll(edge_and_random(F, k, batch, 0xB0 + k).reshape(batch, d), 2, axis=0).reshape(-1).copy()  # edge elements meet uniform ones
    fa = ring.elementwise_crt(a.copy())
    assert np.array_equal(fa, O.pow2_fwd(F, a, k, batch, 4)) and canonical(fa)
    assert np.array_equal(ring.elementwise_icrt(fa.copy()), a)
    prod = ring.mul(a, b)
    assert np.array_equal(prod, O.pow2_ring_mul(F, a, b, k, batch, 4)) and canonical(prod)
    fb = ring.elementwise_crt(b.copy())
    assert np.array_equal(ring.mul_ntt_rhs(a, fb), prod)
    expect_clean(tag + " uniform and edge operands")
    # structured operands: equal legs, zero differences, sums that land on 0 or p all the time (D = 2^21 runs the tile kernels of
    # 2^13 behind canonical strided passes: skipped there, the operands are built by Python loops)
    els = _structured_operands(F, "goldilocks", k, 0x5EED + k) if k <= 20 else []
    n = len(els)
    s = O.to_mont(F, [v for e in els for v in e])
    if n:
        fs = ring.elementwise_crt(s.copy())
        assert np.array_equal(fs, O.pow2_fwd(F, s, k, n, 4)) and canonical(fs)
        assert np.array_equal(ring.elementwise_icrt(fs.copy()), s)
        s2 = s.reshape(n, -1)
    for shift in ((0, 3) if n else ()):
        t = np.ascontiguousarray(np.roll(s2, shift, axis=0)).reshape(-1)
        got = ring.mul(s, t)
        assert np.array_equal(got, O.pow2_ring_mul(F, s, t, k, n, 4)) and canonical(got), "structured, shift %d" % shift
        assert np.array_equal(ring.mul_ntt_rhs(s, ring.elementwise_crt(t.copy())), got)
    expect_clean(tag + " structured operands")
    # crafted NTT-domain operands: lazy sums that land exactly on p
    if k >= 13:
        cb = 2 if k <= 17 else 1
        raw = crafted(k, cb, 0x1A2 + k)
        want = O.pow2_inv(F, raw, k, cb, 4)
        got = ring.elementwise_icrt(raw.copy())
        assert np.array_equal(got, want) and canonical(got)
        assert np.array_equal(ring.elementwise_crt(got.copy()), raw)
        one = O.to_mont(F, [1] * (cb * d))
        bb = O.pow2_inv(F, one, k, cb, 4)
        assert np.array_equal(ring.mul(want, bb), want)
        assert np.array_equal(ring.mul_ntt_rhs(want, one), want)
        expect_clean(tag + " crafted operands")
    ring.close()
    print("%-28s clean" % tag, flush=True)


def run_lanes(k, batch):
    """the two-lane plans (chunks on two streams; since round 4 their column pass is cols256_keep_kernel): a batch of several
    chunks plus a ragged one, uniform elements with eight crafted ones in front"""
    ring = CyclotomicRing("goldilocks", k, device=0)
    counters()
    d = 1 << k
    a = O.fill_uniform(F, 0xC0 + k, 0, batch * d).copy()
    b = O.fill_uniform(F, 0xD0 + k, 0, batch * d).copy()
    head = 8 if k == 16 else 1
    a[:head * d] = O.pow2_inv(F, crafted(k, head, 0x2B + k), k, head, 4)   # a = icrt(crafted): its forward transform is the crafted block
    prod = ring.mul(a, b)
    fa = ring.elementwise_crt(a.copy())
    assert canonical(prod) and canonical(fa)
    assert np.array_equal(ring.elementwise_icrt(fa.copy()), a)
    assert np.array_equal(ring.mul_ntt_rhs(a, ring.elementwise_crt(b.copy())), prod)
    sample = sorted({0, 1, head, batch // 2, batch - 1})
    ea = np.concatenate([a[e * d:(e + 1) * d] for e in sample])
    eb = np.concatenate([b[e * d:(e + 1) * d] for e in sample])
    want = O.pow2_ring_mul(F, ea, eb, k, len(sample), 4)
    wf = O.pow2_fwd(F, ea, k, len(sample), 4)
    for i, e in enumerate(sample):
        assert np.array_equal(prod[e * d:(e + 1) * d], want[i * d:(i + 1) * d]), e
        assert np.array_equal(fa[e * d:(e + 1) * d], wf[i * d:(i + 1) * d]), e
    expect_clean("D=2^%d, %d elements on two lanes" % (k, batch))
    ring.close()
    print("%-28s clean" % ("D=2^%d x %d on two lanes" % (k, batch)), flush=True)


def negative_control():
    """the hooks are live: raw operand words >= p (not a field element's image) must be counted wherever a canonical routine takes
    them -- the inverse transform's first network starts with twiddle-1 butterflies, the ring add is Goldilocks::add.  (The FORWARD
    transform is not a negative control: its first pass is all shifted, lazy butterflies, which take any 64-bit word by design.)"""
    ring = CyclotomicRing("goldilocks", 16, device=0)
    counters()
    a = O.fill_uniform(F, 7, 0, 1 << 16).copy()
    a[::2] = np.uint64(P)          # p itself, p + 5: representatives >= p
    a[1::4] = np.uint64(P + 5)
    ring.elementwise_icrt(a.copy())
    c1 = counters()
    assert c1[0] > 0, "the checking build did not count non-canonical inputs of the inverse transform: %r" % (c1,)
    ring.add(a.copy(), a.copy())
    c2 = counters()
    assert c2[0] > 0, "the checking build did not count non-canonical inputs of the ring add: %r" % (c2,)
    ring.close()
    print("negative control: %d + %d canonical-input violations counted for raw words >= p" % (c1[0], c2[0]), flush=True)


def main():
    assert os.environ.get("SR_LIB_PATH", "").endswith("_check.so"), "run with SR_LIB_PATH=<the checking build>"
    negative_control()
    for k in (10, 12, 13, 16, 17, 20, 21):
        run_plan(k)
    for k in (16, 20):
        run_plan(k, {"SR_GL_COLS256": "0"})     # the older plan: strided passes + lazy 4096-point tiles
    run_lanes(16, 264)     # two chunks of 128 and one of 8: cols256_keep_kernel on both lanes
    run_lanes(20, 17)      # chunks of 8, 8 and 1 at D = 2^20
    print("rep_invariants: OK", flush=True)


if __name__ == "__main__":
    main()
