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
         "word >= p left the library", "'canonical out' routine returned >= p",
         "Stark limb sum left the int32 range of the invariants", "Stark product operands could overflow the column accumulator"]


def counters(reset=True):
    buf = (ctypes.c_uint64 * 8)()
    rc = _lib.load().sr_selftest_rep_counters(buf, 1 if reset else 0)
    assert rc == 0, "not the checking build: %s" % _lib.last_error()
    return [int(v) for v in buf]


def expect_clean(what):
    c = counters()
    bad = ["%s: %d" % (NAMES[i], c[i]) for i in range(len(NAMES)) if c[i]]
    assert not bad, "%s: %s" % (what, "; ".join(bad))
    return c[7]     # the Stark limb high-water mark since the last reset


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


def max_limb_images(d, flavour):
    """Stark memory images (4 u64 words per coefficient, any canonical value is some element's Montgomery image) whose nine 28-bit
    limbs are as large as canonical values get: 2^251 - 1 (all limbs 0xfffffff, the top one 2^27 - 1), its negative, and values
    with every limb's top bits set."""
    PS = (1 << 251) + 17 * (1 << 192) + 1
    hi = (1 << 251) - 1
    vals = {"max": [hi] * d,
            "alternating": [hi if i & 1 else PS - hi for i in range(d)],
            "topbits": [(hi - (i * 0x0123456789ABCDEF % (1 << 24)) * sum(1 << (28 * j) for j in range(9))) % PS for i in range(d)]}[flavour]
    out = np.empty(4 * d, dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[4 * i + j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def run_stark(k, env=None):
    """The Stark rings compute in nine signed 28-bit limbs with postponed carries (csrc/stark_lazy.hpp): the invariants are limb bounds
    (no add / sub may leave int32, no product may overflow its 64-bit column accumulator) which the kernels keep by reducing weakly
    every few stages.  Uniform data stays far from the bounds; the operands here are memory images with every limb at its maximum."""
    S = O.STARK
    for key, val in (env or {}).items():
        os.environ[key] = val
    try:
        ring = CyclotomicRing("stark", k, device=0)
    finally:
        for key in (env or {}):
            del os.environ[key]
    tag = "Stark D=2^%d%s" % (k, (" " + ",".join("%s=%s" % kv for kv in env.items())) if env else "")
    counters()
    d = 1 << k
    batch = 6
    a = edge_and_random(S, k, batch, 0x3A0 + k)
    b = edge_and_random(S, k, batch, 0x3B0 + k)
    for e, flavour in enumerate(("max", "alternating", "topbits")):
        a[(e + 2) * 4 * d:(e + 3) * 4 * d] = max_limb_images(d, flavour)                    # elements 2..4 of a
        b[((e + 1) % 3 + 2) * 4 * d:((e + 1) % 3 + 3) * 4 * d] = max_limb_images(d, flavour)  # meet another flavour in b
    fa = ring.elementwise_crt(a.copy())
    prod = ring.mul(a, b)
    assert np.array_equal(ring.elementwise_icrt(fa.copy()), a)
    assert np.array_equal(ring.mul_ntt_rhs(a, ring.elementwise_crt(b.copy())), prod)
    if k <= 12:
        assert np.array_equal(fa, O.pow2_fwd(S, a, k, batch, 4))
        assert np.array_equal(prod, O.pow2_ring_mul(S, a, b, k, batch, 4))
    else:
        for e in (2, 3, 4):
            sl = slice(e * 4 * d, (e + 1) * 4 * d)
            assert np.array_equal(fa[sl], O.pow2_fwd(S, a[sl], k, 1))
            assert np.array_equal(prod[sl], O.pow2_ring_mul(S, a[sl], b[sl], k, 1, 4))
    # the inverse transform fed max-limb NTT-domain words, then the forward one on what comes out
    raw = np.concatenate([max_limb_images(d, f) for f in ("max", "alternating", "topbits")])
    back = ring.elementwise_icrt(raw.copy())
    assert np.array_equal(ring.elementwise_crt(back.copy()), raw)
    if k <= 12:
        assert np.array_equal(back, O.pow2_inv(S, raw, k, 3, 4))
    high = expect_clean(tag)
    ring.close()
    print("%-36s clean; largest |limb| in any add / sub %d = 2^31 - %.2f x 2^28" % (tag, high, ((1 << 31) - high) / (1 << 28)), flush=True)
    return high


def run_stark_sums():
    """the linear-algebra kernels sum Montgomery products lazily (a weak reduction every four terms): long rows of max-limb images"""
    import torch

    S = O.STARK
    PS = (1 << 251) + 17 * (1 << 192) + 1
    k, nrows, ncols = 4, 3, 41
    d = 1 << k
    ring = CyclotomicRing("stark", k, device=0)
    counters()
    flav = ("max", "alternating", "topbits")
    m = np.concatenate([max_limb_images(d, flav[(r + c) % 3]) for r in range(nrows) for c in range(ncols)])
    v = np.concatenate([max_limb_images(d, flav[c % 2]) for c in range(ncols)])
    rinv = pow(1 << 256, -1, PS)

    def ints(img):
        return [int(img[4 * i]) | int(img[4 * i + 1]) << 64 | int(img[4 * i + 2]) << 128 | int(img[4 * i + 3]) << 192 for i in range(len(img) // 4)]

    mi, vi = ints(m), ints(v)
    want = []
    for r in range(nrows):
        for i in range(d):
            want.append(sum(mi[(r * ncols + c) * d + i] * vi[c * d + i] for c in range(ncols)) * rinv % PS)   # images: (a R)(b R) / R
    tm = torch.from_numpy(m.view(np.int64)).cuda()
    tv = torch.from_numpy(v.view(np.int64)).cuda()
    ty = torch.empty(nrows * 4 * d, dtype=torch.int64, device="cuda")
    ring.matvec_ntt_dev(ty, tm, tv, nrows, ncols)
    assert ints(ty.cpu().numpy().view(np.uint64)) == want
    bm = np.concatenate([max_limb_images(d, flav[(c + q) % 3]) for c in range(ncols) for q in range(2)])    # ncols x 2, row-major
    bi = ints(bm)
    want2 = []
    for r in range(nrows):
        for q in range(2):
            for i in range(d):
                want2.append(sum(mi[(r * ncols + c) * d + i] * bi[(c * 2 + q) * d + i] for c in range(ncols)) * rinv % PS)
    tb = torch.from_numpy(bm.view(np.int64)).cuda()
    tc = torch.empty(nrows * 2 * 4 * d, dtype=torch.int64, device="cuda")
    ring.matmul_ntt_dev(tc, tm, tb, nrows, ncols, 2)
    assert ints(tc.cpu().numpy().view(np.uint64)) == want2
    high = expect_clean("Stark matrix products over rows of %d max-limb images" % ncols)
    ring.close()
    print("%-36s clean; largest |limb| in any add / sub %d" % ("Stark sums of %d products" % ncols, high), flush=True)


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
    which = sys.argv[1:] or ["goldilocks", "stark"]
    if "goldilocks" in which:
        negative_control()
        for k in (10, 12, 13, 16, 17, 20, 21):
            run_plan(k)
        for k in (16, 20):
            run_plan(k, {"SR_GL_COLS256": "0"})     # the older plan: strided passes + lazy 4096-point tiles
        run_lanes(16, 520)     # four chunks of 128 and one of 8: cols256_keep_kernel on both lanes, the plain pass on the ragged one
        run_lanes(20, 33)      # four chunks of 8 and one of 1 at D = 2^20
    if "stark" in which:
        highs = [run_stark(k) for k in (4, 8, 10, 11, 12, 13, 14, 15, 16)]   # one tile per element, then 1, 2 or 3 strided stages in front
        highs += [run_stark(k, {"SR_STARK_TUNED": "0"}) for k in (6, 10, 12)]   # the generic LDS kernels on the same lazy arithmetic
        highs += [run_stark(k, {"SR_ST_WHOLE_MAX": v}) for k, v in ((10, "9"), (12, "12"))]
        run_stark_sums()
        print("Stark limb high-water mark over all plans: %d of %d (2^31 - 16)" % (max(highs), (1 << 31) - 16), flush=True)
    print("rep_invariants: OK", flush=True)


if __name__ == "__main__":
    main()
