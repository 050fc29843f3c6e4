"""ORACLE (test infrastructure, not product code): pure-Python big-integer model.

Independent restatement of the reference's CRT/NTT hot path in ordinary integer
arithmetic (no Montgomery form, no C).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this.  It is slow on purpose and is used to
(1) check the reference's literal KATs (tests/golden/reference_kats.json) and
(2) cross-check the C restatement (oracle/sr_oracle.c) on small sizes.

Parity status: pinned at STANDARD-FORM level by the reference's literal KATs
(goldilocks/ntt.rs:563-787, stark_prime/ntt.rs:377-545, babybear/ntt.rs:866-1019).
The raw Montgomery limb image (a * 2^(64N) mod p, ark-ff 0.4.2 convention) is an
assumption no reference test pins ("parity unpinned" at byte level).

Each function cites the reference file:line it follows (paths relative to
crates/ring/src/cyclotomic_ring/).
"""

# ---------------------------------------------------------------- primes
GOLDILOCKS_P = 2**64 - 2**32 + 1            # models/goldilocks/mod.rs:20-24, generator 7
BABYBEAR_P = 2013265921                      # models/babybear/mod.rs:21-25, generator 31
STARK_P = 2**251 + 17 * 2**192 + 1           # models/stark_prime/mod.rs:20-24, generator 3

FROG_P = 15912092521325583641                # models/frog_ring/mod.rs:19-25, generator 3 ("next" row 4)

PRIMES = {
    "frog": (FROG_P, 3, 1),
    "goldilocks": (GOLDILOCKS_P, 7, 1),      # (p, generator, u64 limbs N)
    "babybear": (BABYBEAR_P, 31, 1),         # stored as Fp64 (N=1, R=2^64): babybear/mod.rs:25
    "stark": (STARK_P, 3, 4),
}


def mont_r(name):
    p, _, n = PRIMES[name]
    return pow(2, 64 * n, p)


def to_mont(name, x):
    """standard integer -> ark-ff in-memory residue a*R mod p (R = 2^(64N))."""
    p, _, n = PRIMES[name]
    return (x << (64 * n)) % p


def from_mont(name, x):
    p, _, n = PRIMES[name]
    return x * pow(pow(2, 64 * n, p), -1, p) % p


def brv(x, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (x & 1)
        x >>= 1
    return r


# ---------------------------------------------------------------- power-of-two negacyclic ring
def psi(name, log2_d):
    """psi = g^((p-1)/(2D)) -- SURVEY Appendix A; for stark D=16 this is ROOTS_OF_UNITY_32[1]
    (stark_prime/ntt.rs:16-49)."""
    p, g, _ = PRIMES[name]
    return pow(g, (p - 1) >> (log2_d + 1), p)


def pow2_fwd(name, a, log2_d):
    """Forward negacyclic NTT, Cooley-Tukey, natural in / bit-reversed-exponent out.
    Generalises stark_prime/ntt.rs:121-235: stage s, block b uses psi^brv_k(2^s+b)."""
    p = PRIMES[name][0]
    d = 1 << log2_d
    assert len(a) == d
    a = list(a)
    ps = psi(name, log2_d)
    for s in range(log2_d):
        half = d >> (s + 1)
        for b in range(1 << s):
            w = pow(ps, brv((1 << s) + b, log2_d), p)
            base = b * 2 * half
            for i in range(half):
                u = a[base + i]
                v = w * a[base + half + i] % p
                a[base + i] = (u + v) % p
                a[base + half + i] = (u - v) % p
    return a


def pow2_inv(name, a, log2_d):
    """Inverse (Gentleman-Sande), stages k-1..0, then D^-1.  stark_prime/ntt.rs:245-346
    (the reference folds D^-1 into the last stage: ntt.rs:339-345; same values)."""
    p = PRIMES[name][0]
    d = 1 << log2_d
    assert len(a) == d
    a = list(a)
    ps_inv = pow(psi(name, log2_d), -1, p)
    for s in range(log2_d - 1, -1, -1):
        half = d >> (s + 1)
        for b in range(1 << s):
            w = pow(ps_inv, brv((1 << s) + b, log2_d), p)
            base = b * 2 * half
            for i in range(half):
                u = a[base + i]
                v = a[base + half + i]
                a[base + i] = (u + v) % p
                a[base + half + i] = w * (u - v) % p
    dinv = pow(d, -1, p)
    return [x * dinv % p for x in a]


def pow2_reduce(name, c, log2_d):
    """reduce mod X^D+1: lo[i] -= hi[i]  (stark_prime/mod.rs:40-47)."""
    p = PRIMES[name][0]
    d = 1 << log2_d
    out = list(c[:d]) + [0] * (d - min(d, len(c)))
    for i, x in enumerate(c[d:]):
        out[i] = (out[i] - x) % p
    return out


def schoolbook(name, a, b):
    """coeff_form.rs:54-67 poly_mul: 2D-1 term convolution (before reduce)."""
    p = PRIMES[name][0]
    c = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                c[i + j] = (c[i + j] + x * y) % p
    return c


def pow2_ring_mul(name, a, b, log2_d):
    p = PRIMES[name][0]
    fa = pow2_fwd(name, a, log2_d)
    fb = pow2_fwd(name, b, log2_d)
    return pow2_inv(name, [x * y % p for x, y in zip(fa, fb)], log2_d)


# ---------------------------------------------------------------- X^D - X^(D/2) + 1 rings (D = 24, 72)
def roots24(name):
    """ROOTS_OF_UNITY_24[k] = omega^k, omega = g^((p-1)/24)
    (goldilocks/ntt.rs:15-40, babybear/ntt.rs:16-41)."""
    p, g, _ = PRIMES[name]
    w = pow(g, (p - 1) // 24, p)
    return [pow(w, k, p) for k in range(24)]


def _three_stage_fwd(name, a):
    """goldilocks/ntt.rs:146-225 and babybear/ntt.rs:154-233 (identical shape, D = len(a))."""
    p = PRIMES[name][0]
    R = roots24(name)
    D = len(a)
    a = list(a)
    h = D // 2
    for i in range(h):
        ci, cj = a[i], a[h + i]
        z = R[4] * cj % p
        a[i] = (ci + z) % p
        a[h + i] = (ci + cj - z) % p
    q = D // 4
    for i in range(q):
        for base, r in ((0, 2), (h, 10)):
            ci, cj = a[base + i], a[base + q + i]
            t = R[r] * cj % p
            a[base + i] = (ci + t) % p
            a[base + q + i] = (ci - t) % p
    e = D // 8
    for i in range(e):
        for base, r in ((0, 1), (q, 7), (h, 5), (3 * q, 11)):
            ci, cj = a[base + i], a[base + e + i]
            t = R[r] * cj % p
            a[base + i] = (ci + t) % p
            a[base + e + i] = (ci - t) % p
    return a


def _three_stage_inv(name, a):
    """goldilocks/ntt.rs:250-318 and babybear/ntt.rs:249-316."""
    p = PRIMES[name][0]
    R = roots24(name)
    D = len(a)
    a = list(a)
    h, q, e = D // 2, D // 4, D // 8
    kappa = pow((2 * R[4] - 1) % p, -1, p)   # babybear/ntt.rs:136; goldilocks/ntt.rs:42-43 (doc says 2z-1, value is its inverse)
    inv8 = pow(8, -1, p)
    inv4 = pow(4, -1, p)
    for i in range(e):
        for base, r in ((0, 23), (q, 17), (h, 19), (3 * q, 13)):
            ci, cj = a[base + i], a[base + e + i]
            a[base + i] = (ci + cj) % p
            a[base + e + i] = R[r] * (ci - cj) % p
    for i in range(q):
        for base, r in ((0, 22), (h, 14)):
            ci, cj = a[base + i], a[base + q + i]
            a[base + i] = (ci + cj) % p
            a[base + q + i] = R[r] * (ci - cj) % p
    for i in range(h):
        ci, cj = a[i], a[h + i]
        kd = kappa * (ci - cj) % p
        a[i] = inv8 * (ci + cj - kd) % p
        a[h + i] = inv4 * kd % p
    return a


# ---- goldilocks homogenize maps: goldilocks/ntt.rs:326-437.  Each entry: (dst index <- (src index, root index or 'neg'))
_G_HOMO = [
    None,                                            # e=1: identity
    {1: (1, "neg")},                                 # 13 : c1 = -c1              :350-352
    {1: (1, 2), 2: (2, 4)},                          # 7                           :360-363
    {1: (1, 6), 2: (2, 12)},                         # 19                          :372-375
    {1: (2, 3), 2: (1, 1)},                          # 5                           :384-388
    {1: (2, 11), 2: (1, 5)},                         # 17                          :398-402
    {1: (2, 7), 2: (1, 3)},                          # 11                          :412-416
    {1: (2, 15), 2: (1, 7)},                         # 23                          :426-430
]
_G_DEHOMO = [
    None,
    {1: (1, "neg")},                                 # :355-357
    {1: (1, 22), 2: (2, 20)},                        # :366-369
    {1: (1, 18), 2: (2, 12)},                        # :378-381
    {1: (2, 23), 2: (1, 21)},                        # :391-395
    {1: (2, 19), 2: (1, 13)},                        # :405-409
    {1: (2, 21), 2: (1, 17)},                        # :419-423
    {1: (2, 17), 2: (1, 9)},                         # :433-437
]


def _apply_maps(name, c, maps, width, pre_perm=None, post_perm=None):
    p = PRIMES[name][0]
    R = roots24(name)
    c = list(c)
    for blk, m in enumerate(maps):
        seg = c[blk * width:(blk + 1) * width]
        if pre_perm and (blk == 0 or m is not None):
            seg = pre_perm(seg)
        if m:
            new = list(seg)
            for dst, (src, r) in m.items():
                if r == "neg":
                    new[dst] = (-seg[src]) % p
                elif r is None:
                    new[dst] = seg[src]
                else:
                    new[dst] = seg[src] * R[r] % p
            seg = new
        if post_perm and (blk == 0 or m is not None):
            seg = post_perm(seg)
        c[blk * width:(blk + 1) * width] = seg
    return c


def g24_homogenize(c):
    return _apply_maps("goldilocks", c, _G_HOMO, 3)


def g24_dehomogenize(c):
    return _apply_maps("goldilocks", c, _G_DEHOMO, 3)


def g24_crt(a):
    """goldilocks/ntt.rs:135-228."""
    return g24_homogenize(_three_stage_fwd("goldilocks", a))


def g24_icrt(a):
    """goldilocks/ntt.rs:240-319."""
    return _three_stage_inv("goldilocks", g24_dehomogenize(a))


def g24_reduce(c):
    """goldilocks/mod.rs:75-98 (input length <= 2D)."""
    p = GOLDILOCKS_P
    D = 24
    c = list(c) + [0] * max(0, D - len(c))
    get = lambda i: c[i] if i < len(c) else 0
    for i in range(D // 2):
        c[i] = (c[i] - get(D + i) - get(D + D // 2 + i)) % p
    for i in range(D // 2, D):
        c[i] = (c[i] + get(D // 2 + i)) % p
    return c[:D]


# ---- babybear homogenize maps: babybear/ntt.rs:324-588
def _bb_perm(seg):
    """SWAPS (1,3),(2,6),(5,7): babybear/ntt.rs:580-588."""
    seg = list(seg)
    for i, j in ((1, 3), (2, 6), (5, 7)):
        seg[i], seg[j] = seg[j], seg[i]
    return seg


# homogenize: maps applied first, then permutation (blocks 1..7); block 0 permutation only
_B_HOMO = [
    None,
    {1: (7, 10), 7: (4, 5), 4: (1, 1), 2: (5, 7), 5: (8, 11), 8: (2, 2), 3: (3, 4), 6: (6, 8)},      # 13 :351-365
    {1: (4, 3), 4: (7, 5), 7: (1, None), 2: (8, 6), 8: (5, 3), 5: (2, 1), 3: (3, 2), 6: (6, 4)},     # 7  :385-399
    {1: (1, 2), 2: (2, 4), 3: (3, 6), 4: (4, 8), 5: (5, 10), 6: (6, "neg"), 7: (7, 14), 8: (8, 16)},  # 19 :419-429
    {1: (2, 1), 2: (4, 2), 4: (8, 4), 8: (7, 3), 7: (5, 2), 5: (1, None), 3: (6, 3), 6: (3, 1)},     # 5  :445-458
    {1: (8, 15), 8: (1, 1), 2: (7, 13), 7: (2, 3), 3: (6, 11), 6: (3, 5), 4: (5, 9), 5: (4, 7)},      # 17 :477-494
    {1: (5, 6), 5: (7, 8), 7: (8, 9), 8: (4, 4), 4: (2, 2), 2: (1, 1), 3: (6, 7), 6: (3, 3)},         # 11 :517-530
    {1: (2, 5), 2: (4, 10), 4: (8, 20), 8: (7, 17), 7: (5, "neg"), 5: (1, 2), 3: (6, 15), 6: (3, 7)}, # 23 :549-562
]
# dehomogenize: permutation first, then maps
_B_DEHOMO = [
    None,
    {1: (4, 23), 4: (7, 19), 7: (1, 14), 2: (8, 22), 8: (5, 13), 5: (2, 17), 3: (3, 20), 6: (6, 16)},   # :368-382
    {1: (7, None), 7: (4, 19), 4: (1, 21), 2: (5, 23), 5: (8, 21), 8: (2, 18), 3: (3, 22), 6: (6, 20)}, # :402-416
    {1: (1, 22), 2: (2, 20), 3: (3, 18), 4: (4, 16), 5: (5, 14), 6: (6, "neg"), 7: (7, 10), 8: (8, 8)},  # :432-442
    {1: (5, None), 5: (7, 22), 7: (8, 21), 8: (4, 20), 4: (2, 22), 2: (1, 23), 3: (6, 23), 6: (3, 21)},  # :461-474
    {1: (8, 23), 8: (1, 9), 2: (7, 21), 7: (2, 11), 3: (6, 19), 6: (3, 13), 4: (5, 17), 5: (4, 15)},     # :497-514
    {1: (2, 23), 2: (4, 22), 4: (8, 20), 8: (7, 15), 7: (5, 16), 5: (1, 18), 3: (6, 21), 6: (3, 17)},    # :533-546
    {1: (5, 22), 5: (7, "neg"), 7: (8, 7), 8: (4, 4), 4: (2, 14), 2: (1, 19), 3: (6, 17), 6: (3, 9)},    # :565-578
]


def bb72_homogenize(c):
    return _apply_maps("babybear", c, _B_HOMO, 9, post_perm=_bb_perm)


def bb72_dehomogenize(c):
    return _apply_maps("babybear", c, _B_DEHOMO, 9, pre_perm=_bb_perm)


def bb72_crt(a):
    """babybear/ntt.rs:143-236."""
    return bb72_homogenize(_three_stage_fwd("babybear", a))


def bb72_icrt(a):
    """babybear/ntt.rs:238-317."""
    return _three_stage_inv("babybear", bb72_dehomogenize(a))


def bb72_reduce(c):
    """babybear/mod.rs:87-110."""
    p = BABYBEAR_P
    D = 72
    c = list(c) + [0] * max(0, D - len(c))
    get = lambda i: c[i] if i < len(c) else 0
    for i in range(D // 2):
        c[i] = (c[i] - get(D + i) - get(D + D // 2 + i)) % p
    for i in range(D // 2, D):
        c[i] = (c[i] + get(D // 2 + i)) % p
    return c[:D]


# ---------------------------------------------------------------- extension-field slot products (ntt_form.rs:159-189)
def fq3_mul(name, x, y):
    """Fp3 = Fq[u]/(u^3 - NONRESIDUE), (c0,c1,c2); NONRESIDUE = ROOTS[1]
    (goldilocks/mod.rs:42 = 2^40; babybear/mod.rs:40 = 503591070)."""
    p = PRIMES[name][0]
    nr = roots24(name)[1]
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] += x[i] * y[j]
    return [(c[0] + nr * c[3]) % p, (c[1] + nr * c[4]) % p, c[2] % p]


def fq9_mul(x, y):
    """BabyBear Fq9 = Fq3[v]/(v^3 - u) (babybear/mod.rs:51-66, fq9.rs:7-58): memory index 3i+j
    holds the X^(i+3j) coefficient of Fq[X]/(X^9 - NONRESIDUE) (v = X, u = X^3)."""
    p = BABYBEAR_P
    nr = roots24("babybear")[1]
    px = [0] * 9
    py = [0] * 9
    for i in range(3):
        for j in range(3):
            px[i + 3 * j] = x[3 * i + j]
            py[i + 3 * j] = y[3 * i + j]
    c = [0] * 17
    for i in range(9):
        for j in range(9):
            c[i + j] += px[i] * py[j]
    r = [(c[k] + nr * (c[k + 9] if k + 9 < 17 else 0)) % p for k in range(9)]
    out = [0] * 9
    for i in range(3):
        for j in range(3):
            out[3 * i + j] = r[i + 3 * j]
    return out


def g24_ntt_mul(x, y):
    out = []
    for s in range(8):
        out += fq3_mul("goldilocks", x[3 * s:3 * s + 3], y[3 * s:3 * s + 3])
    return out


def bb72_ntt_mul(x, y):
    out = []
    for s in range(8):
        out += fq9_mul(x[9 * s:9 * s + 9], y[9 * s:9 * s + 9])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# balanced (gadget) decomposition -- SURVEY 8f #2; plain-integer restatement of
#   decompose_balanced_in_place   crates/ring/src/balanced_decomposition/mod.rs:62-117
#   signed representative         balanced_decomposition/fq_convertible.rs:21-35, stark_prime/decomposition.rs:41-53
#   rounded_div                   crates/linear_algebra/src/ops.rs:64-80
#   recompose                     balanced_decomposition/mod.rs:119-131
def signed_representative(x, p):
    """[0, (p-1)/2] -> itself, ](p-1)/2, p[ -> x - p"""
    return x - p if x > (p - 1) // 2 else x


def _trunc_divmod(a, b):
    """Rust / and % on signed integers: quotient rounds toward zero, remainder has the sign of the dividend"""
    q = abs(a) // b
    q = -q if a < 0 else q
    return q, a - q * b


def rounded_div(dividend, divisor):
    if (dividend ^ divisor) >= 0:
        return _trunc_divmod(dividend + divisor // 2, divisor)[0]
    return _trunc_divmod(dividend - divisor // 2, divisor)[0]


def decompose_balanced(x, p, b, k):
    """k balanced base-b digits (signed integers) of the field element x (standard form); raises IndexError where the
    reference indexes past `out` (more than k digits needed)"""
    if b in (0, 1):
        raise ValueError("cannot decompose in basis 0 or 1")
    if b % 2:
        raise ValueError("decomposition basis must be even")
    curr = signed_representative(x, p)
    out = [0] * k
    i = 0
    while True:
        q, rem = _trunc_divmod(curr, b)
        if abs(rem) <= b // 2:
            out[i] = rem          # IndexError when i == k, like the reference's panic
            curr = q
        else:
            out[i] = rem + b if rem < 0 else rem - b
            curr = q + rounded_div(rem, b)
        i += 1
        if curr == 0:
            break
    return out


def recompose(digits, b, p):
    acc = 0
    for d in reversed(digits):
        acc = (acc * b + d) % p
    return acc


# ---------------------------------------------------------------------------------------------------------------------
# frog ring Fq[X]/(X^16 + 1) -> 4 x Fq4 -- SURVEY 8f #4; restatement of models/frog_ring/ntt.rs:108-267 and mod.rs:36-60
def roots8():
    """ROOTS_OF_UNITY_8[k] = w^k, w = 3^((p-1)/8) (frog_ring/ntt.rs:15-24; checked against the literals in the tests)"""
    w = pow(3, (FROG_P - 1) // 8, FROG_P)
    return [pow(w, k, FROG_P) for k in range(8)]


# (source index, root index) per output index; root -1 = copy, -2 = negate.  frog_ring/ntt.rs:215-291
_FROG_HOMO = [
    [(0, -1), (2, -1), (1, -1), (3, -1)],          # nonresidue_to_nonresidue: swap(2, 1)
    [(0, -1), (2, 2), (1, 1), (3, 3)],             # nonresidue_to_5_to_nonresidue
    [(0, -1), (2, 1), (3, 6), (1, -2)],            # nonresidue_to_3_to_nonresidue
    [(0, -1), (2, 3), (3, 5), (1, 1)],             # nonresidue_to_7_to_nonresidue
]
_FROG_DEHOMO = [
    [(0, -1), (2, -1), (1, -1), (3, -1)],
    [(0, -1), (2, 7), (1, 6), (3, 5)],             # nonresidue_to_nonresidue_to_5
    [(0, -1), (3, -2), (1, 7), (2, 2)],            # nonresidue_to_nonresidue_to_3
    [(0, -1), (3, 7), (1, 5), (2, 3)],             # nonresidue_to_nonresidue_to_7
]


def _frog_maps(c, maps):
    p, R = FROG_P, roots8()
    out = list(c)
    for blk in range(4):
        seg = c[4 * blk:4 * blk + 4]
        for i, (src, r) in enumerate(maps[blk]):
            v = seg[src]
            out[4 * blk + i] = v if r == -1 else ((-v) % p if r == -2 else v * R[r] % p)
    return out


def frog16_homogenize(c):
    return _frog_maps(c, _FROG_HOMO)


def frog16_dehomogenize(c):
    return _frog_maps(c, _FROG_DEHOMO)


def frog16_crt(a):
    """serial_frog_crt_in_place (ntt.rs:114-151): two radix-2 stages with ROOTS[2], then ROOTS[1] / ROOTS[3]; homogenize."""
    p, R = FROG_P, roots8()
    a = list(a)
    assert len(a) == 16
    for i in range(8):
        x, z = a[i], R[2] * a[8 + i] % p
        a[i], a[8 + i] = (x + z) % p, (x - z) % p
    for i in range(4):
        x, z = a[i], R[1] * a[4 + i] % p
        a[i], a[4 + i] = (x + z) % p, (x - z) % p
        x, z = a[8 + i], R[3] * a[12 + i] % p
        a[8 + i], a[12 + i] = (x + z) % p, (x - z) % p
    return frog16_homogenize(a)


def frog16_icrt(a):
    """serial_frog_icrt_in_place (ntt.rs:163-200)"""
    p, R = FROG_P, roots8()
    inv4 = pow(4, -1, p)
    a = frog16_dehomogenize(list(a))
    for i in range(4):
        x, y = a[i], a[4 + i]
        a[i], a[4 + i] = (x + y) % p, R[7] * (x - y) % p
        x, y = a[8 + i], a[12 + i]
        a[8 + i], a[12 + i] = (x + y) % p, R[5] * (x - y) % p
    for i in range(8):
        x, y = a[i], a[8 + i]
        a[i], a[8 + i] = inv4 * (x + y) % p, inv4 * (R[6] * (x - y) % p) % p
    return a


def frog16_reduce(c):
    """reduce_in_place (frog_ring/mod.rs:72-79): lo -= hi, missing coefficients are zero"""
    p = FROG_P
    get = lambda i: c[i] if i < len(c) else 0
    return [(get(i) - get(16 + i)) % p for i in range(16)]


def fq4_mul(x, y):
    """Fq4 = Fq2[v]/(v^2 - u), Fq2 = Fq[u]/(u^2 - NONRESIDUE), NONRESIDUE = ROOTS[1] (frog_ring/mod.rs:36-60);
    memory (c0.c0, c0.c1, c1.c0, c1.c1)"""
    p, nr = FROG_P, roots8()[1]

    def m2(a, b):
        return [(a[0] * b[0] + nr * a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p]

    def mul_u(a):
        return [nr * a[1] % p, a[0]]

    a0, a1, b0, b1 = x[0:2], x[2:4], y[0:2], y[2:4]
    t = mul_u(m2(a1, b1))
    c0 = [(u + v) % p for u, v in zip(m2(a0, b0), t)]
    c1 = [(u + v) % p for u, v in zip(m2(a0, b1), m2(a1, b0))]
    return c0 + c1


def frog16_ntt_mul(x, y):
    out = []
    for s_ in range(4):
        out += fq4_mul(x[4 * s_:4 * s_ + 4], y[4 * s_:4 * s_ + 4])
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Cyclotomic::rot (crates/ring/src/traits.rs:54-66): multiply by X modulo the ring's polynomial
def rot(c, p, trinomial=False):
    """X^D + 1 (stark_prime/mod.rs:87-95, frog_ring/mod.rs:126-134): new[0] = -old[D-1], new[i] = old[i-1];
    X^D - X^(D/2) + 1 (goldilocks/mod.rs:138-149, babybear/mod.rs:150-161): additionally new[D/2] += old[D-1]"""
    last = c[-1]
    out = [(-last) % p] + list(c[:-1])
    if trinomial:
        out[len(c) // 2] = (out[len(c) // 2] + last) % p
    return out


# ---------------------------------------------------------------------------------------------------------------------
# ark-serialize wire format: coeff_form.rs:154-189 / ntt_form.rs:24 serialise the flat coefficient array;
# each coefficient is ark-ff 0.4.2 Fp::serialize_with_flags with EmptyFlags (third party, restated from the published format):
# the standard-form integer as ceil(MODULUS_BIT_SIZE / 8) little-endian bytes.  Parity unpinned (no golden bytes in the reference).
def wire_bytes(name):
    return (PRIMES[name][0].bit_length() + 7) // 8


def serialize(name, std_coeffs):
    w = wire_bytes(name)
    return b"".join(int(x).to_bytes(w, "little") for x in std_coeffs)


def deserialize(name, data):
    """-> list of standard-form integers; ValueError("InvalidData") for an integer >= p (Fp::from_bigint -> None)."""
    w, p = wire_bytes(name), PRIMES[name][0]
    if len(data) % w:
        raise ValueError("truncated")
    out = [int.from_bytes(data[i:i + w], "little") for i in range(0, len(data), w)]
    if any(x >= p for x in out):
        raise ValueError("InvalidData")
    return out


def serialize_vec(name, elems):
    """Vec<R>: u64 little-endian length, then the elements (ark-serialize Vec<T>)."""
    return len(elems).to_bytes(8, "little") + b"".join(serialize(name, e) for e in elems)


def serialize_matrix(name, rows):
    """Matrix<R> = its Vec<Vec<R>> (crates/linear_algebra/src/matrix.rs:111-124)."""
    return len(rows).to_bytes(8, "little") + b"".join(serialize_vec(name, r) for r in rows)


def serialize_sparse(name, nrows, ncols, rows):
    """SparseMatrix<R>: nrows u64, ncols u64, then Vec<Vec<(R, usize)>> (sparse_matrix.rs:158-175); usize goes out as u64."""
    out = nrows.to_bytes(8, "little") + ncols.to_bytes(8, "little") + len(rows).to_bytes(8, "little")
    for row in rows:
        out += len(row).to_bytes(8, "little")
        for elem, col in row:
            out += serialize(name, elem) + int(col).to_bytes(8, "little")
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Monomial helpers (crates/ring/src/monomial.rs:17-93), standard-form coefficient lists; Zq::center / sign ring.rs:160-181
def monomial(d, i, coeff):
    m = [0] * d                      # monomial.rs:17-21 (index out of range panics there: IndexError here)
    m[i] = coeff
    return m


def psi_table(d, p):
    """psi = sum_{i in [1, d/2)} i (X^i - X^(d-i))  (monomial.rs:36-49)"""
    out = [0] * d
    for i in range(1, d // 2):
        out[i] = (out[i] + i) % p
        out[d - i] = (out[d - i] - i) % p
    return out


def center(a, p):
    return p - a if a > (p - 1) // 2 else a


def exp_monomial(d, a, p):
    """exp(a) = sign(a) X^a as the reference computes it: the UNIT monomial X^centered for a >= 0 and X^(d - centered) otherwise
    (monomial.rs:56-66); centered must fit usize and index the array"""
    c = center(a, p)
    idx = c if a <= (p - 1) // 2 else d - c
    if not 0 <= idx < d:
        raise IndexError("monomial index out of range")
    return monomial(d, idx, 1)


def exp_signed(d, a, p):
    """monomial(centered, 1) * sign(a)  (monomial.rs:72-78)"""
    c = center(a, p)
    if not 0 <= c < d:
        raise IndexError("monomial index out of range")
    return monomial(d, c, 1 if a <= (p - 1) // 2 else p - 1)


def psi_range_check(name, log2_d, a):
    """ct(exp(a) * psi) == a  (monomial.rs:84-93); the product is the ring product modulo X^d + 1"""
    p = PRIMES[name][0]
    d = 1 << log2_d
    b = exp_monomial(d, a, p)
    prod = pow2_reduce(name, schoolbook(name, psi_table(d, p), b), log2_d)
    return prod[0] == a % p
