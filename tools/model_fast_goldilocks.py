"""Index-level Python model of the tuned Goldilocks path (ntt_goldilocks.hpp), checked against the
oracle's pure-Python model.  Development aid: validates the decomposition before it is written in HIP.

  forward = merged negacyclic radix-2 stages 0..c-1 (strided, twiddles tw[2^s+b])        [cols pass]
            -> twist by gamma_b^i (block b of 4096)                                      [cols pass]
            -> cyclic DFT_4096 = 16 x 16 x 16, DIF, shift-only radix-16 butterflies
               (omega_16 = 2^12), one table multiply between passes                      [rows kernel]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle"))
import random

import pyref as P

p = P.GOLDILOCKS_P


def brv(x, bits):
    return P.brv(x, bits)


# omega_64 = 7^((p-1)/64) = 8^13, so the 16th root the transform actually uses is 2^(12*13) = 2^156 = -2^60
W16_EXP = 156


def dft16_fwd(x):
    """DIF, natural in, bit-reversed out; omega_16 = 2^156."""
    x = list(x)
    half = 8
    step = W16_EXP
    while half >= 1:
        for base in range(0, 16, 2 * half):
            for j in range(half):
                a, b = x[base + j], x[base + j + half]
                x[base + j] = (a + b) % p
                x[base + j + half] = (a - b) * pow(2, (step * j) % 192, p) % p
        half //= 2
        step = (step * 2) % 192
    return x


def dft16_inv(x):
    """inverse network (unnormalised): bit-reversed in, natural out."""
    x = list(x)
    half = 1
    steps = {1: (W16_EXP * 8) % 192, 2: (W16_EXP * 4) % 192, 4: (W16_EXP * 2) % 192, 8: W16_EXP}
    while half <= 8:
        step = steps[half]
        for base in range(0, 16, 2 * half):
            for j in range(half):
                u, v = x[base + j], x[base + j + half] * pow(2, (192 - (step * j) % 192) % 192, p) % p
                x[base + j] = (u + v) % p
                x[base + j + half] = (u - v) % p
        half *= 2
    return x


# ------------------------------------------------------------------------------------------------
# Round 3: decimation-in-time butterflies with a LAZY sum / difference (ntt_goldilocks.hpp dit_phased<..., LAZY>).
# A twiddled DIT butterfly adds and subtracts t = v 2^E, which comes out of the shift product canonical (t < p); its other
# input a may then be ANY 64-bit representative:  a + t wraps at most once (+ eps on carry lands below 2^64 because
# a + t - 2^64 <= p - 2), a - t borrows at most once (- eps on borrow stays >= 0 because t - a <= p).  Six VALU instead
# of seven, outputs are arbitrary u64 representatives.  Butterflies with twiddle 1 keep the canonical seven-instruction
# form and, in both networks below, only ever see canonical inputs (their inputs come out of twiddle-1 butterflies all the
# way back to the network's inputs).  This section models the REPRESENTATIVES bit for bit and asserts those claims.
# ------------------------------------------------------------------------------------------------
M64 = 1 << 64
EPS = (1 << 32) - 1


def lazy_bf(a, v, e):
    """(a, v) -> (a + v 2^e, a - v 2^e) on u64 representatives; e in [0, 192), e % 96 != 0"""
    assert 0 <= a < M64 and 0 <= v < M64 and e % 96 != 0
    t = v * pow(2, e % 96, p) % p                     # mul_pow2: any u64 in, canonical out
    s = a + t
    if s >= M64:
        s = s - M64 + EPS
        assert s < M64, "second wrap"
    d = a - t
    if d < 0:
        d = d + M64 - EPS
        assert d >= 0, "second borrow"
    return (d, s) if e >= 96 else (s, d)


def canon_bf(a, b):
    assert a < p and b < p, "twiddle-1 butterfly on a non-canonical input"
    return (a + b) % p, (a - b) % p


def brv_bits(x, bits):
    return brv(x, bits) if bits else 0


def dft16_fwd_dit(x, lazy=False):
    """DIT, natural in, bit-reversed out: the same function as dft16_fwd.  Stage u pairs (lo, lo + (8 >> u)); every butterfly of
    block blk = lo // (16 >> u) carries omega_16^((8 >> u) brv_u(blk))."""
    x = list(x)
    for u in range(4):
        half = 8 >> u
        for i in range(8):
            blk, pos = i // half, i % half
            lo = blk * 2 * half + pos
            hi = lo + half
            e = (W16_EXP * half * brv_bits(blk, u)) % 192
            if lazy and e % 96:
                x[lo], x[hi] = lazy_bf(x[lo], x[hi], e)
            elif lazy:
                assert e == 0
                x[lo], x[hi] = canon_bf(x[lo], x[hi])
            else:
                t = x[hi] * pow(2, e, p) % p
                x[lo], x[hi] = (x[lo] + t) % p, (x[lo] - t) % p
    return x


def dft16_inv_lazy(x):
    """dft16_inv on representatives: bit-reversed in, natural out"""
    x = list(x)
    half = 1
    steps = {1: (W16_EXP * 8) % 192, 2: (W16_EXP * 4) % 192, 4: (W16_EXP * 2) % 192, 8: W16_EXP}
    while half <= 8:
        for base in range(0, 16, 2 * half):
            for j in range(half):
                e = (192 - (steps[half] * j) % 192) % 192
                lo, hi = base + j, base + j + half
                if e % 96:
                    x[lo], x[hi] = lazy_bf(x[lo], x[hi], e)
                else:
                    assert e == 0
                    x[lo], x[hi] = canon_bf(x[lo], x[hi])
        half *= 2
    return x


def cols_passA_fwd_lazy(x, tw_exp):
    """the four merged negacyclic stages of cols256 pass A (every twiddle a shift, none trivial) on representatives"""
    x = list(x)
    for u in range(4):
        half = 8 >> u
        for jj in range(16):
            if jj & half:
                continue
            e = tw_exp[(1 << u) + (jj >> (4 - u))]
            assert e % 96
            x[jj], x[jj + half] = lazy_bf(x[jj], x[jj + half], e)
    return x


def check_lazy_networks(seed=9, rounds=200):
    rng = random.Random(seed)
    edge = [0, 1, p - 1, p - 2, EPS, EPS + 1, (1 << 63), p - EPS, (1 << 32), p >> 1]
    T = cols_tables(16)
    for r in range(rounds):
        if r < 40:
            x = [rng.choice(edge) for _ in range(16)]
        else:
            x = [rng.randrange(p) for _ in range(16)]
        want = dft16_fwd(x)
        assert dft16_fwd_dit(x) == want
        got = dft16_fwd_dit(x, lazy=True)
        assert [v % p for v in got] == want and got[0] < p      # slot 0 skips the table product behind the network: canonical
        want_i = dft16_inv(x)
        got_i = dft16_inv_lazy(x)
        assert [v % p for v in got_i] == want_i and got_i[0] < p
        # pass A: any u64 on the `a` leg is fine, so non-canonical operands would be too; compare with the canonical stages
        ref = list(x)
        for u in range(4):
            half = 8 >> u
            for jj in range(16):
                if jj & half:
                    continue
                e = T["tw_exp"][(1 << u) + (jj >> (4 - u))]
                t = ref[jj + half] * pow(2, e, p) % p
                ref[jj], ref[jj + half] = (ref[jj] + t) % p, (ref[jj] - t) % p
        assert [v % p for v in cols_passA_fwd_lazy(x, T["tw_exp"])] == ref
    return True


def tables(k):
    c = k - 12
    psi = P.psi("goldilocks", k)
    D = 1 << k
    w4096 = pow(psi, 2 * D // 4096, p)
    w256 = pow(w4096, 16, p)
    T = {}
    T["twist_f"] = [[pow(psi, (2 * brv(b, c) + 1) * i, p) for i in range(4096)] for b in range(1 << c)]
    T["W1f"] = [[pow(w4096, i0 * brv(r, 4), p) for i0 in range(256)] for r in range(16)]
    T["W2f"] = [[pow(w256, i0 * brv(s, 4), p) for i0 in range(16)] for s in range(16)]
    inv = lambda v: pow(v, -1, p)
    dinv = inv(D)
    T["twist_i"] = [[inv(T["twist_f"][b][i]) * inv(4096) % p for i in range(4096)] for b in range(1 << c)]
    T["W1i"] = [[inv(v) for v in row] for row in T["W1f"]]
    T["W2i"] = [[inv(v) for v in row] for row in T["W2f"]]
    T["tw"] = [pow(psi, brv(i, k), p) for i in range(D)]
    T["itw"] = [inv(v) for v in T["tw"]]
    T["cols_scale"] = inv(1 << c)
    return T


def strided_fwd(a, k, s_lo, M, T, twist):
    D = 1 << k
    a = list(a)
    Bsz = D >> s_lo
    S = Bsz >> M
    for h in range(1 << s_lo):
        for i in range(S):
            x = [a[h * Bsz + j * S + i] for j in range(1 << M)]
            for t in range(M):
                s = s_lo + t
                half = 1 << (M - 1 - t)
                for j in range(1 << M):
                    if j & half:
                        continue
                    b = (h << t) + (j >> (M - t))
                    w = T["tw"][(1 << s) + b]
                    u, v = x[j], x[j + half] * w % p
                    x[j], x[j + half] = (u + v) % p, (u - v) % p
            for j in range(1 << M):
                v = x[j]
                if twist:
                    bfin = (h << M) + j
                    v = v * T["twist_f"][bfin][i] % p
                a[h * Bsz + j * S + i] = v
    return a


def strided_inv(a, k, s_lo, M, T, twist):
    D = 1 << k
    a = list(a)
    Bsz = D >> s_lo
    S = Bsz >> M
    for h in range(1 << s_lo):
        for i in range(S):
            x = [a[h * Bsz + j * S + i] for j in range(1 << M)]
            if twist:
                x = [x[j] * T["twist_i"][(h << M) + j][i] % p for j in range(1 << M)]
            for t in range(M - 1, -1, -1):
                s = s_lo + t
                half = 1 << (M - 1 - t)
                for j in range(1 << M):
                    if j & half:
                        continue
                    b = (h << t) + (j >> (M - t))
                    w = T["itw"][(1 << s) + b]
                    u, v = x[j], x[j + half]
                    x[j], x[j + half] = (u + v) % p, (u - v) * w % p
            for j in range(1 << M):
                a[h * Bsz + j * S + i] = x[j]
    return a


def rows_fwd(tile, T):
    """cyclic DFT_4096, bit-reversed output, exactly the three register passes of the kernel."""
    y = list(tile)
    # pass 1: thread t = i0, slots i1 at i1*256 + t
    for t in range(256):
        x = dft16_fwd([y[j * 256 + t] for j in range(16)])
        for r in range(16):
            y[r * 256 + t] = x[r] * T["W1f"][r][t] % p
    # pass 2: thread t = (rho, i0)
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        x = dft16_fwd([y[rho * 256 + j * 16 + i0] for j in range(16)])
        for s in range(16):
            y[rho * 256 + s * 16 + i0] = x[s] * T["W2f"][s][i0] % p
    # pass 3: 16 contiguous
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_fwd(y[16 * t:16 * t + 16])
    return y


def rows_inv(tile, T):
    y = list(tile)
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_inv(y[16 * t:16 * t + 16])
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        x = dft16_inv([y[rho * 256 + s * 16 + i0] * T["W2i"][s][i0] % p for s in range(16)])
        for j in range(16):
            y[rho * 256 + j * 16 + i0] = x[j]
    for t in range(256):
        x = dft16_inv([y[r * 256 + t] * T["W1i"][r][t] % p for r in range(16)])
        for j in range(16):
            y[j * 256 + t] = x[j]
    return y


def fast_fwd(a, k, T):
    c = k - 12
    a = strided_fwd(a, k, 0, c, T, True)
    out = []
    for b in range(1 << c):
        out += rows_fwd(a[b * 4096:(b + 1) * 4096], T)
    return out


def fast_inv(A, k, T):
    c = k - 12
    a = []
    for b in range(1 << c):
        a += rows_inv(A[b * 4096:(b + 1) * 4096], T)
    a = strided_inv(a, k, 0, c, T, True)
    s = T["cols_scale"]
    return [v * s % p for v in a]


# ---------------------------------------------------------------- D = 4096 >> q <= 4096: 2^q ring elements per tile
def dft16_fwd_q(x, q):
    """the last 4 - q DIF stages only: 2^q independent cyclic DFTs of size 16 >> q on consecutive groups"""
    x = list(x)
    half, step = 8, W16_EXP
    while half >= 1:
        if half <= (8 >> q):
            for base in range(0, 16, 2 * half):
                for j in range(half):
                    a, b = x[base + j], x[base + j + half]
                    x[base + j] = (a + b) % p
                    x[base + j + half] = (a - b) * pow(2, (step * j) % 192, p) % p
        half //= 2
        step = (step * 2) % 192
    return x


def dft16_inv_q(x, q):
    x = list(x)
    steps = {1: (W16_EXP * 8) % 192, 2: (W16_EXP * 4) % 192, 4: (W16_EXP * 2) % 192, 8: W16_EXP}
    half = 1
    while half <= (8 >> q):
        step = steps[half]
        for base in range(0, 16, 2 * half):
            for j in range(half):
                u, v = x[base + j], x[base + j + half] * pow(2, (192 - (step * j) % 192) % 192, p) % p
                x[base + j] = (u + v) % p
                x[base + j + half] = (u - v) % p
        half *= 2
    return x


def twist_exp(q):
    """psi^256 = 2^twist_exp(q) for D = 4096 >> q:  psi^(2D/64) = omega_64 = 2^39, so psi^256 = 2^(39 * 2^(q+1))"""
    return (39 << (q + 1)) % 192


def small_tables(k):
    q = 12 - k
    D = 1 << k
    psi = P.psi("goldilocks", k)
    wD = pow(psi, 2, p)                      # omega_D: the cyclic size the stride-256 pass starts is D itself
    w256 = pow(psi, 2 * D // 256, p)
    T = {"q": q, "D": D}
    assert pow(psi, 256, p) == pow(2, twist_exp(q), p)
    inv = lambda v: pow(v, -1, p)
    m0 = lambda r: brv(r & ((1 << (4 - q)) - 1), 4 - q)
    # the column part psi^t of the twist psi^(256 j' + t) commutes with the stride-256 sub-DFT: merged into W1
    T["W1f"] = [[pow(wD, i0 * m0(r), p) * pow(psi, i0, p) % p for i0 in range(256)] for r in range(16)]
    T["W2f"] = [[pow(w256, i0 * brv(s, 4), p) for i0 in range(16)] for s in range(16)]
    T["W1i"] = [[inv(v) * inv(D) % p for v in row] for row in T["W1f"]]   # carries psi^-t and D^-1
    T["W2i"] = [[inv(v) for v in row] for row in T["W2f"]]
    return T


def small_fwd(tile, T):
    """tile: 4096 coefficients = 2^q ring elements of degree D, element-major; returns their transforms"""
    q, D = T["q"], T["D"]
    y = list(tile)
    E = twist_exp(q)
    for t in range(256):
        # row part of the twist: (psi^256)^j' = 2^(E j'), j' = row index inside the ring element: shifts
        x = [y[j * 256 + t] * pow(2, (E * (j & ((16 >> q) - 1))) % 192, p) % p for j in range(16)]
        x = dft16_fwd_q(x, q)
        for r in range(16):
            y[r * 256 + t] = x[r] * T["W1f"][r][t] % p
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        x = dft16_fwd([y[rho * 256 + j * 16 + i0] for j in range(16)])
        for s_ in range(16):
            y[rho * 256 + s_ * 16 + i0] = x[s_] * T["W2f"][s_][i0] % p
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_fwd(y[16 * t:16 * t + 16])
    return y


def small_inv(tile, T):
    q, D = T["q"], T["D"]
    y = list(tile)
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_inv(y[16 * t:16 * t + 16])
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        x = dft16_inv([y[rho * 256 + s_ * 16 + i0] * T["W2i"][s_][i0] % p for s_ in range(16)])
        for j in range(16):
            y[rho * 256 + j * 16 + i0] = x[j]
    for t in range(256):
        x = dft16_inv_q([y[r * 256 + t] * T["W1i"][r][t] % p for r in range(16)], q)
        E = twist_exp(q)
        for j in range(16):
            y[j * 256 + t] = x[j] * pow(2, (192 - (E * (j & ((16 >> q) - 1))) % 192) % 192, p) % p
    return y


# ---------------------------------------------------------------- D = 256 * N2, N2 >= 256: one 8-stage column pass (cols256)
THETA_EXP_NOTE = "theta = psi^(D/256) = 7^((p-1)/512) does not depend on D"


def cols_tables(k):
    """tables of the 8-stage column pass and the twist it leaves for N2-point cyclic rows, D = 256 * N2"""
    D = 1 << k
    N2 = D >> 8
    psi = P.psi("goldilocks", k)
    theta = pow(psi, N2, p)
    assert theta == pow(7, (p - 1) // 512, p)
    inv = lambda v: pow(v, -1, p)
    T = {"k": k, "N2": N2}
    T["tw"] = [pow(psi, brv(i, k), p) for i in range(32)]          # stages 0..3 only: 4th .. 32nd roots of unity = shifts
    T["tw_exp"] = []
    for v in T["tw"]:
        e = [e for e in range(192) if pow(2, e, p) == v]
        T["tw_exp"].append(e[0] if e else None)
    assert all(e is not None for e in T["tw_exp"][1:16])           # index 2^u + b, u < 4: every one a power of two
    # W layer between the two register passes: block h (after 4 stages), leg rg: gamma'_h^(rg N2), gamma'_h = psi^(2 brv4(h) + 1)
    T["Wcf"] = [[pow(theta, (2 * brv(h, 4) + 1) * rg, p) for rg in range(16)] for h in range(16)]
    T["Wci"] = [[inv(v) for v in row] for row in T["Wcf"]]
    T["twist_f"] = [[pow(psi, (2 * brv(b, 8) + 1) * i, p) for i in range(N2)] for b in range(256)]
    T["twist_i"] = [[inv(v) * inv(D) % p for v in row] for row in T["twist_f"]]   # carries D^-1: every network is unnormalised
    w256 = pow(psi, 2 * D // 256, p)
    T["W2f"] = [[pow(w256, i0 * brv(s_, 4), p) for i0 in range(16)] for s_ in range(16)]
    T["W2i"] = [[inv(v) for v in row] for row in T["W2f"]]
    return T


def cols256_fwd(a, T):
    k, N2 = T["k"], T["N2"]
    a = list(a)
    for i2 in range(N2):
        col = [a[i1 * N2 + i2] for i1 in range(256)]
        # pass A: lane rg holds legs rg + 16 jj; stages 0..3 of the merged negacyclic network, twiddles = compile-time shifts
        for rg in range(16):
            x = [col[rg + 16 * jj] for jj in range(16)]
            for u in range(4):
                half = 8 >> u
                for jj in range(16):
                    if jj & half:
                        continue
                    e = T["tw_exp"][(1 << u) + (jj >> (4 - u))]
                    t = x[jj + half] * pow(2, e, p) % p
                    x[jj], x[jj + half] = (x[jj] + t) % p, (x[jj] - t) % p
            for h in range(16):
                col[rg + 16 * h] = x[h] * T["Wcf"][h][rg] % p      # block h, leg rg: position 16 h + rg ("s*16 + rg")
        # pass B: lane rg' = block h holds its 16 legs: cyclic DFT_16, bit-reversed out; then the twist of final block b = 16 h + sigma
        for h in range(16):
            x = dft16_fwd([col[16 * h + rg] for rg in range(16)])
            for sg in range(16):
                b = 16 * h + sg
                a[b * N2 + i2] = x[sg] * T["twist_f"][b][i2] % p
    return a


def cols256_inv(a, T):
    k, N2 = T["k"], T["N2"]
    a = list(a)
    for i2 in range(N2):
        col = [0] * 256
        for h in range(16):
            x = dft16_inv([a[(16 * h + sg) * N2 + i2] * T["twist_i"][16 * h + sg][i2] % p for sg in range(16)])
            for rg in range(16):
                col[16 * h + rg] = x[rg]
        for rg in range(16):
            x = [col[rg + 16 * h] * T["Wci"][h][rg] % p for h in range(16)]
            for u in range(3, -1, -1):
                half = 8 >> u
                for jj in range(16):
                    if jj & half:
                        continue
                    e = (192 - T["tw_exp"][(1 << u) + (jj >> (4 - u))]) % 192
                    s_, d = (x[jj] + x[jj + half]) % p, (x[jj] - x[jj + half]) * pow(2, e, p) % p
                    x[jj], x[jj + half] = s_, d
            for jj in range(16):
                a[(rg + 16 * jj) * N2 + i2] = x[jj]
    return a


def rows256_fwd(row, T):
    """cyclic DFT_256, natural in, bit-reversed out: passes 2 and 3 of rows_fwd"""
    y = list(row)
    for i0 in range(16):
        x = dft16_fwd([y[j * 16 + i0] for j in range(16)])
        for s_ in range(16):
            y[s_ * 16 + i0] = x[s_] * T["W2f"][s_][i0] % p
    for t in range(16):
        y[16 * t:16 * t + 16] = dft16_fwd(y[16 * t:16 * t + 16])
    return y


def rows256_inv(row, T):
    y = list(row)
    for t in range(16):
        y[16 * t:16 * t + 16] = dft16_inv(y[16 * t:16 * t + 16])
    for i0 in range(16):
        x = dft16_inv([y[s_ * 16 + i0] * T["W2i"][s_][i0] % p for s_ in range(16)])
        for j in range(16):
            y[j * 16 + i0] = x[j]
    return y


# ------------------------------------------------------------------------------------------------
# Whole tiles on representatives, as the kernels run them (round 3): lazy networks, table products that SKIP index 0 (factor 1),
# and G::canon on the one value per lane that reaches a network without a product in front of it although it is an arbitrary slot
# of another lane's lazy network (inverse direction only: in the forward direction the skipped value is slot 0 of the lane's own
# network, canonical by construction).  canon=False reproduces the tile without that step: the twiddle-1 butterflies then see
# non-canonical inputs (canon_bf asserts) on inputs built to make a lazy sum land on p exactly -- see crafted_inverse_block.
# ------------------------------------------------------------------------------------------------
def rows256_fwd_lazy(row, T, last_canonical=False):
    y = list(row)
    for i0 in range(16):
        x = dft16_fwd_dit([y[j * 16 + i0] for j in range(16)], lazy=True)
        y[i0] = x[0]                                                    # slot 0: no product
        assert x[0] < p
        for s_ in range(1, 16):
            y[s_ * 16 + i0] = x[s_] * T["W2f"][s_][i0] % p
    for t in range(16):
        y[16 * t:16 * t + 16] = dft16_fwd_dit(y[16 * t:16 * t + 16], lazy=not last_canonical)
    return y


def rows256_inv_lazy(row, T, canon=True):
    y = list(row)
    for t in range(16):
        y[16 * t:16 * t + 16] = dft16_inv_lazy(y[16 * t:16 * t + 16])
    for i0 in range(16):
        xs = [y[i0]] + [y[s_ * 16 + i0] * T["W2i"][s_][i0] % p for s_ in range(1, 16)]
        if canon:
            xs[0] %= p                                                  # G::canon
        x = dft16_inv_lazy(xs)
        for j in range(16):
            y[j * 16 + i0] = x[j]
    return y


def rows_fwd_lazy(tile, T, last_canonical=False):
    y = list(tile)
    for t in range(256):
        x = dft16_fwd_dit([y[j * 256 + t] for j in range(16)], lazy=True)
        y[t] = x[0]
        for r in range(1, 16):
            y[r * 256 + t] = x[r] * T["W1f"][r][t] % p
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        x = dft16_fwd_dit([y[rho * 256 + j * 16 + i0] for j in range(16)], lazy=True)
        y[rho * 256 + i0] = x[0]
        for s_ in range(1, 16):
            y[rho * 256 + s_ * 16 + i0] = x[s_] * T["W2f"][s_][i0] % p
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_fwd_dit(y[16 * t:16 * t + 16], lazy=not last_canonical)
    return y


def rows_inv_lazy(tile, T, canon=True):
    y = list(tile)
    for t in range(256):
        y[16 * t:16 * t + 16] = dft16_inv_lazy(y[16 * t:16 * t + 16])
    for t in range(256):
        rho, i0 = t >> 4, t & 15
        xs = [y[rho * 256 + i0]] + [y[rho * 256 + s_ * 16 + i0] * T["W2i"][s_][i0] % p for s_ in range(1, 16)]
        if canon:
            xs[0] %= p
        x = dft16_inv_lazy(xs)
        for j in range(16):
            y[rho * 256 + j * 16 + i0] = x[j]
    for t in range(256):
        xs = [y[t]] + [y[r * 256 + t] * T["W1i"][r][t] % p for r in range(1, 16)]
        if canon:
            xs[0] %= p
        x = dft16_inv_lazy(xs)
        for j in range(16):
            y[j * 256 + t] = x[j]
    return y


def crafted_inverse_block(rng, slot):
    """16 canonical inputs of dft16_inv whose output `slot` is 0 mod p while the two legs of its last butterfly are not: the lazy sum
    leaves the representative p there (a + t = p exactly).  The network is linear: fix 15 inputs, solve for the last."""
    base = [rng.randrange(1, p) for _ in range(15)] + [0]
    unit = [0] * 15 + [1]
    f0 = dft16_inv(base)[slot]
    f1 = dft16_inv(unit)[slot]
    base[15] = (-f0) * pow(f1, -1, p) % p
    assert dft16_inv(base)[slot] == 0
    return base


def crafted_inverse_block256(rng, T, slots):
    """256 canonical inputs of one 256-block of an inverse tile (16 first-level networks, table products, 16 second-level networks)
    such that the SECOND-level network of lane i0 leaves 0 mod p with non-zero legs in its output slots[i0]: the value the third
    network of a 4096-point tile takes without a product in front.  Built backwards: second-level inputs by crafted_inverse_block,
    divided by the table factors, pulled through the inverse of the first-level network (dft16_fwd / 16)."""
    inv16 = pow(16, -1, p)
    y = [0] * 256
    for i0 in range(16):
        xs = crafted_inverse_block(rng, slots[i0])
        for s_ in range(16):
            y[s_ * 16 + i0] = xs[s_] * (pow(T["W2i"][s_][i0], -1, p) if s_ else 1) % p
    out = []
    for t in range(16):
        out += [v * inv16 % p for v in dft16_fwd(y[16 * t:16 * t + 16])]
    assert [v % p for v in sum((dft16_inv(out[16 * t:16 * t + 16]) for t in range(16)), [])] == y
    return out


def check_lazy_tiles(seed=21):
    rng = random.Random(seed)
    T16 = cols_tables(16)
    T12 = tables(13)
    # random tiles: lazy == canonical modulo p, both directions, both tile sizes
    row = [rng.randrange(p) for _ in range(256)]
    assert [v % p for v in rows256_fwd_lazy(row, T16)] == rows256_fwd(row, T16)
    assert rows256_fwd_lazy(row, T16, last_canonical=True) == rows256_fwd(row, T16)
    assert [v % p for v in rows256_inv_lazy(row, T16)] == rows256_inv(row, T16)
    tile = [rng.randrange(p) for _ in range(4096)]
    assert [v % p for v in rows_fwd_lazy(tile, T12)] == rows_fwd(tile, T12)
    assert rows_fwd_lazy(tile, T12, last_canonical=True) == rows_fwd(tile, T12)
    assert [v % p for v in rows_inv_lazy(tile, T12)] == rows_inv(tile, T12)
    # crafted: the 16-blocks that feed the product-free x[0] of the next network (lanes 16 m) leave the representative p in slots 1..15
    hits = 0
    for slot in range(1, 16):
        row = [rng.randrange(p) for _ in range(256)]
        row[0:16] = crafted_inverse_block(rng, slot)                    # lane 0: its slot `slot` is lane `slot`'s x[0]
        first = dft16_inv_lazy(row[0:16])
        if first[slot] == p:
            hits += 1
            try:
                rows256_inv_lazy(row, T16, canon=False)
                raise SystemExit("the uncanonicalised tile went through: the crafted input does not bite")
            except AssertionError:
                pass
        assert [v % p for v in rows256_inv_lazy(row, T16)] == rows256_inv(row, T16)
        tile = [rng.randrange(p) for _ in range(4096)]
        tile[0:16] = crafted_inverse_block(rng, slot)
        assert [v % p for v in rows_inv_lazy(tile, T12)] == rows_inv(tile, T12)
    assert hits >= 4, "crafted blocks never produced the representative p (%d)" % hits
    # second level (4096-point tiles): block 0 of the tile built so that the second-level networks leave p where the third takes it
    tile = [rng.randrange(p) for _ in range(4096)]
    tile[0:256] = crafted_inverse_block256(rng, T12, [1 + (i0 * 7) % 15 for i0 in range(16)])
    try:
        rows_inv_lazy(tile, T12, canon=False)
        raise SystemExit("the uncanonicalised 4096-point tile went through: the second-level crafted block does not bite")
    except AssertionError:
        pass
    assert [v % p for v in rows_inv_lazy(tile, T12)] == rows_inv(tile, T12)
    return True


def check_cols256(k=16, seed=5):
    rng = random.Random(seed)
    T = cols_tables(k)
    D, N2 = 1 << k, T["N2"]
    a = [rng.randrange(p) for _ in range(D)]
    want = P.pow2_fwd("goldilocks", a, k)
    mid = cols256_fwd(a, T)
    assert N2 == 256
    got = []
    for b in range(256):
        got += rows256_fwd(mid[b * 256:(b + 1) * 256], T)
    assert got == want, "cols256 forward mismatch"
    back = []
    for b in range(256):
        back += rows256_inv(got[b * 256:(b + 1) * 256], T)
    assert cols256_inv(back, T) == a, "cols256 inverse mismatch"
    return True


if __name__ == "__main__":
    rng = random.Random(3)
    # radix-16 networks
    x = [rng.randrange(p) for _ in range(16)]
    w = pow(2, W16_EXP, p)
    ref = [sum(x[i] * pow(w, i * m, p) for i in range(16)) % p for m in range(16)]
    got = dft16_fwd(x)
    assert got == [ref[brv(r, 4)] for r in range(16)]
    assert dft16_inv(got) == [16 * v % p for v in x]
    for k in (12, 13):
        T = tables(k)
        a = [rng.randrange(p) for _ in range(1 << k)]
        want = P.pow2_fwd("goldilocks", a, k)
        got = fast_fwd(a, k, T)
        assert got == want, "forward mismatch k=%d" % k
        assert fast_inv(got, k, T) == a, "inverse mismatch k=%d" % k
        print("k=%d ok" % k)
    # two strided passes (c = 2 as 1 + 1) compose
    k = 14
    T = tables(k)
    a = [rng.randrange(p) for _ in range(1 << k)]
    one = strided_fwd(a, k, 0, 2, T, True)
    two = strided_fwd(strided_fwd(a, k, 0, 1, T, False), k, 1, 1, T, True)
    assert one == two
    back = strided_inv(strided_inv(one, k, 1, 1, T, True), k, 0, 1, T, False)
    assert back == strided_inv(one, k, 0, 2, T, True)
    for k in (8, 10, 11):
        T = small_tables(k)
        D = 1 << k
        tile = [rng.randrange(p) for _ in range(4096)]
        got = small_fwd(tile, T)
        for e in range(4096 // D):
            assert got[e * D:(e + 1) * D] == P.pow2_fwd("goldilocks", tile[e * D:(e + 1) * D], k), "small fwd k=%d" % k
        assert small_inv(got, T) == tile
        print("small k=%d ok" % k)
    check_cols256(16)
    print("cols256 + rows256 k=16 ok")
    check_lazy_networks()
    print("lazy DIT networks ok")
    check_lazy_tiles()
    print("lazy tiles ok")
    print("model OK")
