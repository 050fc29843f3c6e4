"""The index-level Python model of the tuned Goldilocks decomposition (tools/model_fast_goldilocks.py) against the
oracle's pure-Python model: strided merged stages + twist + cyclic 16x16x16 with omega_16 = 2^156 == the reference's
negacyclic transform.  CPU only; this is the specification ntt_goldilocks.hpp was written from."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import model_fast_goldilocks as M
import pyref as P


def test_radix16_networks_are_the_cyclic_dft_with_omega_2_pow_156():
    rng = random.Random(3)
    p = P.GOLDILOCKS_P
    # omega_64 = 7^((p-1)/64) = 8^13  =>  omega_16 = 2^(12*13) = 2^156
    assert pow(7, (p - 1) // 64, p) == pow(8, 13, p)
    x = [rng.randrange(p) for _ in range(16)]
    w = pow(2, M.W16_EXP, p)
    ref = [sum(x[i] * pow(w, i * m, p) for i in range(16)) % p for m in range(16)]
    got = M.dft16_fwd(x)
    assert got == [ref[P.brv(r, 4)] for r in range(16)]
    assert M.dft16_inv(got) == [16 * v % p for v in x]


def test_decomposition_equals_reference_transform():
    rng = random.Random(4)
    p = P.GOLDILOCKS_P
    for k in (12, 13):
        T = M.tables(k)
        a = [rng.randrange(p) for _ in range(1 << k)]
        want = P.pow2_fwd("goldilocks", a, k)
        got = M.fast_fwd(a, k, T)
        assert got == want
        assert M.fast_inv(got, k, T) == a


def test_small_degree_mode_equals_reference_transform():
    """D = 4096 >> q: 2^q ring elements per tile, twist applied by the rows kernel, leading radix-16 stages skipped."""
    rng = random.Random(5)
    p = P.GOLDILOCKS_P
    for k in (8, 9, 11):
        T = M.small_tables(k)
        D = 1 << k
        tile = [rng.randrange(p) for _ in range(4096)]
        got = M.small_fwd(tile, T)
        for e in range(4096 // D):
            assert got[e * D:(e + 1) * D] == P.pow2_fwd("goldilocks", tile[e * D:(e + 1) * D], k)
        assert M.small_inv(got, T) == tile


def test_cols256_plus_rows256_equals_reference_transform():
    """D = 2^16 = 256 x 256: shift-only 8-stage column pass (compile-time twiddles 2^(39 brv5(i)), theta W layer, cyclic DFT_16,
    block twist) followed by 256-point cyclic rows."""
    assert M.check_cols256(16, seed=6)


def test_lazy_decimation_in_time_networks_keep_their_invariants():
    """Round 3: the forward DFT_16 as a decimation-in-time network equals the DIF one; with LAZY twiddled butterflies (six VALU:
    any 64-bit `a`, canonical shift product) no sum wraps twice, no difference borrows twice, twiddle-1 butterflies only ever see
    canonical inputs, slot 0 stays canonical, and every result is the canonical network's modulo p (representatives modelled bit
    for bit, edge values included)."""
    assert M.check_lazy_networks(seed=10, rounds=300)


def test_lazy_tiles_need_the_canonicalised_skip_value_and_have_it():
    """Whole 256-point and 4096-point tiles on representatives, table products skipping index 0 as the kernels do: equal to the
    canonical tiles modulo p in both directions; on operands built so that a lazy sum lands on p in the value that reaches the next
    inverse network without a product in front (first level: crafted_inverse_block, second level: crafted_inverse_block256) the tile
    WITHOUT G::canon trips the canonical-input assertion of a twiddle-1 butterfly, the tile with it is right."""
    assert M.check_lazy_tiles(seed=22)
