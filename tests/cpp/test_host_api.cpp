// C++ parity tests of the host-side mirror (include/stark_rings.hpp), shaped after the reference's own tests:
//   crt.rs:85-147            trait-level round trips, single and elementwise (all models)
//   flatten.rs:58-139        flatten / promote
//   stark_prime/mod.rs:161-177, goldilocks/mod.rs:231-247   NTT product == schoolbook product reduced mod Phi
//   goldilocks/ntt.rs:136 etc.  wrong length panics
// The expected values come from the oracle (oracle/sr_oracle.h), which this TEST links; the product does not.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/stark_rings.hpp"
#include "../../oracle/sr_oracle.h"

using namespace stark_rings;

static int failures = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);    \
            failures++;                                                    \
        }                                                                  \
    } while (0)

static std::vector<uint64_t> uniform(int field, uint64_t seed, size_t n_coeffs) {
    std::vector<uint64_t> v(n_coeffs * sro_limbs(field));
    sro_fill_uniform(field, seed, 0, n_coeffs, v.data());
    return v;
}

static void pow2_suite(sr_ring ring, int field, int log2d, size_t batch) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t d = cfg.dimension(), L = cfg.limbs();
    EXPECT(d == (size_t)1 << log2d);
    auto a = uniform(field, 11, batch * d), b = uniform(field, 12, batch * d);

    // test_crt_trait_vec_conversion_* (crt.rs:103-121)
    RqPolyVec orig(cfg, a);
    RqNTTVec ntt = RqPolyVec(cfg, a).elementwise_crt();
    std::vector<uint64_t> want = a;
    sro_pow2_fwd_batch(field, want.data(), log2d, batch, 2);
    EXPECT(ntt.words() == want);
    RqPolyVec back = std::move(ntt).elementwise_icrt();
    EXPECT(back == orig);

    // single element through CyclotomicConfig::crt / icrt (ring_config.rs:29-30)
    std::vector<uint64_t> one(a.begin(), a.begin() + d * L);
    auto c1 = cfg.crt(one);
    EXPECT(std::equal(c1.begin(), c1.end(), want.begin()));
    EXPECT(cfg.icrt(c1) == one);

    // test_mul_crt: ntt_form_1 * ntt_form_2 then icrt == coeff_1 * coeff_2 == schoolbook reduced
    RqNTTVec n1 = RqPolyVec(cfg, a).elementwise_crt(), n2 = RqPolyVec(cfg, b).elementwise_crt();
    RqPolyVec via_ntt = (std::move(n1) * n2).elementwise_icrt();
    RqPolyVec direct = RqPolyVec(cfg, a) * RqPolyVec(cfg, b);
    EXPECT(via_ntt == direct);
    if (d <= 256) {
        std::vector<uint64_t> sb((2 * d - 1) * L);
        sro_schoolbook(field, a.data(), b.data(), d, sb.data());
        cfg.reduce_in_place(sb);  // test_reduce (stark_prime/mod.rs:139-159)
        EXPECT(sb.size() == d * L);
        EXPECT(std::equal(sb.begin(), sb.end(), direct.words().begin()));
    }

    // crt is linear: crt(a + b) == crt(a) + crt(b); (a + b) - b == a   (Add / Sub of both forms)
    {
        RqPolyVec sum(cfg, a);
        sum += RqPolyVec(cfg, b);
        RqNTTVec lhs = std::move(sum).elementwise_crt();
        RqNTTVec rhs = RqPolyVec(cfg, a).elementwise_crt();
        rhs += RqPolyVec(cfg, b).elementwise_crt();
        EXPECT(lhs == rhs);
        rhs -= RqPolyVec(cfg, b).elementwise_crt();
        EXPECT(std::move(rhs).elementwise_icrt() == orig);
    }

    // flatten / promote (flatten.rs:128-138)
    auto flat = flatten_to_coeffs(RqPolyVec(cfg, a));
    EXPECT(flat == a);
    auto promoted = promote_from_coeffs<RqPolyVec>(cfg, flat);
    EXPECT(promoted.has_value() && *promoted == orig);
    flat.pop_back();
    EXPECT(!promote_from_coeffs<RqPolyVec>(cfg, flat).has_value());

    // "Panics if coefficients.len() != D"
    bool threw = false;
    try {
        std::vector<uint64_t> bad(d * L + L);
        cfg.crt_in_place(bad.data(), bad.size());
    } catch (const std::length_error &) {
        threw = true;
    }
    EXPECT(threw);
}

static void small_suite(sr_ring ring, int field, int D, int ext, void (*crt)(uint64_t *), void (*mul)(uint64_t *, const uint64_t *),
                        void (*icrt)(uint64_t *)) {
    CyclotomicConfig cfg(ring);
    EXPECT(cfg.dimension() == (size_t)D);
    EXPECT(cfg.crt_field_extension_degree() == ext);
    const size_t batch = 100;  // crt.rs:106-121 uses 100 x 100
    auto a = uniform(field, 21, batch * D), b = uniform(field, 22, batch * D);
    RqNTTVec na = RqPolyVec(cfg, a).elementwise_crt();
    RqNTTVec nb = RqPolyVec(cfg, b).elementwise_crt();
    std::vector<uint64_t> wa = a, wb = b;
    for (size_t e = 0; e < batch; e++) {
        crt(wa.data() + e * D);
        crt(wb.data() + e * D);
    }
    EXPECT(na.words() == wa);
    RqNTTVec prod = na * nb;
    for (size_t e = 0; e < batch; e++) mul(wa.data() + e * D, wb.data() + e * D);
    EXPECT(prod.words() == wa);
    RqPolyVec c = std::move(prod).elementwise_icrt();
    for (size_t e = 0; e < batch; e++) icrt(wa.data() + e * D);
    EXPECT(c.words() == wa);
    EXPECT(c == RqPolyVec(cfg, a) * RqPolyVec(cfg, b));
    EXPECT(std::move(na).elementwise_icrt() == RqPolyVec(cfg, a));
}

// linear algebra over RqNTT: test_matrix_mul_vec / mul_mat (linear_algebra/src/matrix.rs:233-262), sparse mul_vec
// (sparse_matrix.rs tests), against sums of the oracle's slot products via RqNTTVec arithmetic on single elements
static void linalg_suite(sr_ring ring, int field, int log2d) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t w = cfg.words_per_elem(), d = cfg.dimension();
    const size_t n = 3, m = 4, p = 2;
    auto A = uniform(field, 21, n * m * d), B = uniform(field, 22, m * p * d), v = uniform(field, 23, m * d);
    auto elem = [&](const std::vector<uint64_t> &src, size_t i) { return std::vector<uint64_t>(src.begin() + i * w, src.begin() + (i + 1) * w); };
    auto dot = [&](const std::vector<std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> &terms) {
        RqNTTVec acc(cfg, std::vector<uint64_t>(w, 0));
        for (auto &t : terms) {
            RqNTTVec x(cfg, t.first);
            x *= RqNTTVec(cfg, t.second);
            acc += x;
        }
        return acc.words();
    };
    MatrixNTT MA(cfg, n, m, A), MB(cfg, m, p, B);
    RqNTTVec V(cfg, v);
    auto y = MA.checked_mul_vec(V);
    EXPECT(y.has_value());
    for (size_t r = 0; r < n && y; r++) {
        std::vector<std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> terms;
        for (size_t c = 0; c < m; c++) terms.push_back({elem(A, r * m + c), elem(v, c)});
        EXPECT(elem(y->words(), r) == dot(terms));
    }
    EXPECT(!MA.checked_mul_vec(RqNTTVec(cfg, uniform(field, 24, (m + 1) * d))).has_value());   // DifferentLengths -> None
    bool threw = false;
    try {
        MA.try_mul_vec(RqNTTVec(cfg, uniform(field, 24, (m - 1) * d)));
    } catch (const std::length_error &) {
        threw = true;
    }
    EXPECT(threw);
    auto Y = MA.checked_mul_mat(MB);
    EXPECT(Y.has_value() && Y->nrows() == n && Y->ncols() == p);
    for (size_t i = 0; i < n && Y; i++)
        for (size_t j = 0; j < p; j++) {
            std::vector<std::pair<std::vector<uint64_t>, std::vector<uint64_t>>> terms;
            for (size_t t = 0; t < m; t++) terms.push_back({elem(A, i * m + t), elem(B, t * p + j)});
            EXPECT(elem(Y->words(), i * p + j) == dot(terms));
        }
    EXPECT(!MB.checked_mul_mat(MB).has_value());
    // sparse: row 0 two entries (one column twice), row 1 empty, row 2 one entry
    std::vector<std::vector<SparseMatrixNTT::Entry>> rows(3);
    rows[0] = {{elem(A, 0), 1}, {elem(A, 1), 3}, {elem(A, 2), 1}};
    rows[2] = {{elem(A, 5), 0}};
    SparseMatrixNTT S(cfg, 3, m, rows);
    auto sy = S.checked_mul_vec(V);
    EXPECT(sy.has_value());
    if (sy) {
        EXPECT(elem(sy->words(), 0) == dot({{elem(A, 0), elem(v, 1)}, {elem(A, 1), elem(v, 3)}, {elem(A, 2), elem(v, 1)}}));
        EXPECT(elem(sy->words(), 1) == std::vector<uint64_t>(w, 0));
        EXPECT(elem(sy->words(), 2) == dot({{elem(A, 5), elem(v, 0)}}));
    }
    EXPECT(!S.checked_mul_vec(RqNTTVec(cfg, uniform(field, 25, (m + 2) * d))).has_value());
    rows[1] = {{elem(A, 3), m}};  // column out of range: the reference panics on v[col]
    threw = false;
    try {
        SparseMatrixNTT(cfg, 3, m, rows).checked_mul_vec(V);
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
}

// test_gadget_decompose / test_gadget_recompose (balanced_decomposition/mod.rs:469-514) and parity with the oracle
static void gadget_suite(sr_ring ring, int field, int log2d) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t d = cfg.dimension(), L = cfg.limbs(), batch = 7, k = field == SRO_STARK ? 64 : 17;
    auto a = uniform(field, 31, batch * d);
    RqPolyVec v(cfg, a);
    RqPolyVec dig = gadget_decompose(v, 16, k);
    std::vector<uint64_t> want(batch * k * d * L);
    EXPECT(sro_decompose_balanced(field, a.data(), d, batch, 16, k, want.data()) == 0);
    EXPECT(dig.words() == want);
    EXPECT(gadget_recompose(dig, 16, k) == v);
    bool threw = false;
    try {
        gadget_decompose(v, 7, k);  // "decomposition basis must be even"
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
    threw = false;
    try {
        gadget_decompose(v, 2, 8);  // out[i] out of bounds in the reference
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
}

// CanonicalSerialize / CanonicalDeserialize round trips and parity with the oracle's restatement of the ark-serialize bytes
// (coeff_form.rs:154-189, matrix.rs:111-145, sparse_matrix.rs:158-200)
static void wire_suite(sr_ring ring, int field, int log2d) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t d = cfg.dimension(), L = cfg.limbs(), batch = 5, wb = sro_wire_bytes(field);
    EXPECT(cfg.wire_coeff_bytes() == wb);
    auto a = uniform(field, 41, batch * d);
    RqPolyVec v(cfg, a);
    std::vector<uint8_t> bytes = serialize_compressed(v);
    EXPECT(bytes.size() == 8 + batch * d * wb);
    EXPECT(bytes[0] == batch && bytes[1] == 0 && bytes[7] == 0);
    std::vector<uint8_t> want(batch * d * wb);
    sro_serialize(field, a.data(), batch * d, want.data());
    EXPECT(std::vector<uint8_t>(bytes.begin() + 8, bytes.end()) == want);
    EXPECT(deserialize_vec<RqPolyVec>(cfg, bytes) == v);
    // an integer >= p is InvalidData: all-ones bytes in one coefficient
    std::vector<uint8_t> bad = bytes;
    for (size_t b = 0; b < wb; b++) bad[8 + 3 * wb + b] = 0xFF;
    bool threw = false;
    try {
        deserialize_vec<RqPolyVec>(cfg, bad);
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
    threw = false;
    try {
        deserialize_vec<RqPolyVec>(cfg, std::vector<uint8_t>(bytes.begin(), bytes.end() - 1));   // short input
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
    // Matrix and SparseMatrix framing
    const size_t nrows = 2, ncols = 2;
    MatrixNTT m(cfg, nrows, ncols, std::vector<uint64_t>(a.begin(), a.begin() + nrows * ncols * d * L));
    std::vector<uint8_t> mb = serialize_compressed(m, cfg);
    EXPECT(mb.size() == 8 + nrows * (8 + ncols * d * wb));
    MatrixNTT m2 = deserialize_matrix(cfg, mb);
    EXPECT(m2.nrows() == nrows && m2.ncols() == ncols && m2.words() == m.words());
    std::vector<std::vector<SparseMatrixNTT::Entry>> rows(3);
    rows[0].emplace_back(std::vector<uint64_t>(a.begin(), a.begin() + d * L), 4);
    rows[2].emplace_back(std::vector<uint64_t>(a.begin() + d * L, a.begin() + 2 * d * L), 0);
    rows[2].emplace_back(std::vector<uint64_t>(a.begin() + 2 * d * L, a.begin() + 3 * d * L), 6);
    std::vector<uint8_t> sb = serialize_compressed(cfg, 3, 7, rows);
    EXPECT(sb.size() == 24 + 3 * 8 + 3 * (d * wb + 8));
    SparseParts sp = deserialize_sparse(cfg, sb);
    EXPECT(sp.nrows == 3 && sp.ncols == 7 && sp.coeffs == rows);
}

// test_monomial_ops / test_monomial_range_check (crates/ring/src/monomial.rs:100-137) on the ring they use (frog, D = 16), and
// on power-of-two rings of the other fields
static void monomial_suite(sr_ring ring, int field, int log2d) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t d = cfg.dimension(), L = cfg.limbs();
    const int64_t half = (int64_t)d / 2;
    RqPolyVec zero = zero_monomial(cfg), one = unit_monomial(cfg, 0);
    RqPolyVec s = zero;
    s += one;
    EXPECT(s == one);                                            // 0 + 1 = 1
    std::vector<uint64_t> std_one(4, 0), mont(4, 0);
    std_one[0] = 1;
    sro_to_mont(field, std_one.data(), mont.data(), 1);
    for (size_t l = 0; l < L; l++) EXPECT(one.words()[l] == mont[l]);
    if (d >= 16) {
        RqPolyVec x2 = monomial(cfg, 2, 1), x15 = monomial(cfg, d - 1, 1);
        RqPolyVec p = x2 * x15;                                  // X^2 * X^(d-1) = -X
        RqPolyVec minus_x = monomial(cfg, 1, -1);
        EXPECT(p == minus_x);
    }
    EXPECT(psi_range_check(cfg, 1));
    EXPECT(psi_range_check(cfg, half - 1));
    EXPECT(!psi_range_check(cfg, half));
    EXPECT(psi_range_check(cfg, -1));
    EXPECT(psi_range_check(cfg, -(half - 1)));
    EXPECT(!psi_range_check(cfg, -half));
    EXPECT(psi_range_check(cfg, 0));
    bool threw = false;
    try {
        psi_range_check(cfg, (int64_t)d + 3);                    // unit_monomial indexes past the array: panic in the reference
    } catch (const std::out_of_range &) {
        threw = true;
    }
    EXPECT(threw);
    EXPECT(exp_signed(cfg, -3) == monomial(cfg, 3, -1));
}

int main() {
    try {
        pow2_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 10, 3);   // BASELINE configs[0]: D = 2^10 (batch 1 is element 0)
        pow2_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 13, 2);
        pow2_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 8, 5);
        pow2_suite(SR_RING_STARK_POW2, SRO_STARK, 4, 100);            // the reference's stark_prime ring
        pow2_suite(SR_RING_STARK_POW2, SRO_STARK, 8, 3);
        gadget_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 6);
        gadget_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 5);
        gadget_suite(SR_RING_STARK_POW2, SRO_STARK, 4);
        linalg_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 6);
        linalg_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 5);
        linalg_suite(SR_RING_STARK_POW2, SRO_STARK, 4);
        wire_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 6);
        wire_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 5);
        wire_suite(SR_RING_STARK_POW2, SRO_STARK, 4);
        wire_suite(SR_RING_FROG_16, SRO_FROG, 0);
        monomial_suite(SR_RING_FROG_16, SRO_FROG, 0);                 // the reference's own test ring
        monomial_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 6);
        monomial_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 5);
        monomial_suite(SR_RING_STARK_POW2, SRO_STARK, 4);
        small_suite(SR_RING_GOLDILOCKS_24, SRO_GOLDILOCKS, 24, 3, sro_g24_crt, sro_g24_ntt_mul, sro_g24_icrt);
        small_suite(SR_RING_BABYBEAR_72, SRO_BABYBEAR, 72, 9, sro_bb72_crt, sro_bb72_ntt_mul, sro_bb72_icrt);
        small_suite(SR_RING_FROG_16, SRO_FROG, 16, 4, sro_frog16_crt, sro_frog16_ntt_mul, sro_frog16_icrt);   // frog_ring/mod.rs tests
    } catch (const std::exception &e) {
        std::printf("EXCEPTION %s\n", e.what());
        return 2;
    }
    std::printf(failures ? "host api: %d FAILURES\n" : "host api: all ok\n", failures);
    return failures ? 1 : 0;
}
