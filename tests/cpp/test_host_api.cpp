// C++ parity tests of the host-side mirror (include/stark_rings.hpp), shaped after the reference's own tests:
//   crt.rs:85-147            trait-level round trips, single and elementwise (all models)
//   flatten.rs:58-139        flatten / promote
//   stark_prime/mod.rs:161-177, goldilocks/mod.rs:231-247   NTT product == schoolbook product reduced mod Phi
//   goldilocks/ntt.rs:136 etc.  wrong length panics
// The expected values come from the oracle (oracle/sr_oracle.h), which this TEST links; the product does not.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <hip/hip_runtime_api.h>  // hipMalloc / hipMemcpy for the device-pointer wrappers (the mirror itself needs no HIP header)

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

    // Neg, Mul<primitive>, Add / Sub<primitive> (round 4; coeff_form.rs:270-278, 610-700 and test_primitive_ops :736-746; ntt_form.rs:191-203,
    // 373-505): one * 1 == one, one * 0 == zero, one + 1 == one + one, one - 1 == zero; a + (-a) == 0; crt commutes with all three
    {
        std::vector<uint64_t> img1(L), img0(L, 0), imgm1(L), one_el(d * L, 0), tmp(L);
        const uint64_t std1[4] = {1, 0, 0, 0};
        sro_to_mont(field, std1, img1.data(), 1);
        std::copy(img1.begin(), img1.end(), one_el.begin());
        RqPolyVec one(cfg, one_el), zero(cfg, std::vector<uint64_t>(d * L, 0));
        {
            RqPolyVec m1 = -RqPolyVec(cfg, one_el);                 // the image of -1 is coefficient 0 of -one
            std::copy(m1.words().begin(), m1.words().begin() + L, imgm1.begin());
        }
        RqPolyVec t = one;
        t *= img1;
        EXPECT(t == one);
        t *= img0;
        EXPECT(t == zero);
        RqPolyVec two = one;
        two += img1;
        RqPolyVec oo = one;
        oo += one;
        EXPECT(two == oo);
        two += imgm1;                                               // Sub passes the negated scalar
        two += imgm1;
        EXPECT(two == zero);
        RqPolyVec s = RqPolyVec(cfg, a);
        s += -RqPolyVec(cfg, a);
        EXPECT(s == RqPolyVec(cfg, std::vector<uint64_t>(a.size(), 0)));
        // through crt: scaling and negation commute with it; adding c to coefficient 0 is adding c to every slot
        std::vector<uint64_t> c7(L);
        const uint64_t std7[4] = {7, 0, 0, 0};
        sro_to_mont(field, std7, c7.data(), 1);
        RqPolyVec pa(cfg, a);
        pa *= c7;
        RqNTTVec na = RqPolyVec(cfg, a).elementwise_crt();
        na *= c7;
        EXPECT(std::move(pa).elementwise_crt() == na);
        RqPolyVec pb(cfg, a);
        pb += c7;
        RqNTTVec nb2 = RqPolyVec(cfg, a).elementwise_crt();
        nb2 += c7;
        EXPECT(std::move(pb).elementwise_crt() == nb2);
        EXPECT((-RqPolyVec(cfg, a)).elementwise_crt() == -(RqPolyVec(cfg, a).elementwise_crt()));
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
    // MulAssign<&R> for Matrix<R> / SparseMatrix<R> (matrix.rs:207-211, sparse_matrix.rs:303-307): every entry times one ring element;
    // (r M) v == r (M v) entry by entry
    {
        RqNTTVec R(cfg, elem(B, 1));
        MatrixNTT MR(cfg, n, m, A);
        MR *= R;
        for (size_t i = 0; i < n * m; i++) {
            RqNTTVec x(cfg, elem(A, i));
            x *= R;
            EXPECT(elem(MR.words(), i) == x.words());
        }
        auto ry = MR.checked_mul_vec(V);
        RqNTTVec scaled(cfg, y ? y->words() : std::vector<uint64_t>(n * w, 0));
        scaled.mul_assign_elem(R);
        EXPECT(ry.has_value() && ry->words() == scaled.words());
        SparseMatrixNTT SR(cfg, 3, m, rows);
        SR *= R;
        auto sry = SR.checked_mul_vec(V);
        RqNTTVec sscaled(cfg, sy ? sy->words() : std::vector<uint64_t>(3 * w, 0));
        sscaled.mul_assign_elem(R);
        EXPECT(sry.has_value() && sry->words() == sscaled.words());
        threw = false;
        try {
            MR *= RqNTTVec(cfg, uniform(field, 26, 2 * d));
        } catch (const std::length_error &) {
            threw = true;
        }
        EXPECT(threw);
    }
    // impl Sum / Product for RqNTT and RqPoly (ntt_form.rs:640-670, coeff_form.rs:507-537): the reference's left folds, element by element
    {
        RqNTTVec all(cfg, A);
        RqNTTVec acc(cfg, elem(A, 0)), prod(cfg, elem(A, 0));
        for (size_t i = 1; i < n * m; i++) {
            acc += RqNTTVec(cfg, elem(A, i));
            prod *= RqNTTVec(cfg, elem(A, i));
        }
        EXPECT(all.sum() == acc);
        EXPECT(all.product() == prod);
        RqNTTVec none(cfg, std::vector<uint64_t>());
        EXPECT(none.sum().words() == std::vector<uint64_t>(w, 0));
        RqNTTVec one_times(cfg, elem(A, 2));
        one_times *= none.product();   // one() is the multiplicative identity
        EXPECT(one_times.words() == elem(A, 2));
        RqPolyVec polys(cfg, std::vector<uint64_t>(A.begin(), A.begin() + 3 * w));
        RqPolyVec pacc(cfg, elem(A, 0));
        pacc *= RqPolyVec(cfg, elem(A, 1));
        pacc *= RqPolyVec(cfg, elem(A, 2));
        EXPECT(polys.product() == pacc);
        RqPolyVec sacc(cfg, elem(A, 0));
        sacc += RqPolyVec(cfg, elem(A, 1));
        sacc += RqPolyVec(cfg, elem(A, 2));
        EXPECT(polys.sum() == sacc);
    }
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

// round 3: the reference's whole basis type (b: u128, mod.rs:62), DecomposeToVec (mod.rs:119-161), GadgetDecompose / GadgetRecompose
// for Matrix<R>, SparseMatrix<R> and &[(R, usize)] (mod.rs:208-352)
static void gadget_wide_suite(sr_ring ring, int field, int log2d) {
    CyclotomicConfig cfg(ring, log2d);
    const size_t d = cfg.dimension(), w = cfg.words_per_elem(), batch = 5;
    auto a = uniform(field, 51, batch * d);
    RqPolyVec v(cfg, a);
    // a basis of 2^64 and more: one-limb fields have |x| < b / 2, so digit 0 is x and the rest are zero; Stark needs real digits
    const u128 big = ((u128)3 << 70) + 6;
    const size_t k = field == SRO_STARK ? 5 : 2;
    RqPolyVec dig = gadget_decompose(v, big, k);
    EXPECT(dig.len() == batch * k);
    EXPECT(gadget_recompose(dig, big, k) == v);
    if (field != SRO_STARK)
        for (size_t e = 0; e < batch; e++) {
            EXPECT(std::equal(a.begin() + e * w, a.begin() + (e + 1) * w, dig.words().begin() + e * k * w));
            for (size_t q = 0; q < w; q++) EXPECT(dig.words()[(e * k + 1) * w + q] == 0);
        }
    // the same digits through the 64-bit oracle for a basis below 2^64 passed as u128
    const size_t k16 = field == SRO_STARK ? 64 : 17;
    std::vector<uint64_t> want(batch * k16 * w);
    EXPECT(sro_decompose_balanced(field, a.data(), d, batch, 16, k16, want.data()) == 0);
    EXPECT(gadget_decompose(v, (u128)16, k16).words() == want);
    // decompose_to_vec: element e -> its own Vec of k digits
    auto vecs = decompose_to_vec(v, 16, k16);
    EXPECT(vecs.size() == batch);
    for (size_t e = 0; e < batch && e < vecs.size(); e++) {
        EXPECT(vecs[e].len() == k16);
        EXPECT(std::equal(vecs[e].words().begin(), vecs[e].words().end(), want.begin() + e * k16 * w));
        EXPECT(gadget_recompose(vecs[e], 16, k16) == RqPolyVec(cfg, std::vector<uint64_t>(a.begin() + e * w, a.begin() + (e + 1) * w)));
    }
    // 2^127 and above is a negative basis after the reference's `b as i128` (mod.rs:73): refused
    bool threw = false;
    try {
        gadget_decompose(v, (u128)1 << 127, 2);
    } catch (const std::runtime_error &) {
        threw = true;
    }
    EXPECT(threw);
    // Matrix<R>: 2 x 3 -> 2 x 3k, entry (r, c) -> (r, c k + j); recompose inverts
    const size_t n = 2, m = 3;
    auto mw = uniform(field, 52, n * m * d);
    MatrixPoly M(cfg, n, m, mw);
    MatrixPoly MD = M.gadget_decompose(16, k16);
    EXPECT(MD.nrows() == n && MD.ncols() == m * k16);
    for (size_t r = 0; r < n; r++)
        for (size_t c = 0; c < m; c++) {
            RqPolyVec one(cfg, std::vector<uint64_t>(mw.begin() + (r * m + c) * w, mw.begin() + (r * m + c + 1) * w));
            RqPolyVec dd = gadget_decompose(one, 16, k16);
            EXPECT(std::equal(dd.words().begin(), dd.words().end(), MD.words().begin() + (r * m * k16 + c * k16) * w));
        }
    EXPECT(MD.gadget_recompose(16, k16) == M);
    // SparseMatrix<R> / &[(R, usize)]: zero digits are dropped, columns become c k + j, recompose restores (element, c)
    std::vector<uint64_t> small(w, 0), e1(mw.begin(), mw.begin() + w);
    {
        std::vector<uint64_t> std5(4 * d, 0), img(4 * d, 0);
        for (size_t i = 0; i < d; i++) std5[i * cfg.limbs()] = 5;    // the constant-coefficient polynomial 5 + 5X + ...: one non-zero digit in basis 16
        sro_to_mont(field, std5.data(), img.data(), d);
        small.assign(img.begin(), img.begin() + w);
    }
    std::vector<SparseRow> rows(3);
    rows[0] = {{small, 1}, {e1, 4}};
    rows[2] = {{e1, 0}};
    SparseMatrixPoly S(cfg, 3, 6, rows);
    SparseMatrixPoly SD = S.gadget_decompose(16, k16);
    EXPECT(SD.ncols() == 6 * k16 && SD.nrows() == 3);
    EXPECT(SD.coeffs()[1].empty());
    EXPECT(!SD.coeffs()[0].empty() && SD.coeffs()[0][0].second == 1 * k16 && SD.coeffs()[0][0].first == small);   // 5 = digit 0, the rest dropped
    EXPECT(SD.coeffs()[0].size() >= 2 && SD.coeffs()[0][1].second >= 4 * k16);
    SparseMatrixPoly SR = SD.gadget_recompose(16, k16);
    EXPECT(SR.ncols() == 6 && SR.coeffs() == rows);
}

// Cyclotomic::rot / into_rot_iter (traits.rs:54-91; test_rot of goldilocks/mod.rs:249-262): the i-th item of the iterator is X^i x
static void rot_suite(sr_ring ring, int field, size_t D, int log2d, int trinomial) {
    CyclotomicConfig cfg(ring, log2d);
    EXPECT(cfg.dimension() == D);
    const size_t w = cfg.words_per_elem(), batch = 3;
    auto a = uniform(field, 61, batch * D);
    Rotation it = into_rot_iter(RqPolyVec(cfg, a));
    std::vector<uint64_t> cur = a, nxt(a.size());
    for (int i = 0; i < 5; i++) {
        RqPolyVec item = it.next();
        EXPECT(item.words() == cur);
        for (size_t e = 0; e < batch; e++) sro_rot(field, cur.data() + e * w, D, trinomial, nxt.data() + e * w);
        cur = nxt;
    }
}

// one context per device from one process (here: two contexts on device 0, the only GPU of the test box) and the device-pointer
// wrappers: the group's second context computes with the twiddle block it received by peer copy
static void group_and_device_suite() {
    ContextGroup grp(SR_RING_GOLDILOCKS_POW2, 13, {0, 0});
    EXPECT(grp.size() == 2);
    auto r0 = grp.shard_range(7, 0), r1 = grp.shard_range(7, 1);
    EXPECT(r0.first == 0 && r0.second == 4 && r1.first == 4 && r1.second == 3);
    const int log2d = 13;
    const size_t d = (size_t)1 << log2d, batch = 7;
    auto a = uniform(SRO_GOLDILOCKS, 71, batch * d), b = uniform(SRO_GOLDILOCKS, 72, batch * d);
    std::vector<uint64_t> want(batch * d), got(batch * d);
    sro_pow2_ring_mul_batch(SRO_GOLDILOCKS, want.data(), a.data(), b.data(), log2d, batch, 2);
    uint64_t *da = nullptr, *db = nullptr, *dout = nullptr;
    EXPECT(hipMalloc((void **)&da, batch * d * 8) == hipSuccess && hipMalloc((void **)&db, batch * d * 8) == hipSuccess &&
           hipMalloc((void **)&dout, batch * d * 8) == hipSuccess);
    EXPECT(hipMemcpy(da, a.data(), batch * d * 8, hipMemcpyHostToDevice) == hipSuccess);
    EXPECT(hipMemcpy(db, b.data(), batch * d * 8, hipMemcpyHostToDevice) == hipSuccess);
    for (size_t i = 0; i < grp.size(); i++) {   // each context multiplies ITS shard
        auto sh = grp.shard_range(batch, i);
        grp[i].reserve_scratch(sh.second);
        grp[i].mul_dev(dout + sh.first * d, da + sh.first * d, db + sh.first * d, sh.second, nullptr);
    }
    EXPECT(hipDeviceSynchronize() == hipSuccess);
    EXPECT(hipMemcpy(got.data(), dout, batch * d * 8, hipMemcpyDeviceToHost) == hipSuccess);
    EXPECT(got == want);
    // crt / icrt / slot product on device pointers
    const CyclotomicConfig &cfg = grp[1];
    cfg.elementwise_crt_dev(da, batch, nullptr);
    cfg.elementwise_crt_dev(db, batch, nullptr);
    cfg.ntt_mul_dev(da, db, batch, nullptr);
    cfg.elementwise_icrt_dev(da, batch, nullptr);
    EXPECT(hipDeviceSynchronize() == hipSuccess);
    EXPECT(hipMemcpy(got.data(), da, batch * d * 8, hipMemcpyDeviceToHost) == hipSuccess);
    EXPECT(got == want);
    sr_plan pl = cfg.plan_in_use();
    EXPECT(pl.lanes <= 2);
    (void)hipFree(da);
    (void)hipFree(db);
    (void)hipFree(dout);
    // packed-u32 BabyBear boundary through the mirror
    CyclotomicConfig bb(SR_RING_BABYBEAR_POW2, 12);
    const size_t n = 3 * 4096;
    auto x = uniform(SRO_BABYBEAR, 73, n), y = uniform(SRO_BABYBEAR, 74, n);
    std::vector<uint64_t> wantb(n), gotb(n);
    sro_pow2_ring_mul_batch(SRO_BABYBEAR, wantb.data(), x.data(), y.data(), 12, 3, 2);
    uint64_t *dx = nullptr, *dy = nullptr;
    uint32_t *px = nullptr, *py = nullptr;
    EXPECT(hipMalloc((void **)&dx, n * 8) == hipSuccess && hipMalloc((void **)&dy, n * 8) == hipSuccess &&
           hipMalloc((void **)&px, n * 4) == hipSuccess && hipMalloc((void **)&py, n * 4) == hipSuccess);
    EXPECT(hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice) == hipSuccess);
    bb.pack32_dev(px, dx, 3, nullptr);
    bb.pack32_dev(py, dy, 3, nullptr);
    bb.mul_packed32_dev(px, px, py, 3, nullptr);
    bb.unpack32_dev(dx, px, 3, nullptr);
    EXPECT(hipDeviceSynchronize() == hipSuccess);
    EXPECT(hipMemcpy(gotb.data(), dx, n * 8, hipMemcpyDeviceToHost) == hipSuccess);
    EXPECT(gotb == wantb);
    (void)hipFree(dx);
    (void)hipFree(dy);
    (void)hipFree(px);
    (void)hipFree(py);
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
        gadget_wide_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 5);
        gadget_wide_suite(SR_RING_BABYBEAR_POW2, SRO_BABYBEAR, 4);
        gadget_wide_suite(SR_RING_STARK_POW2, SRO_STARK, 3);
        rot_suite(SR_RING_GOLDILOCKS_POW2, SRO_GOLDILOCKS, 64, 6, 0);
        rot_suite(SR_RING_STARK_POW2, SRO_STARK, 16, 4, 0);               // stark_prime/mod.rs:179-192
        rot_suite(SR_RING_GOLDILOCKS_24, SRO_GOLDILOCKS, 24, 0, 1);       // goldilocks/mod.rs:249-262 (X^24 - X^12 + 1)
        rot_suite(SR_RING_FROG_16, SRO_FROG, 16, 0, 0);
        group_and_device_suite();
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
