// StarkL: the Starknet prime p = 2^251 + 17 * 2^192 + 1 in nine signed 28-bit limbs with lazy carries -- the arithmetic the
// transform kernels of the Stark rings compute in (the memory image and every element-wise entry point stay `Stark`, fields.hpp).
//
// Why: with 8 x 32-bit limbs every partial product of a Montgomery multiplication needs its carry caught (v_mad_u64_u32 +
// v_addc_co_u32, 267 VALU instructions per product) and every add / sub is a carry chain plus a conditional correction.  In
// 28-bit limbs nine 56..59-bit partial products fit one 64-bit accumulator with room to spare, so a column is plain
// v_mad_i64_i32 with NO carry handling, and p's limbs are (1, 0, 0, 0, 0, 0, 2^24, 1, 2^27): the reduction needs no
// multiplications by p at all (three shifted adds per step, and -p^-1 = -1 mod 2^28).  Sums and differences are nine
// independent 32-bit adds (the quick-issue VALU class); carries and the reduction modulo p are postponed:
//
//   value(x) = sum_i l[i] * 2^(28 i),  l[i] signed.  Invariants: |l[i]| < 2^31 - 16 (int32), |value| < 16 p.
//   mul_tw(a, w) = a w 2^-280 (mod p), R = 2^(28 * 10): ten reduction steps over nine-limb operands, so the result is
//       < |a w| / 2^280 + p, i.e. in (-2^231, p + 2^231) whatever lazy state `a` was in; result limbs 0..7 in [0, 2^28), limb 8
//       small and possibly negative.  `w` is a table twiddle (limbs in [0, 2^28)) or a weakly reduced element.
//   add / sub: limb-wise, no carry: each grows the limb bound by the other operand's.
//   relax: one signed carry sweep, limbs 0..7 back into [0, 2^28).  fold: subtract floor(value / 2^251) * p (value back into
//       (-2^206, 2^251 + 2^206)); weak_reduce = relax + fold.  canonical: the unique representative in [0, p), limbs normalised.
//   Transform kernels (ntt_generic.hpp) call relax / weak_reduce where a chain of additions could outgrow the invariants:
//       a Cooley-Tukey stage adds at most 2^28 per limb and p in value (the v leg goes through mul_tw), so one weak_reduce
//       every six stages suffices; a Gentleman-Sande stage doubles its sum leg, which is weakly reduced every stage.
//   Data stays in the memory image's Montgomery form x 2^256 throughout; twiddles are in table form w 2^280, so
//       mul_tw(x 2^256, w 2^280) = x w 2^256.  The slot product of two data elements comes out as a b 2^232: the generic
//       set-up code folds the missing 2^24 into the inverse transform's scale constants (kappa_bits = 280 in capi.hip).
//
// Everything is plain C++ (host and device compile the same source); constants the compiler would otherwise turn into shift
// sequences are made opaque so that each reduction term is one v_mad_u64_u32.
// Checked against Python big integers (tests/test_host_fields.py via sr_selftest_field_op) and, through the kernels, by
// every Stark parity test.
#pragma once
#include "fields.hpp"

namespace sr {

struct S9 {
    int32_t l[9];
};

struct StarkL {
    using elem = S9;
    using storage = U256Storage;
    static constexpr int kStorageWords64 = 4;
    static constexpr int kLdsWords = 9;
    static constexpr uint32_t kGenerator = 3;
    static constexpr int kBoundaryBits = 256;
    static constexpr int kTwoAdicity = 192;
    static constexpr int kTableBits = 280;
    static constexpr uint32_t M28 = (1u << 28) - 1u;

    SR_HD static elem zero() {
        elem z;
#pragma unroll
        for (int i = 0; i < 9; i++) z.l[i] = 0;
        return z;
    }
    // memory image (4 little-endian u64 limbs, canonical) -> nine 28-bit limbs
    SR_HD static elem load(const storage *p) {
        uint32_t w[9];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t q = p->q[i];
            w[2 * i] = (uint32_t)q;
            w[2 * i + 1] = (uint32_t)(q >> 32);
        }
        w[8] = 0;
        elem e;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int bit = 28 * i, j = bit >> 5, sh = bit & 31;
            const uint64_t pair = ((uint64_t)w[j + 1] << 32) | w[j];
            e.l[i] = (int32_t)((uint32_t)(pair >> sh) & M28);
        }
        e.l[8] = (int32_t)w[7];  // bits 224..255 (below 2^28 for a canonical input)
        return e;
    }
    // canonicalises, then packs
    SR_HD static void store(storage *p, const elem &x) {
        const elem c = canonical(x);
        uint32_t u[10], o[8];
#pragma unroll
        for (int i = 0; i < 9; i++) u[i] = (uint32_t)c.l[i];
        u[9] = 0;
        o[0] = u[0] | (u[1] << 28);
        o[1] = (u[1] >> 4) | (u[2] << 24);
        o[2] = (u[2] >> 8) | (u[3] << 20);
        o[3] = (u[3] >> 12) | (u[4] << 16);
        o[4] = (u[4] >> 16) | (u[5] << 12);
        o[5] = (u[5] >> 20) | (u[6] << 8);
        o[6] = (u[6] >> 24) | (u[7] << 4);
        o[7] = u[8];
#pragma unroll
        for (int i = 0; i < 4; i++) p->q[i] = (uint64_t)o[2 * i] | ((uint64_t)o[2 * i + 1] << 32);
    }

    SR_HD static elem add(const elem &a, const elem &b) {
        elem r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            r.l[i] = (int32_t)((uint32_t)a.l[i] + (uint32_t)b.l[i]);  // wraps, never UB, on garbage input
            repcheck::stark_limb((long long)a.l[i] + (long long)b.l[i]);
        }
        return r;
    }
    SR_HD static elem sub(const elem &a, const elem &b) {
        elem r;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            r.l[i] = (int32_t)((uint32_t)a.l[i] - (uint32_t)b.l[i]);
            repcheck::stark_limb((long long)a.l[i] - (long long)b.l[i]);
        }
        return r;
    }
    SR_HD static elem relax(const elem &a) {
        elem x = a;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int32_t c = x.l[i] >> 28;  // arithmetic
            x.l[i] &= (int32_t)M28;
            x.l[i + 1] = (int32_t)((uint32_t)x.l[i + 1] + (uint32_t)c);
        }
        return x;
    }
    // relaxed in; value - floor(value / 2^251) p out
    SR_HD static elem fold(const elem &a) {
        elem x = a;
        const int32_t q = x.l[8] >> 27;
        x.l[8] &= (1 << 27) - 1;
        x.l[7] -= q;
        x.l[6] -= q * (1 << 24);
        x.l[0] -= q;
        return x;
    }
    SR_HD static elem weak_reduce(const elem &a) { return fold(relax(a)); }
    // The unique representative in [0, p), limbs normalised.  After relax + fold the value is floor-reduced modulo 2^251 -- in
    // (-2^206, 2^251 + 2^206): limbs 1..5 are normalised, limb 8 is below 2^27, and only limbs 0, 6, 7 have moved (by q, q 2^24,
    // q for the signed quotient q).  The carries of limbs 6 and 7 are settled in place.  A borrow or carry out of limb 0, a negative
    // value or one of 2^251 and more (together about 2^-25 of all inputs) take the slow path: a full sweep, then + p if
    // negative, - p if that does not go negative.
    SR_HD static elem canonical(const elem &a) {
        elem x = fold(relax(a));
        int32_t c = x.l[6] >> 28;
        x.l[6] &= (int32_t)M28;
        x.l[7] += c;
        c = x.l[7] >> 28;
        x.l[7] &= (int32_t)M28;
        x.l[8] += c;
        if ((uint32_t)x.l[0] > M28 || (uint32_t)x.l[8] >= (1u << 27)) {  // limb 0 left [0, 2^28) either way, or the value [0, 2^251)
            x = relax(x);
            if (x.l[8] < 0) {
                x.l[0] += 1;
                x.l[6] += 1 << 24;
                x.l[7] += 1;
                x.l[8] += 1 << 27;
                x = relax(x);
            } else if (x.l[8] >= (1 << 27)) {
                elem y = x;
                y.l[0] -= 1;
                y.l[6] -= 1 << 24;
                y.l[7] -= 1;
                y.l[8] -= 1 << 27;
                y = relax(y);
                if (y.l[8] >= 0) x = y;
            }
        }
        return x;
    }
    SR_HD static bool valid(const elem &a) {  // canonical representative?
        const elem c = canonical(a);
        bool same = true;
#pragma unroll
        for (int i = 0; i < 9; i++) same &= c.l[i] == a.l[i];
        return same;
    }

    // a * w * 2^-280 mod p (see the header comment for the operand ranges).  One 64-bit accumulator runs through all eighteen columns.
    // Device build: ONE asm statement per column (generated: tools/gen_stark_mul_cols.py -> stark_mul_cols.inc).  Left to itself the
    // compiler starts every column in a fresh accumulator and joins it to the running one with an extra 64-bit add (18 per
    // product); as one statement per multiply-add it padded nearly every one of them with a wait state it could not know to be
    // unnecessary (112 s_nop per 174 VALU) -- the accumulator dependency is interlocked by the hardware.
#if defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__ elem mul_tw(const elem &a, const elem &w) {
        const uint32_t c24 = 1u << 24, c27 = 1u << 27;
        int64_t acc = 0;
        uint64_t cy;
        uint32_t m[10];
        elem r;
        repcheck::stark_columns(a.l, w.l);
#include "stark_mul_cols.inc"
        return r;
    }
#else
    static void mac_s(int64_t &acc, int32_t x, int32_t y) { acc += (int64_t)x * (int64_t)y; }
    static void mac_u(int64_t &acc, uint32_t x, uint32_t y) { acc = (int64_t)((uint64_t)acc + (uint64_t)x * y); }
    static elem mul_tw(const elem &a, const elem &w) {
        const uint32_t c24 = 1u << 24, c27 = 1u << 27, c1 = 1u;
        int64_t acc = 0;
        uint32_t m[10];
        elem r;
#pragma unroll
        for (int k = 0; k < 18; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) mac_s(acc, a.l[i], w.l[j]);
            }
#pragma unroll
            for (int i = 0; i < 10; i++) {
                const int j = k - i;
                if (i < k || k >= 10) {
                    if (j == 6) mac_u(acc, m[i], c24);
                    if (j == 7) mac_u(acc, m[i], c1);
                    if (j == 8) mac_u(acc, m[i], c27);
                }
            }
            if (k < 10) {
                m[k] = (0u - (uint32_t)acc) & M28;
                mac_u(acc, m[k], c1);  // + m_k p_0: the low 28 bits are now zero
                acc >>= 28;
            } else {
                r.l[k - 10] = (int32_t)((uint32_t)acc & M28);
                acc >>= 28;
            }
        }
        r.l[8] = (int32_t)acc;
        return r;
    }
#endif
    // the slot product inside the fused kernel: both operands are data; they are weakly reduced first
    SR_HD static elem mul_data(const elem &a, const elem &b) { return mul_tw(weak_reduce(a), weak_reduce(b)); }

    // Sums of products of memory images (the linear-algebra kernels): pre(a, b) = a b 2^-280 may be summed lazily (a weak
    // reduction every four terms); post(x) = x 2^24 = mul_tw(x, 2^304 mod p) makes the sum the memory image of sum a_i b_i.
    SR_HD static elem mul_boundary_pre(const elem &a, const elem &b) { return mul_tw(a, b); }
    SR_HD static elem boundary_post(const elem &x) {
        elem c;
        c.l[0] = 0x1; c.l[1] = 0xe000000; c.l[2] = 0xfffffff; c.l[3] = 0xfffffff; c.l[4] = 0xfffffff;
        c.l[5] = 0xfffffff; c.l[6] = 0xffffff; c.l[7] = 0x1; c.l[8] = 0x5e00000;
        return mul_tw(x, c);
    }
    // table form of a small integer: x * 2^280 mod p = mul_tw(x, 2^560 mod p)
    SR_HD static elem tw_from_u64(uint64_t x) {
        elem e = zero(), r2;
        e.l[0] = (int32_t)(x & M28);
        e.l[1] = (int32_t)((x >> 28) & M28);
        e.l[2] = (int32_t)(x >> 56);
        r2.l[0] = 0xa943fef; r2.l[1] = 0x56; r2.l[2] = 0x37e0004; r2.l[3] = 0xfffffd7; r2.l[4] = 0x30fffff;
        r2.l[5] = 0x13; r2.l[6] = 0xe800000; r2.l[7] = 0x13d83e4; r2.l[8] = 0x5c;
        return canonical(mul_tw(e, r2));
    }
    SR_HD static elem tw_one() { return tw_from_u64(1); }

    // LDS accessors: limb-major (SoA), nine words per element
    SR_HD static elem lds_get(const uint32_t *lds, int idx, int n) {
        elem e;
#pragma unroll
        for (int i = 0; i < 9; i++) e.l[i] = (int32_t)lds[i * n + idx];
        return e;
    }
    SR_HD static void lds_put(uint32_t *lds, int idx, int n, const elem &v) {
#pragma unroll
        for (int i = 0; i < 9; i++) lds[i * n + idx] = (uint32_t)v.l[i];
    }
};

// hooks the field-agnostic kernels call; no-ops for the fields whose add / sub / mul_tw keep elements canonical
template <class F>
struct Lazy {
    static constexpr bool value = false;
    SR_HD static typename F::elem weak(const typename F::elem &x) { return x; }
    SR_HD static typename F::elem table(const typename F::elem &x) { return x; }
    SR_HD static typename F::elem mul_data(const typename F::elem &a, const typename F::elem &b) { return F::mul_tw(a, b); }
};
template <>
struct Lazy<StarkL> {
    static constexpr bool value = true;
    SR_HD static S9 weak(const S9 &x) { return StarkL::weak_reduce(x); }
    SR_HD static S9 table(const S9 &x) { return StarkL::canonical(x); }  // what goes into a twiddle table
    SR_HD static S9 mul_data(const S9 &a, const S9 &b) { return StarkL::mul_data(a, b); }
};

}  // namespace sr
