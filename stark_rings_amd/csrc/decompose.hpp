// Balanced (gadget) decomposition and recomposition, coefficient-wise over a batch of ring elements -- SURVEY 8f #2.
//
//   decompose_balanced_in_place   crates/ring/src/balanced_decomposition/mod.rs:62-117
//   signed representative         balanced_decomposition/fq_convertible.rs:21-35 (Fp64), stark_prime/decomposition.rs:41-53 (Fp256)
//   rounded_div                   crates/linear_algebra/src/ops.rs:64-80
//   ring elements coefficient-wise   cyclotomic_ring/coeff_form.rs:587-605
//   gadget_decompose / recompose     balanced_decomposition/mod.rs:163-175, 119-131, 177-189
//
// One lane = one coefficient.  The value leaves Montgomery form (x * R^-1 = mul_boundary(x, 1)), becomes sign + magnitude
// (the reference's i128 / BigInt), and k times: rem = |curr| mod b; |rem| <= b/2 keeps the digit and the truncated quotient,
// otherwise the digit is -(sign)(b - rem) and the quotient moves one away from zero.  Digits go back to Montgomery form
// (d * R = mul_boundary(d, R^2)) and are written digit-major: digit j of element e is ring element e * k + j of `out`, so the
// stores of a wave are contiguous.  Memory-bound: D w bytes in, k D w bytes out per element.
// Basis: any even b in [2, 2^64) (the reference takes a u128; its gadget bases are 2 .. 2^16; a basis of 2^64 or more is the one
// remaining narrowing, and only the 252-bit Stark prime could use it).  A power of two is a shift and a mask; any other even
// basis pays an integer division per digit (Stark: per 32-bit limb, a 64 / 32-bit division while b <= 2^32 and a restoring
// 96 / 64-bit division above).
#pragma once
#include <type_traits>

#include "fields.hpp"

namespace sr {
namespace dec {

// ---- per-field constants: the integer 1 and R^2 mod p (R = 2^kBoundaryBits) as elements -----------------------------------
template <class F>
struct Consts;
template <>
struct Consts<Goldilocks> {
    SR_HD static uint64_t one() { return 1; }
    SR_HD static uint64_t r2() { return 0xFFFFFFFE00000001ull; }  // 2^128 mod p
    SR_HD static uint64_t from_u64(uint64_t v) { return v; }       // v < p
};
template <>
struct Consts<Frog> {
    SR_HD static uint64_t one() { return 1; }
    SR_HD static uint64_t r2() { return Frog::R2; }  // 2^128 mod p
    SR_HD static uint64_t from_u64(uint64_t v) { return v; }  // v < p
};
template <>
struct Consts<BabyBear> {
    SR_HD static uint32_t one() { return 1; }
    SR_HD static uint32_t r2() { return 663890614u; }  // 2^128 mod p
    SR_HD static uint32_t from_u64(uint64_t v) { return (uint32_t)v; }
};
template <>
struct Consts<Stark> {
    SR_HD static U256 one() {
        U256 e = Stark::zero();
        e.l[0] = 1;
        return e;
    }
    SR_HD static U256 r2() { return Stark::r2(); }  // 2^512 mod p
    SR_HD static U256 from_u64(uint64_t v) {
        U256 e = Stark::zero();
        e.l[0] = (uint32_t)v;
        e.l[1] = (uint32_t)(v >> 32);
        return e;
    }
};

// ---- sign + magnitude of the signed representative ------------------------------------------------------------------------
// Fp64 fields: one u64.  [0, (p-1)/2] stays, ](p-1)/2, p[ -> p - x with the sign set.
template <class F>
struct Mag {
    uint64_t m;
    bool neg;
    SR_HD static Mag from_image(typename F::elem img) {
        const uint64_t x = (uint64_t)F::mul_boundary(img, Consts<F>::one());
        const uint64_t p = (uint64_t)F::P;
        Mag r;
        r.neg = x > (p - 1) / 2;
        r.m = r.neg ? p - x : x;
        return r;
    }
    SR_HD bool is_zero() const { return m == 0; }
    // m /= b, returns m mod b
    SR_HD uint64_t divrem(uint64_t b, int log2b) {
        uint64_t rem;
        if (log2b >= 0) {
            rem = m & (b - 1);
            m = log2b >= 64 ? 0 : m >> log2b;
        } else {
            const uint64_t q = m / b;
            rem = m - q * b;
            m = q;
        }
        return rem;
    }
    SR_HD void inc() { m++; }
};
// Stark: eight 32-bit limbs
template <>
struct Mag<Stark> {
    U256 m;
    bool neg;
    SR_HD static Mag from_image(const U256 &img) {
        U256 x = Stark::mul_boundary(img, Consts<Stark>::one());
        // x > (p-1)/2  <=>  2x > p - 1  <=>  2x >= p (p odd)
        U256 dbl, t;
        Stark::add_raw(dbl, x, x);  // < 2^253
        Mag r;
        r.neg = Stark::geq_p(dbl);
        if (r.neg) {
            Stark::sub_raw(t, Stark::modulus(), x);
            x = t;
        }
        r.m = x;
        return r;
    }
    SR_HD bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= m.l[i];
        return o == 0;
    }
    SR_HD uint64_t divrem(uint64_t b, int log2b) {
        uint64_t rem;
        if (log2b >= 0) {  // power of two, 2^1 .. 2^63: a mask and a shift over the limbs
            rem = ((uint64_t)m.l[0] | ((uint64_t)m.l[1] << 32)) & (b - 1);
            const int ws = log2b >> 5, bs = log2b & 31;
            uint32_t t[10];
#pragma unroll
            for (int i = 0; i < 8; i++) t[i] = m.l[i];
            t[8] = t[9] = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                // limb i of (m >> log2b): bits of limbs i + ws and i + ws + 1 (ws is 0 or 1)
                const uint32_t lo = ws ? t[i + 1] : t[i], hi = ws ? t[i + 2] : t[i + 1];
                m.l[i] = bs ? (uint32_t)((((uint64_t)hi << 32) | lo) >> bs) : lo;
            }
        } else if (b <= (1ull << 32)) {
            uint64_t r = 0;  // < b <= 2^32: (r << 32 | limb) fits 64 bits
#pragma unroll
            for (int i = 7; i >= 0; i--) {
                const uint64_t cur = (r << 32) | m.l[i];
                const uint64_t q = cur / b;
                m.l[i] = (uint32_t)q;
                r = cur - q * b;
            }
            rem = r;
        } else {
            // b above 2^32: (r 2^32 + limb) / b with r < b < 2^64 is a 96-by-64-bit division whose quotient fits 32 bits;
            // restoring division, one quotient bit per step (device code has no 128-bit divide)
            uint64_t r = 0;
#pragma unroll 1
            for (int i = 7; i >= 0; i--) {
                uint32_t q = 0;
                const uint32_t limb = m.l[i];
#pragma unroll 4
                for (int bit = 31; bit >= 0; bit--) {
                    const bool carry = (r >> 63) != 0;
                    r = (r << 1) | ((limb >> bit) & 1u);
                    if (carry || r >= b) {
                        r -= b;
                        q |= 1u << bit;
                    }
                }
                m.l[i] = q;
            }
            rem = r;
        }
        return rem;
    }
    SR_HD void inc() {
        uint32_t c = 1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint64_t s = (uint64_t)m.l[i] + c;
            m.l[i] = (uint32_t)s;
            c = (uint32_t)(s >> 32);
        }
    }
};

SR_HD int exact_log2(uint64_t b) { return (b & (b - 1)) == 0 ? 63 - __builtin_clzll(b) : -1; }

// the Montgomery image of the signed digit (-1)^neg * dig, dig < p
template <class F>
SR_HD typename F::elem digit_image(uint64_t dig, bool neg) {
    typename F::elem v = F::mul_boundary(Consts<F>::from_u64(dig), Consts<F>::r2());
    return (neg && dig) ? F::sub(F::zero(), v) : v;
}

// k balanced digits of one coefficient, written d coefficients apart (digit-major); returns true when more were needed
template <class F, class Store>
__device__ __forceinline__ bool decompose_one(typename F::elem img, uint64_t b, int log2b, size_t k, Store store) {
    Mag<F> cur = Mag<F>::from_image(img);
    for (size_t j = 0; j < k; j++) {
        const uint64_t rem = cur.divrem(b, log2b);
        uint64_t dig = rem;
        bool dneg = cur.neg;
        if (rem > b / 2) {
            dig = b - rem;
            dneg = !cur.neg;
            cur.inc();
        }
        store(j, digit_image<F>(dig, dneg));
    }
    return !cur.is_zero();
}

// in: batch ring elements of d coefficients; out: batch * k ring elements.  *overflow counts coefficients that needed more
// than k digits (the reference indexes out[k] and panics).  One-limb fields move two neighbouring coefficients per lane
// (16-byte accesses: the input pair and each of the k digit pairs) when d is even and the buffers are 16-byte aligned.
template <class F>
__global__ __launch_bounds__(256) void decompose_kernel(typename F::storage *out, const typename F::storage *in, size_t d,
                                                        size_t batch, uint64_t b, int log2b, size_t k,
                                                        unsigned long long *overflow) {
    const size_t n = batch * d;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(typename F::storage) == 8) {
        if ((d & 1) == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0)) {
            for (size_t t2 = gid; t2 < (n >> 1); t2 += stride) {
                const size_t t = t2 << 1, e = t / d, i = t - e * d;
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in + t));
                typename F::storage w[2] = {v.x, v.y};
                typename F::storage *o = out + e * k * d + i;
                // digit j of both coefficients leaves as one 16-byte store: buffer the first coefficient's digits in registers
                // would cost k registers; instead the two digit streams are produced in lock step
                Mag<F> c0 = Mag<F>::from_image(F::load(&w[0])), c1 = Mag<F>::from_image(F::load(&w[1]));
                for (size_t j = 0; j < k; j++) {
                    typename F::storage dg[2];
                    Mag<F> *cs[2] = {&c0, &c1};
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        Mag<F> &cur = *cs[q];
                        const uint64_t rem = cur.divrem(b, log2b);
                        uint64_t dig = rem;
                        bool dneg = cur.neg;
                        if (rem > b / 2) {
                            dig = b - rem;
                            dneg = !cur.neg;
                            cur.inc();
                        }
                        F::store(&dg[q], digit_image<F>(dig, dneg));
                    }
                    u64x2 ov;
                    ov.x = dg[0];
                    ov.y = dg[1];
                    __builtin_nontemporal_store(ov, reinterpret_cast<u64x2 *>(o + j * d));
                }
                const unsigned over = (c0.is_zero() ? 0u : 1u) + (c1.is_zero() ? 0u : 1u);
                if (over) atomicAdd(overflow, (unsigned long long)over);
            }
            return;
        }
    }
    for (size_t t = gid; t < n; t += stride) {
        const size_t e = t / d, i = t - e * d;
        typename F::storage *o = out + e * k * d + i;
        if (decompose_one<F>(F::load(in + t), b, log2b, k, [&](size_t j, const typename F::elem &v) { F::store(o + j * d, v); }))
            atomicAdd(overflow, 1ull);
    }
}

// out[e][i] = sum_j b^j in[e * k + j][i]: Horner from the top digit (mod.rs:119-131, 177-189); two coefficients per lane like
// decompose_kernel
template <class F>
__global__ __launch_bounds__(256) void recompose_kernel(typename F::storage *out, const typename F::storage *in, size_t d,
                                                        size_t batch_out, uint64_t b, size_t k) {
    const size_t n = batch_out * d;
    // Montgomery image of R::from(b): b mod p first for the one-limb fields (b < 2^64 may exceed p); b < p for Stark
    uint64_t bred = b;
    if constexpr (!std::is_same<F, Stark>::value) bred = b % (uint64_t)F::P;
    const typename F::elem bimg = F::mul_boundary(Consts<F>::from_u64(bred), Consts<F>::r2());
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(typename F::storage) == 8) {
        if ((d & 1) == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0)) {
            for (size_t t2 = gid; t2 < (n >> 1); t2 += stride) {
                const size_t t = t2 << 1, e = t / d, i = t - e * d;
                const typename F::storage *src = in + e * k * d + i;
                typename F::elem a0 = F::zero(), a1 = F::zero();
                typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
                for (size_t j = k; j-- > 0;) {
                    const u64x2 v = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(src + j * d));
                    typename F::storage w[2] = {v.x, v.y};
                    a0 = F::add(F::mul_boundary(a0, bimg), F::load(&w[0]));
                    a1 = F::add(F::mul_boundary(a1, bimg), F::load(&w[1]));
                }
                typename F::storage o[2];
                F::store(&o[0], a0);
                F::store(&o[1], a1);
                u64x2 ov;
                ov.x = o[0];
                ov.y = o[1];
                __builtin_nontemporal_store(ov, reinterpret_cast<u64x2 *>(out + t));
            }
            return;
        }
    }
    for (size_t t = gid; t < n; t += stride) {
        const size_t e = t / d, i = t - e * d;
        const typename F::storage *src = in + e * k * d + i;
        typename F::elem acc = F::zero();
        for (size_t j = k; j-- > 0;) acc = F::add(F::mul_boundary(acc, bimg), F::load(src + j * d));
        F::store(out + t, acc);
    }
}

// ---- bases of 2^64 and more (the reference takes a u128) -- Stark only: the one-limb fields have |x| <= (p - 1) / 2 < 2^63 < b / 2,
// so their decomposition is digit 0 = x, every other digit 0, and capi.hip does that with a copy ----------------------------------
struct U128 {
    uint64_t lo, hi;
};
SR_HD bool u128_gt(const U128 &a, const U128 &b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
SR_HD bool u128_geq(const U128 &a, const U128 &b) { return a.hi > b.hi || (a.hi == b.hi && a.lo >= b.lo); }
SR_HD U128 u128_sub(const U128 &a, const U128 &b) {
    U128 r;
    r.lo = a.lo - b.lo;
    r.hi = a.hi - b.hi - (a.lo < b.lo ? 1 : 0);
    return r;
}
SR_HD U256 u256_from_u128(const U128 &v) {
    U256 e = Stark::zero();
    e.l[0] = (uint32_t)v.lo;
    e.l[1] = (uint32_t)(v.lo >> 32);
    e.l[2] = (uint32_t)v.hi;
    e.l[3] = (uint32_t)(v.hi >> 32);
    return e;
}
// m /= b, returns m mod b, for 2^64 <= b < 2^128: restoring division, one quotient bit per step
SR_HD U128 divrem_wide(U256 &m, const U128 &b) {
    U128 r{0, 0};
#pragma unroll 1
    for (int i = 7; i >= 0; i--) {
        uint32_t q = 0;
        const uint32_t limb = m.l[i];
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            const bool carry = (r.hi >> 63) != 0;
            r.hi = (r.hi << 1) | (r.lo >> 63);
            r.lo = (r.lo << 1) | ((limb >> bit) & 1u);
            if (carry || u128_geq(r, b)) {
                r = u128_sub(r, b);
                q |= 1u << bit;
            }
        }
        m.l[i] = q;
    }
    return r;
}
__global__ __launch_bounds__(256) void decompose_wide_kernel(U256Storage *out, const U256Storage *in, size_t d, size_t batch, U128 b,
                                                             size_t k, unsigned long long *overflow) {
    const size_t n = batch * d;
    const U128 half{(b.lo >> 1) | (b.hi << 63), b.hi >> 1};
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / d, i = t - e * d;
        Mag<Stark> cur = Mag<Stark>::from_image(Stark::load(in + t));
        U256Storage *o = out + e * k * d + i;
        for (size_t j = 0; j < k; j++) {
            const U128 rem = divrem_wide(cur.m, b);
            U128 dig = rem;
            bool dneg = cur.neg;
            if (u128_gt(rem, half)) {
                dig = u128_sub(b, rem);
                dneg = !cur.neg;
                cur.inc();
            }
            U256 v = Stark::mul_boundary(u256_from_u128(dig), Consts<Stark>::r2());
            if (dneg && (dig.lo | dig.hi)) v = Stark::sub(Stark::zero(), v);
            Stark::store(o + j * d, v);
        }
        if (!cur.is_zero()) atomicAdd(overflow, 1ull);
    }
}
__global__ __launch_bounds__(256) void recompose_wide_kernel(U256Storage *out, const U256Storage *in, size_t d, size_t batch_out,
                                                             U128 b, size_t k) {
    const size_t n = batch_out * d;
    const U256 bimg = Stark::mul_boundary(u256_from_u128(b), Consts<Stark>::r2());  // b < 2^128 < p
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / d, i = t - e * d;
        const U256Storage *src = in + e * k * d + i;
        U256 acc = Stark::zero();
        for (size_t j = k; j-- > 0;) acc = Stark::add(Stark::mul_boundary(acc, bimg), Stark::load(src + j * d));
        Stark::store(out + t, acc);
    }
}

}  // namespace dec
}  // namespace sr
