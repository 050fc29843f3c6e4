// Tuned Goldilocks path for Fp[X]/(X^D+1), 2^8 <= D <= 2^22 (BASELINE configs 1, 2 and 4).
//
// Same function as the generic kernels -- the reference's stark_prime-style negacyclic NTT
// (crates/ring/src/cyclotomic_ring/models/stark_prime/ntt.rs:121-346 generalised),
// outputs bit-identical -- but decomposed for the gfx950 integer pipe.  In Goldilocks 2 has order 192 (2^96 = -1) and
// omega_64 = 7^((p-1)/64) = 8^13 = 2^39: every root of unity of order <= 64 is a power of two, and multiplying by one
// is a shift plus one fold instead of a general product.  The decomposition keeps every butterfly twiddle a
// compile-time shift and leaves ONE general (table) product per coefficient between two register passes.
//
//   column stages + twist: the first c merged negacyclic radix-2 stages split X^D + 1 into 2^c factors
//       X^N2 - gamma_b^N2 (N2 = D >> c, gamma_b = psi^(2 brv_c(b) + 1)); multiplying block b, position i by gamma_b^i
//       turns each block into a plain CYCLIC DFT_N2 problem.
//         2^16 <= D <= 2^20: cols256_kernel, c = 8 in one launch, shift-only (see the comment at the kernel);
//         2^13 .. 2^15 and 2^21, 2^22: strided_kernel<M> / strided256_kernel with table twiddles.
//   rows: cyclic DFT_N2 out of shift-only radix-16 register passes (omega_16 = 2^156 = -2^60), 16 coefficients per lane,
//       exchanges through a padded 34 KiB LDS tile: rows256_kernel (N2 = 256: 16 x 16), rows_kernel (N2 = 4096:
//       16 x 16 x 16; 512..2048: leading stages skipped, several blocks per tile).
//   D <= 4096: no column launch at all -- a tile holds 4096 / D whole ring elements and rows_kernel<.., TW = true>
//       applies the twist itself (a compile-time shift per register slot and a column factor merged into its table).
//   The fused ring product keeps fwd(a) in registers while fwd(b) runs, multiplies slot-wise
//   (ntt_form.rs:177-189), and runs the inverse from registers: a, b read once, c written once.
//   Inverse = mirror image; D^-1 (and, for the fused product, R^-1 = 2^-64, see fields.hpp) is folded
//   into the inverse twist table (D <= 4096: into the inverse W1 table).
//
// tools/model_fast_goldilocks.py is the index-level model of this file, checked against the oracle.
#pragma once
#include <utility>

#include "fields.hpp"

namespace sr {
namespace gl {

using G = Goldilocks;
typedef uint64_t u64;
typedef uint32_t u32;

constexpr int kW16Exp = 156;  // omega_16 = 2^156 (= (8^13)^4)

// Operand reads and result writes happen once per launch: non-temporal.  Intermediates (column-pass output, rows output) are written
// and read WITHOUT the hint, so that they allocate in the 256 MB Infinity Cache: the ring product runs in chunks small enough for a
// chunk's intermediates to stay there (gl_fast_ring_mul_lanes).  Tables keep the default policy and stay in L2.
__device__ __forceinline__ u64 ld_stream(const u64 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(u64 *p, u64 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ u64 ld_scratch(const u64 *p) { return *p; }
__device__ __forceinline__ void st_scratch(u64 *p, u64 v) { *p = v; }
// a word that leaves the library (a result of sr_ntt_fwd / sr_ntt_inv / sr_ring_mul): canonical by contract
__device__ __forceinline__ void st_result(u64 *p, u64 v) {
    repcheck::leaves_library(v);
    st_stream(p, v);
}
// the NTT slots a plain forward transform leaves in place of its operand (default cache policy, as ever)
__device__ __forceinline__ void st_slots(u64 *p, u64 v) {
    repcheck::leaves_library(v);
    *p = v;
}
constexpr int kTile = 4096;
constexpr int kLdsElems = kTile + kTile / 16;  // padded: pos + (pos >> 4)

// x * 2^E mod p for a compile-time 0 < E < 96.  ANY 64-bit representative in (nothing below assumes x < p: the three branches
// only need l2 + hl eps < 2^64 + p, which holds for every x -- tools/ubench/field_check.hip runs all 95 exponents on
// non-canonical operands), canonical out: 5 / 7 / 7 VALU for E < 32 / < 64 / < 96.
template <int E>
SR_HD u64 mul_pow2(u64 x) {
    static_assert(E > 0 && E < 96, "shift out of range");
    constexpr int q = E / 32, r = E % 32;
    const u64 xs = x << r;                                  // low 64 bits of x * 2^r
    const u32 y2 = r ? (u32)(x >> (64 - r)) : 0u;           // bits 64.. of x * 2^r  (< 2^31)
    u64 res;
    if constexpr (q == 0) {
        res = G::mad_eps_fix(xs, y2);                       // xs + y2 * eps: nothing to subtract (reduce128 with a zero top word)
    } else if constexpr (q == 1) {
        res = G::reduce128(xs << 32, (xs >> 32) | ((u64)y2 << 32));
    } else {                                                // (y0 + y1 2^32 + y2 2^64) * 2^64 = y0 * eps - (y1 + y2 2^32)
        const u32 y0 = (u32)xs;
        const u64 B = ((u64)y0 << 32) - y0;                 // y0 * (2^32 - 1) < p
        const u64 C = (xs >> 32) | ((u64)y2 << 32);         // < 2^63 < p
        res = G::sub(B, C);
    }
    repcheck::canonical_out(res);
    return res;
}

// decimation-in-frequency butterfly: (a, b) -> (a + b, (a - b) * 2^E), E in [0, 192); canonical in, canonical out.
// FUSED: sum and difference from one statement (Goldilocks::addsub); otherwise add and sub as separate routines (the form the
// whole-ring-element tiles keep: they are at 160 VGPRs already).
template <int E, bool FUSED = true>
SR_HD void bf_dif(u64 &a, u64 &b) {
    if constexpr (FUSED && E < 96) {
        u64 s, d;
        G::addsub(a, b, s, d);
        a = s;
        if constexpr (E == 0) b = d;
        else b = mul_pow2<E>(d);
        return;
    }
    const u64 s = G::add(a, b);
    u64 d;
    if constexpr (E == 0) {
        d = G::sub(a, b);
    } else if constexpr (E < 96) {
        d = mul_pow2<E>(G::sub(a, b));
    } else if constexpr (E == 96) {
        d = G::sub(b, a);
    } else {
        d = mul_pow2<E - 96>(G::sub(b, a));
    }
    a = s;
    b = d;
}
// decimation-in-time butterfly: (u, v) -> (u + v 2^E, u - v 2^E).  LAZY: a butterfly with a shift (E % 96 != 0) adds and subtracts
// the CANONICAL v 2^E to ANY 64-bit representative u with six VALU instead of seven (Goldilocks::addsub_lazy) and leaves arbitrary
// representatives; butterflies with twiddle 1 keep the canonical form and need canonical inputs -- in the DFT_16 networks below
// their inputs always come out of twiddle-1 butterflies, all the way back to the network's inputs (tools/model_fast_goldilocks.py
// asserts it; the SR_GL_CHECK_REPS build counts violations on the device).
template <int E, bool FUSED = true, bool LAZY = false>
SR_HD void bf_dit(u64 &u, u64 &v) {
    if constexpr (LAZY && E % 96 != 0) {
        const u64 t = mul_pow2<E % 96>(v);
        u64 s, d;
        G::addsub_lazy(u, t, s, d);
        u = E >= 96 ? d : s;
        v = E >= 96 ? s : d;
        return;
    }
    if constexpr (FUSED && E < 96) {
        u64 t = v;
        if constexpr (E != 0) t = mul_pow2<E>(v);
        u64 s, d;
        G::addsub(u, t, s, d);
        u = s;
        v = d;
        return;
    } else if constexpr (FUSED) {
        u64 t = v;
        if constexpr (E != 96) t = mul_pow2<E - 96>(v);
        u64 s, d;
        G::addsub(u, t, s, d);   // u - t is the "sum" leg, u + t the "difference" leg
        u = d;
        v = s;
        return;
    }
    if constexpr (E == 0) {
        const u64 s = G::add(u, v), d = G::sub(u, v);
        u = s;
        v = d;
    } else if constexpr (E < 96) {
        const u64 t = mul_pow2<E>(v);
        const u64 s = G::add(u, t), d = G::sub(u, t);
        u = s;
        v = d;
    } else if constexpr (E == 96) {
        const u64 s = G::sub(u, v), d = G::add(u, v);
        u = s;
        v = d;
    } else {
        const u64 t = mul_pow2<E - 96>(v);
        const u64 s = G::sub(u, t), d = G::add(u, t);
        u = s;
        v = d;
    }
}

template <int HALF, int STEP, int BASE, bool FUSED, int... Js>
SR_HD void dif_group(u64 *x, std::integer_sequence<int, Js...>) {
    (bf_dif<(STEP * Js) % 192, FUSED>(x[BASE + Js], x[BASE + Js + HALF]), ...);
}
template <int HALF, int STEP, int BASE, bool FUSED, bool LAZY = false, int... Js>
SR_HD void dit_group(u64 *x, std::integer_sequence<int, Js...>) {
    (bf_dit<(192 - (STEP * Js) % 192) % 192, FUSED, LAZY>(x[BASE + Js], x[BASE + Js + HALF]), ...);
}
template <int E>
SR_HD u64 shift96(u64 x) {  // x 2^E, 0 <= E < 96
    if constexpr (E == 0) return x;
    else return mul_pow2<E>(x);
}
// the stages of the cyclic DFT_16 as a decimation-in-TIME network with natural order in and bit-reversed order out (the same
// function as the DIF network dft16_fwd; what the forward kernels use with lazy butterflies): stage U pairs (lo, lo + (8 >> U));
// every butterfly of block blk = lo / (16 >> U) carries omega_16^((8 >> U) brv_U(blk)).  tools/model_fast_goldilocks.py: dft16_fwd_dit
constexpr int brv_n(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
template <int U>
struct StageNat {
    template <int I>
    struct Bf {
        static constexpr int half = 8 >> U;
        static constexpr int lo = (I / half) * 2 * half + I % half, hi = lo + half;
        static constexpr int E = (kW16Exp * half * brv_n(I / half, U)) % 192;
    };
};
// The butterflies of a stage run PHASE BY PHASE -- every a + eps, then every pair of carry chains, then every pair of masked
// corrections, then every shift product -- so that no statement's result is read by the statement right behind it (the compiler's
// post-asm wait state has nowhere to go) and independent work sits between producer and consumer everywhere: same values, same
// seven VALU per butterfly as one Goldilocks::addsub each, a third fewer non-VALU issue slots.
// BF<I> describes butterfly I of the stage: register slots lo, hi and its twiddle exponent E in [0, 192).
template <template <int> class BF, int... Is>
SR_HD void dif_phased(u64 *x, std::integer_sequence<int, Is...>) {  // (a, b) -> (a + b, (a - b) 2^E); canonical in and out
    constexpr int n = 8;  // indexed by the butterfly number (a group may be any subset of a stage's eight)
    u64 t[n], c1[n], c2[n];
    u32 s0[n], s1[n], d0[n], d1[n];
    (repcheck::canonical_in(x[BF<Is>::lo], x[BF<Is>::hi]), ...);
    ((t[Is] = G::plus_eps(x[BF<Is>::lo])), ...);
    (G::addsub_chains<(BF<Is>::E >= 96)>(t[Is], x[BF<Is>::lo], x[BF<Is>::hi], s0[Is], s1[Is], d0[Is], d1[Is], c1[Is], c2[Is]), ...);
    ((x[BF<Is>::lo] = (u64)s0[Is] | ((u64)s1[Is] << 32), x[BF<Is>::hi] = (u64)d0[Is] | ((u64)d1[Is] << 32)), ...);
    (G::addsub_fix(x[BF<Is>::lo], x[BF<Is>::hi], c1[Is], c2[Is]), ...);
    ((x[BF<Is>::hi] = shift96<BF<Is>::E % 96>(x[BF<Is>::hi])), ...);
}
template <bool LZ>
SR_HD u64 dit_pre_add(u64 a, u64 t) {   // the a + eps of the canonical sum; a lazy butterfly has none
    if constexpr (LZ) {
        repcheck::canonical_in(t, t);     // any a, canonical t (the shift product)
        return a;
    } else {
        repcheck::canonical_in(a, t);
        return G::plus_eps(a);
    }
}
template <bool LZ>
SR_HD void dit_fix(u64 &s, u64 &d, u64 c1, u64 c2) {
    if constexpr (LZ) G::addsub_lazy_fix(s, d, c1, c2);
    else G::addsub_fix(s, d, c1, c2);
}
template <template <int> class BF, bool LAZY = false, int... Is>
SR_HD void dit_phased(u64 *x, std::integer_sequence<int, Is...>) {  // (u, v) -> (u + v 2^E, u - v 2^E)
    constexpr int n = 8;
    u64 t[n], c1[n], c2[n], sv[n], dv[n];
    u32 s0[n], s1[n], d0[n], d1[n];
    // v 2^E first (E >= 96: v 2^(E - 96), and the legs swap: u - t is the sum leg)
    ((x[BF<Is>::hi] = shift96<BF<Is>::E % 96>(x[BF<Is>::hi])), ...);
    ((t[Is] = dit_pre_add<(LAZY && BF<Is>::E % 96 != 0)>(x[BF<Is>::lo], x[BF<Is>::hi])), ...);
    (G::addsub_chains<false>(t[Is], x[BF<Is>::lo], x[BF<Is>::hi], s0[Is], s1[Is], d0[Is], d1[Is], c1[Is], c2[Is]), ...);
    ((sv[Is] = (u64)s0[Is] | ((u64)s1[Is] << 32), dv[Is] = (u64)d0[Is] | ((u64)d1[Is] << 32)), ...);
    (dit_fix<(LAZY && BF<Is>::E % 96 != 0)>(sv[Is], dv[Is], c1[Is], c2[Is]), ...);
    ((x[BF<Is>::lo] = BF<Is>::E >= 96 ? dv[Is] : sv[Is], x[BF<Is>::hi] = BF<Is>::E >= 96 ? sv[Is] : dv[Is]), ...);
}
// the stages of the cyclic DFT_16 networks: butterfly I = block I / HALF, position I % HALF
template <int HALF, int STEP, bool DIT>
struct StageOf {
    template <int I>
    struct Bf {
        static constexpr int lo = (I / HALF) * 2 * HALF + I % HALF, hi = lo + HALF;
        static constexpr int E = DIT ? (192 - (STEP * (I % HALF)) % 192) % 192 : (STEP * (I % HALF)) % 192;
    };
};
// P butterflies go through the phases together: 8 = a whole stage of a DFT_16 (kPhased, what every kernel but the fused 4096-point
// products uses); P = 0: butterfly by butterfly (Goldilocks::addsub); P < 0: add and sub as separate routines.  The fused
// 4096-point product kernels keep a whole transformed tile in registers beside the one in flight and spill under the phased form
// (28-31 VGPRs): they take P = 0, and the whole-ring-element tiles (D <= 4096, already at 160 VGPRs) P = -1.
constexpr int kPhased = 8;
template <int OFF, int... Is>
constexpr std::integer_sequence<int, (OFF + Is)...> seq_from(std::integer_sequence<int, Is...>) { return {}; }
template <template <int> class BF, bool DIT, int N, int P, bool LAZY = false>
SR_HD void stage_in_groups(u64 *x) {
    constexpr int G = P < N ? P : N;
    if constexpr (DIT) {
        dit_phased<BF, LAZY>(x, std::make_integer_sequence<int, G>{});
        if constexpr (G < N) dit_phased<BF, LAZY>(x, seq_from<G>(std::make_integer_sequence<int, N - G>{}));
    } else {
        dif_phased<BF>(x, std::make_integer_sequence<int, G>{});
        if constexpr (G < N) dif_phased<BF>(x, seq_from<G>(std::make_integer_sequence<int, N - G>{}));
    }
}
template <int HALF, int STEP, int P = kPhased, int... Bs>
SR_HD void dif_stage(u64 *x, std::integer_sequence<int, Bs...>) {
    if constexpr (P <= 0) (dif_group<HALF, STEP, Bs * 2 * HALF, P == 0>(x, std::make_integer_sequence<int, HALF>{}), ...);
    else stage_in_groups<StageOf<HALF, STEP, false>::template Bf, false, HALF * (int)sizeof...(Bs), P>(x);
}
template <int HALF, int STEP, int P = kPhased, bool LAZY = false, int... Bs>
SR_HD void dit_stage(u64 *x, std::integer_sequence<int, Bs...>) {
    if constexpr (P <= 0) (dit_group<HALF, STEP, Bs * 2 * HALF, P == 0, LAZY>(x, std::make_integer_sequence<int, HALF>{}), ...);
    else stage_in_groups<StageOf<HALF, STEP, true>::template Bf, true, HALF * (int)sizeof...(Bs), P, LAZY>(x);
}
template <int U, int P, bool LAZY, int... Is>
SR_HD void dit_nat_bfs(u64 *x, std::integer_sequence<int, Is...>) {
    (bf_dit<StageNat<U>::template Bf<Is>::E, P == 0, LAZY>(x[StageNat<U>::template Bf<Is>::lo], x[StageNat<U>::template Bf<Is>::hi]), ...);
}
template <int U, int P = kPhased, bool LAZY = false>
SR_HD void dit_nat_stage(u64 *x) {
    if constexpr (P <= 0) dit_nat_bfs<U, P, LAZY>(x, std::make_integer_sequence<int, 8>{});
    else stage_in_groups<StageNat<U>::template Bf, true, 8, P, LAZY>(x);
}
// 16-point cyclic DFT with omega_16 = 2^156: natural order in, bit-reversed order out (unnormalised)
template <int P = kPhased>
SR_HD void dft16_fwd(u64 *x) {
    dif_stage<8, kW16Exp, P>(x, std::make_integer_sequence<int, 1>{});
    dif_stage<4, (kW16Exp * 2) % 192, P>(x, std::make_integer_sequence<int, 2>{});
    dif_stage<2, (kW16Exp * 4) % 192, P>(x, std::make_integer_sequence<int, 4>{});
    dif_stage<1, (kW16Exp * 8) % 192, P>(x, std::make_integer_sequence<int, 8>{});
}
// inverse network: bit-reversed order in, natural order out, result = 16 * original.  LAZY: canonical in, arbitrary 64-bit
// representatives out (slot 0, which only twiddle-1 butterflies touch, stays canonical)
template <int P = kPhased, bool LAZY = false>
SR_HD void dft16_inv(u64 *x) {
    dit_stage<1, (kW16Exp * 8) % 192, P, LAZY>(x, std::make_integer_sequence<int, 8>{});
    dit_stage<2, (kW16Exp * 4) % 192, P, LAZY>(x, std::make_integer_sequence<int, 4>{});
    dit_stage<4, (kW16Exp * 2) % 192, P, LAZY>(x, std::make_integer_sequence<int, 2>{});
    dit_stage<8, kW16Exp, P, LAZY>(x, std::make_integer_sequence<int, 1>{});
}
// dft16_fwd as a decimation-in-time network (StageNat): same order in, same order out, same values (mod p)
template <int P = kPhased, bool LAZY = false>
SR_HD void dft16_fwd_dit(u64 *x) {
    dit_nat_stage<0, P, LAZY>(x);
    dit_nat_stage<1, P, LAZY>(x);
    dit_nat_stage<2, P, LAZY>(x);
    dit_nat_stage<3, P, LAZY>(x);
}
// what the D = 2^16 .. 2^20 kernels call: CANON = the results leave the library as they are (plain forward transform)
template <bool CANON = false, int P = kPhased>
SR_HD void dft16_fwd_hot(u64 *x) {
    dft16_fwd_dit<P, !CANON>(x);
}
template <int P = kPhased>
SR_HD void dft16_inv_hot(u64 *x) { dft16_inv<P, true>(x); }

// Q leading stages skipped: 2^Q independent cyclic DFTs of size 16 >> Q on consecutive register groups (used when
// D < 4096 and a tile holds 2^Q ring elements: the stride-256 pass must not mix them); same twiddles as the tail
// of the full network because omega_(16 >> Q) = omega_16^(2^Q)
template <int Q, int P = kPhased>
SR_HD void dft16_fwd_q(u64 *x) {
    if constexpr (Q <= 0) dif_stage<8, kW16Exp, P>(x, std::make_integer_sequence<int, 1>{});
    if constexpr (Q <= 1) dif_stage<4, (kW16Exp * 2) % 192, P>(x, std::make_integer_sequence<int, 2>{});
    if constexpr (Q <= 2) dif_stage<2, (kW16Exp * 4) % 192, P>(x, std::make_integer_sequence<int, 4>{});
    if constexpr (Q <= 3) dif_stage<1, (kW16Exp * 8) % 192, P>(x, std::make_integer_sequence<int, 8>{});
}
template <int Q, int P = kPhased>
SR_HD void dft16_inv_q(u64 *x) {
    if constexpr (Q <= 3) dit_stage<1, (kW16Exp * 8) % 192, P>(x, std::make_integer_sequence<int, 8>{});
    if constexpr (Q <= 2) dit_stage<2, (kW16Exp * 4) % 192, P>(x, std::make_integer_sequence<int, 4>{});
    if constexpr (Q <= 1) dit_stage<4, (kW16Exp * 2) % 192, P>(x, std::make_integer_sequence<int, 2>{});
    if constexpr (Q <= 0) dit_stage<8, kW16Exp, P>(x, std::make_integer_sequence<int, 1>{});
}

// x * 2^E for any compile-time E in [0, 192): 2^96 = -1
template <int E>
SR_HD u64 mul_pow2_signed(u64 x) {
    if constexpr (E == 0) return x;
    else if constexpr (E < 96) return mul_pow2<E>(x);
    else if constexpr (E == 96) return G::neg(x);
    else return G::neg(mul_pow2<E - 96>(x));
}
// Row part of the negacyclic twist when D = 4096 >> Q: register slot J of a lane is row J mod (16 >> Q) of its ring element and
// psi^256 = 2^(39 * 2^(Q+1)) (because psi^(2D/64) = omega_64 = 2^39), so (psi^256)^row is a compile-time shift.
constexpr int twist_exp(int q) { return (39 << (q + 1)) % 192; }
template <int Q, bool INVERSE, int... Js>
SR_HD void twist_rows(u64 *x, std::integer_sequence<int, Js...>) {
    ((x[Js] = mul_pow2_signed<INVERSE ? (192 - (twist_exp(Q) * (Js & ((16 >> Q) - 1))) % 192) % 192
                                      : (twist_exp(Q) * (Js & ((16 >> Q) - 1))) % 192>(x[Js])),
     ...);
}

struct Tables {
    const u64 *tw, *itw;                     // merged-stage twiddles (shared with the generic path)
    const u64 *twist_f;                      // [b * 4096 + i] = gamma_b^i
    const u64 *twist_i_plain, *twist_i_mul;  // gamma_b^-i * D^-1   (and * 2^-64 for the fused product)
    const u64 *w1f, *w1i, *w1i_mul;          // [rho * 256 + i0] = omega_N^(+- i0 m0(rho)), N = min(D, 4096); for D <= 4096 they also carry
                                             // the column part psi^(+-i0) of the twist and (inverse) D^-1 (w1i_mul: times 2^-64)
    const u64 *w2f, *w2i;                    // [sigma * 16 + i0] = omega_256^(+- i0 brv4(sigma))
    const u64 *wcf, *wci;                    // [h * 16 + rg] = theta^(+- (2 brv4(h) + 1) rg), theta = psi^(D/256)  (cols256_kernel)
};

// ------------------------------------------------------------------------------------------------
// strided pass: merged stages [s_lo, s_lo + M) of the negacyclic transform, 2^M legs per lane.
// grid.x = npoly * 2^s_lo * (S / 256), S = D >> (s_lo + M) the leg stride.  FWD applies the twist after
// its butterflies, INV before (TWIST only on the pass adjacent to the rows kernel, where S = 4096).
// ------------------------------------------------------------------------------------------------
template <int M, int DIR, bool TWIST>
__global__ __launch_bounds__(256) void strided_kernel(u64 *data, const u64 *src, int k, int s_lo, const u64 *tw, const u64 *twist) {
    constexpr int R = 1 << M;
    const int ls = k - s_lo - M;  // log2 S
    const size_t S = (size_t)1 << ls;
    const unsigned chunks = (unsigned)(S >> 8);
    const unsigned ci = blockIdx.x % chunks;
    const unsigned rest = blockIdx.x / chunks;
    const unsigned h = rest & ((1u << s_lo) - 1u);
    const size_t poly = rest >> s_lo;
    const unsigned i = ci * 256u + threadIdx.x;
    const size_t off = (poly << k) + ((size_t)h << (k - s_lo)) + i;
    u64 *base = data + off;
    const u64 *sbase = src + off;  // src == data: in place; otherwise the pass reads src and leaves it intact

    u64 x[R];
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = sbase[(size_t)j << ls];

    if (DIR == 0) {
#pragma unroll
        for (int t = 0; t < M; t++) {
            const int half = 1 << (M - 1 - t);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (j & half) continue;
                const u64 w = tw[(1u << (s_lo + t)) + (h << t) + (unsigned)(j >> (M - t))];
                const u64 u = x[j], v = G::mul(x[j + half], w);
                x[j] = G::add(u, v);
                x[j + half] = G::sub(u, v);
            }
        }
        if (TWIST) {
#pragma unroll
            for (int j = 0; j < R; j++) x[j] = G::mul(x[j], twist[((size_t)((h << M) + j) << 12) + i]);
        }
    } else {
        if (TWIST) {
#pragma unroll
            for (int j = 0; j < R; j++) x[j] = G::mul(x[j], twist[((size_t)((h << M) + j) << 12) + i]);
        }
#pragma unroll
        for (int t = M - 1; t >= 0; t--) {
            const int half = 1 << (M - 1 - t);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (j & half) continue;
                const u64 w = tw[(1u << (s_lo + t)) + (h << t) + (unsigned)(j >> (M - t))];
                const u64 u = x[j], v = x[j + half];
                x[j] = G::add(u, v);
                x[j + half] = G::mul(G::sub(u, v), w);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) base[(size_t)j << ls] = x[j];
}

// ------------------------------------------------------------------------------------------------
// strided pass of EIGHT merged stages [s_lo, s_lo + 8) in one trip over HBM (used when D >= 2^20, where two
// 4-stage passes would stream every operand twice): a workgroup owns 256 legs (stride S) x 16 consecutive
// columns = 4096 coefficients.  Pass A: lane (col, rg) holds legs rg + 16 jj and runs stages 0..3 (leg distance
// 128..16, twiddles wave-uniform); one exchange through the padded LDS tile (write pattern jj*256 + t, read
// pattern rg'*256 + jj*16 + col: the same two conflict-free patterns as the rows kernel); pass B: legs 16 rg' + jj,
// stages 4..7 (distance 8..1, 15 per-lane twiddles).  128-byte global segments (16 columns x 8 B).
// grid.x = npoly * 2^s_lo * (S / 16).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pad(int pos);  // defined with the rows kernel below

template <int DIR, bool TWIST>
__global__ __launch_bounds__(256, 4) void strided256_kernel(u64 *data, const u64 *src, int k, int s_lo, const u64 *tw, const u64 *twist) {
    __shared__ u64 lds[kLdsElems];
    const int t = threadIdx.x;
    const int ls = k - s_lo - 8;  // log2 S
    const unsigned chunks = 1u << (ls - 4);
    const unsigned ci = blockIdx.x & (chunks - 1u);
    const unsigned rest = blockIdx.x >> (ls - 4);
    const unsigned h = rest & ((1u << s_lo) - 1u);
    const size_t poly = rest >> s_lo;
    const int col = t & 15, rg = t >> 4;
    const unsigned i = ci * 16u + (unsigned)col;  // position inside the leg
    const size_t off = (poly << k) + ((size_t)h << (k - s_lo)) + i;
    u64 *base = data + off;
    const u64 *sbase = src + off;
    const int base2 = rg * 256 + col;
    u64 x[16];

    if (DIR == 0) {
#pragma unroll
        for (int jj = 0; jj < 16; jj++) x[jj] = sbase[(size_t)(rg + 16 * jj) << ls];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int half = 8 >> u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                if (jj & half) continue;
                const u64 w = tw[(1u << (s_lo + u)) + (h << u) + (unsigned)(jj >> (4 - u))];
                const u64 a = x[jj], v = G::mul(x[jj + half], w);
                x[jj] = G::add(a, v);
                x[jj + half] = G::sub(a, v);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) lds[pad(jj * 256 + t)] = x[jj];
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 16; jj++) x[jj] = lds[pad(base2 + jj * 16)];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int half = 8 >> u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                if (jj & half) continue;
                const u64 w = tw[(1u << (s_lo + 4 + u)) + (h << (4 + u)) + ((unsigned)rg << u) + (unsigned)(jj >> (4 - u))];
                const u64 a = x[jj], v = G::mul(x[jj + half], w);
                x[jj] = G::add(a, v);
                x[jj + half] = G::sub(a, v);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            u64 v = x[jj];
            if (TWIST) v = G::mul(v, twist[((size_t)((h << 8) + (unsigned)(16 * rg + jj)) << 12) + i]);
            base[(size_t)(16 * rg + jj) << ls] = v;
        }
    } else {
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            u64 v = sbase[(size_t)(16 * rg + jj) << ls];
            if (TWIST) v = G::mul(v, twist[((size_t)((h << 8) + (unsigned)(16 * rg + jj)) << 12) + i]);
            x[jj] = v;
        }
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            const int half = 8 >> u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                if (jj & half) continue;
                const u64 w = tw[(1u << (s_lo + 4 + u)) + (h << (4 + u)) + ((unsigned)rg << u) + (unsigned)(jj >> (4 - u))];
                const u64 a = x[jj], b = x[jj + half];
                x[jj] = G::add(a, b);
                x[jj + half] = G::mul(G::sub(a, b), w);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) lds[pad(base2 + jj * 16)] = x[jj];
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 16; jj++) x[jj] = lds[pad(jj * 256 + t)];
#pragma unroll
        for (int u = 3; u >= 0; u--) {
            const int half = 8 >> u;
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                if (jj & half) continue;
                const u64 w = tw[(1u << (s_lo + u)) + (h << u) + (unsigned)(jj >> (4 - u))];
                const u64 a = x[jj], b = x[jj + half];
                x[jj] = G::add(a, b);
                x[jj + half] = G::mul(G::sub(a, b), w);
            }
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) base[(size_t)(rg + 16 * jj) << ls] = x[jj];
    }
}

// ------------------------------------------------------------------------------------------------
// rows kernel: one workgroup = one 4096-coefficient tile, 256 lanes x 16 coefficients
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int pad(int pos) { return pos + (pos >> 4); }

// ------------------------------------------------------------------------------------------------
// cols256: the first EIGHT merged stages of a whole ring element (s_lo = 0, D = 256 * N2, N2 >= 256) with shift-only
// butterflies.  Same tile and exchange as strided256_kernel (256 legs of stride N2 x 16 consecutive columns), same
// result bit for bit, about a third fewer VALU instructions:
//   pass A  stages 0..3: their twiddles tw[2^u + b] = psi^brv_k(2^u + b) are 4th..32nd roots of unity, and
//           psi^(D/32) = omega_64 = 2^39, so tw[i] = 2^(39 brv_5(i)) for i < 32: compile-time shifts.
//   W layer block h (register slot) of the 16 blocks pass A leaves is a twisted-cyclic problem with root
//           gamma'_h = psi^(2 brv_4(h) + 1); leg rg of it is multiplied by gamma'_h^(rg N2) = theta^((2 brv_4(h) + 1) rg)
//           (theta = psi^(D/256) = 7^((p-1)/512) whatever D), the leg part of its twist, from a 256-entry table.
//   pass B  cyclic DFT_16 over the legs of one block (shift-only radix-16, like the rows kernel), output bit-reversed:
//           slot sigma is final block b = 16 h + sigma; its twist gamma_b^i (the rest of the twist times the four-step
//           cross twiddle: the same table the 4096-point rows kernels use, N2 entries per block) is applied on the way out.
// DIR 1 is the mirror image (unnormalised; D^-1 sits in the inverse twist table).  grid.x = npoly * N2 / 16.
// tools/model_fast_goldilocks.py: cols256_fwd / cols256_inv.
// ------------------------------------------------------------------------------------------------
constexpr int brv5(int i) { return ((i & 1) << 4) | ((i & 2) << 2) | (i & 4) | ((i & 8) >> 2) | ((i & 16) >> 4); }
constexpr int cols_tw_exp(int i) { return (39 * brv5(i)) % 192; }
// the merged negacyclic stages of pass A, phase by phase like the DFT_16 stages (dif_phased / dit_phased above): butterfly I of stage U
// pairs slots (J, J + (8 >> U)) with J = (I / half) 2 half + I % half and carries the twiddle 2^cols_tw_exp(2^U + (J >> (4 - U))).
// Forward: decimation in time with lazy legs (any representatives out: the W layer behind it is a general product); inverse:
// Gentleman-Sande, canonical throughout (its inputs come out of a general product, its results leave the library).
template <int U, bool INV>
struct ColsStageOf {
    template <int I>
    struct Bf {
        static constexpr int half = 8 >> U;
        static constexpr int lo = (I / half) * 2 * half + I % half, hi = lo + half;
        static constexpr int e = cols_tw_exp((1 << U) + (lo >> (4 - U)));
        static constexpr int E = INV ? (192 - e) % 192 : e;
    };
};
template <int U, int P = kPhased>
SR_HD void cols_stage_fwd(u64 *x) {
    stage_in_groups<ColsStageOf<U, false>::template Bf, true, 8, P, true>(x);
}
template <int U, int P = kPhased>
SR_HD void cols_stage_inv(u64 *x) {
    stage_in_groups<ColsStageOf<U, true>::template Bf, false, 8, P>(x);
}

// LC = log2 of the columns a workgroup owns: 16 << LC lanes, 256 legs x 2^LC consecutive columns (2^LC x 8-byte segments).
// Wider segments stream better (tools/ubench/strided_pattern.hip: 4.7 / 5.1 / 6.1 TB/s for 16 / 32 / 64 columns) at the price of
// fewer, larger workgroups.  LC = 4 pads the LDS tile (a 32-lane LDS group spans two legs); LC >= 5 needs no padding: every
// 32-lane group reads or writes 32 consecutive 8-byte words.
template <int LC>
struct ColsTile {
    static constexpr int C = 1 << LC;
    static constexpr int kLanes = 16 * C;
    static constexpr int kElems = LC == 4 ? kLdsElems : 256 * C;
    static __device__ __forceinline__ int idx(int leg, int col) {
        const int pos = leg * C + col;
        return LC == 4 ? pos + (pos >> 4) : pos;
    }
};
// tile = position of the workgroup's tile in the launch: ring element tile >> (log2 N2 - LC), column chunk in the low bits
template <int DIR, int LC>
__device__ __forceinline__ void cols256_tile(const unsigned tile, u64 *data, const u64 *src, int k, const u64 *__restrict__ wc,
                                             const u64 *__restrict__ twist, u64 *lds) {
    using CT = ColsTile<LC>;
    constexpr int C = CT::C;
    const int t = threadIdx.x;
    const int ls = k - 8;  // log2 N2
    const unsigned ci = tile & ((1u << (ls - LC)) - 1u);
    const size_t poly = tile >> (ls - LC);
    const int col = t & (C - 1), rg = t >> LC;
    const unsigned i = ci * (unsigned)C + (unsigned)col;  // column = position inside a leg
    // wave-uniform base + 32-bit byte offsets (a ring element is at most 8 MiB): one v_add_u32 per access
    char *pb = reinterpret_cast<char *>(data + (poly << k));
    const char *ps = reinterpret_cast<const char *>(src + (poly << k));  // src == data: in place; otherwise src is only read
    const char *tb = reinterpret_cast<const char *>(twist);
    const unsigned leg = 8u << ls;                                      // bytes between consecutive legs
    const unsigned offA = (((unsigned)rg << ls) + i) * 8u;              // leg rg (+ 16 jj)
    const unsigned offB = (((unsigned)rg << (ls + 4)) + i) * 8u;        // leg 16 rg (+ sigma) = final block b
    u64 x[16];

    if (DIR == 0) {
#pragma unroll
        for (int jj = 0; jj < 16; jj++) x[jj] = ld_stream(reinterpret_cast<const u64 *>(ps + (offA + (unsigned)jj * 16u * leg)));
        cols_stage_fwd<0>(x);
        cols_stage_fwd<1>(x);
        cols_stage_fwd<2>(x);
        cols_stage_fwd<3>(x);
#pragma unroll
        for (int h = 0; h < 16; h++) x[h] = G::mul(x[h], wc[h * 16 + rg]);
#pragma unroll
        for (int h = 0; h < 16; h++) lds[CT::idx(16 * h + rg, col)] = x[h];  // leg 16 h + rg, column col
        __syncthreads();
        u64 tw[16];
#pragma unroll
        for (int sg = 0; sg < 16; sg++) tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB + (unsigned)sg * leg));
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = lds[CT::idx(16 * rg + j, col)];  // block rg, leg j
        dft16_fwd_hot(x);
#pragma unroll
        for (int sg = 0; sg < 16; sg++)
            st_scratch(reinterpret_cast<u64 *>(pb + (offB + (unsigned)sg * leg)), G::mul(x[sg], tw[sg]));
    } else {
        u64 tw[16];
#pragma unroll
        for (int sg = 0; sg < 16; sg++) {
            x[sg] = ld_scratch(reinterpret_cast<const u64 *>(ps + (offB + (unsigned)sg * leg)));
            tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB + (unsigned)sg * leg));
        }
#pragma unroll
        for (int sg = 0; sg < 16; sg++) x[sg] = G::mul(x[sg], tw[sg]);
        dft16_inv_hot(x);
#pragma unroll
        for (int j = 0; j < 16; j++) lds[CT::idx(16 * rg + j, col)] = x[j];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 16; h++) x[h] = lds[CT::idx(16 * h + rg, col)];
#pragma unroll
        for (int h = 0; h < 16; h++) x[h] = G::mul(x[h], wc[h * 16 + rg]);
        cols_stage_inv<3>(x);
        cols_stage_inv<2>(x);
        cols_stage_inv<1>(x);
        cols_stage_inv<0>(x);
#pragma unroll
        for (int jj = 0; jj < 16; jj++) st_result(reinterpret_cast<u64 *>(pb + (offA + (unsigned)jj * 16u * leg)), x[jj]);
    }
}

// 2^kColsXcdGroup consecutive tiles (column chunks of one ring element: 2 KiB of every leg at 16 columns) per XCD turn
constexpr int kColsXcdGroup = 4;
template <int DIR, int LC>
__global__ __launch_bounds__(16 << LC, 4) void cols256_kernel(u64 *data, const u64 *src, int k, const u64 *__restrict__ wc,
                                                              const u64 *__restrict__ twist, unsigned grouped) {
    __shared__ u64 lds[ColsTile<LC>::kElems];
    __shared__ u64 wl[256];  // the W layer's table (16 of a lane's 48 global loads per tile)
    if (threadIdx.x < 256) wl[threadIdx.x] = wc[threadIdx.x];
    __syncthreads();
    cols256_tile<DIR, LC>(xcd_tile(blockIdx.x, kColsXcdGroup, grouped), data, src, k, wl, twist, lds);
}
// the forward passes of BOTH operands of a ring product in one launch: a one-dimensional grid of 2 * tiles workgroups, the XCD order
// taken over the whole range (the hardware deals workgroups to the XCDs by their LINEAR id, so a second grid dimension would put
// operand b's tiles on other XCDs than xcd_tile assumes whenever `tiles` is not a multiple of 8); virtual tile v < tiles is a's,
// the rest are b's.  `tiles` is a multiple of 2^kColsXcdGroup (N2 / 16 >= 16 column chunks per ring element), so no run of
// consecutive tiles straddles the two operands.
template <int LC>
__global__ __launch_bounds__(16 << LC, 4) void cols256_pair_kernel(u64 *data_a, const u64 *src_a, u64 *data_b, const u64 *src_b, int k,
                                                                   const u64 *__restrict__ wc, const u64 *__restrict__ twist,
                                                                   unsigned grouped, unsigned tiles) {
    __shared__ u64 lds[ColsTile<LC>::kElems];
    __shared__ u64 wl[256];
    if (threadIdx.x < 256) wl[threadIdx.x] = wc[threadIdx.x];
    __syncthreads();
    const unsigned v = xcd_tile(blockIdx.x, kColsXcdGroup, grouped);
    const bool second = v >= tiles;
    cols256_tile<0, LC>(second ? v - tiles : v, second ? data_b : data_a, second ? src_b : src_a, k, wl, twist, lds);
}

// ------------------------------------------------------------------------------------------------
// cols256_keep: the same column pass for the SMALL launches of the two-lane plans (gl_fast_*_lanes: 64 MiB of coefficients per
// launch, two streams).  A workgroup OWNS one column chunk ci and walks over the ring elements of the launch: the 16 twist factors a
// lane needs depend on the column and the leg, not on the ring element, so they are loaded once per workgroup instead of once per
// tile (one 8-byte L2 read per coefficient and pass otherwise: 26 GB per config-2 batch that no HBM counter shows), and the
// 256-entry W table sits in 2 KiB of LDS.  Compiled for four workgroups per CU (a 128-VGPR budget) the scheduler trades registers
// for spills (48-84 bytes; the round-4 harness lost 11 %); compiled for THREE (__launch_bounds__(256, 3): up to 168) the allocator
// ends at 120 / 126 VGPRs with nothing spilled -- which the hardware still packs four to a CU (tests/test_isa_budget.py holds it
// there).  Alone on the chip the kernel is slower than cols256_kernel (3.65 against 3.27 ms per config-2 batch on one stream: one
// more barrier per tile, no fresh workgroup overlapping the old one's tail); beside the other lane's kernels it is faster -- 14.9-15.05
// against 15.55 ms per batch in the harness, 14.96-15.08 against 15.64-15.70 ms through the library (tools/bench_keep_cols.py,
// DESIGN.md 6) -- so only the lane plans use it; the one-stream plan keeps cols256_kernel.
// Workgroup -> (XCD, ci, group): blockIdx & 7 is the XCD the hardware gives the workgroup; inside an XCD slot = blockIdx >> 3 =
// group * chunks + ci; the workgroup handles ring elements xcd + 8 (group + groups r), r = 0, 1, ...: all column chunks of a ring
// element are in flight on ONE XCD at the same time (what xcd_tile() arranges for the plain launch).  Same values bit for bit
// (tests/test_gpu_parity.py: test_keep_and_plain_column_passes_agree).  grid.x = 8 * (N2 / 16) * groups; npoly a multiple of 8.
// data2 / src2 != nullptr: the forward pass of a SECOND operand (b of a ring product) in the same launch, walked over behind the first
// by the same workgroups with the same factors: one launch boundary per lane chunk fewer (-1.1 % on the config-2 step).
// ------------------------------------------------------------------------------------------------
template <int DIR>
__global__ __launch_bounds__(256, 3) void cols256_keep_kernel(u64 *data, const u64 *src, u64 *data2, const u64 *src2, int k,
                                                              const u64 *__restrict__ wc, const u64 *__restrict__ twist, unsigned npoly,
                                                              unsigned groups) {
    using CT = ColsTile<4>;
    __shared__ u64 lds[CT::kElems];
    __shared__ u64 wl[256];
    const int t = threadIdx.x;
    wl[t] = wc[t];
    const int ls = k - 8;  // log2 N2
    const unsigned chunks = 1u << (ls - 4);
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned ci = slot & (chunks - 1u), grp = slot >> (ls - 4);
    const int col = t & 15, rg = t >> 4;
    const unsigned i = ci * 16u + (unsigned)col;
    const char *tb = reinterpret_cast<const char *>(twist);
    const unsigned leg = 8u << ls;
    const unsigned offA0 = (((unsigned)rg << ls) + i) * 8u;          // leg rg (+ 16 jj)
    const unsigned offB0 = (((unsigned)rg << (ls + 4)) + i) * 8u;    // leg 16 rg (+ sigma) = final block b
    u64 x[16], tw[16];
#pragma unroll
    for (int sg = 0; sg < 16; sg++) tw[sg] = *reinterpret_cast<const u64 *>(tb + (offB0 + (unsigned)sg * leg));
    __syncthreads();
    // data2 != nullptr: a second operand (the forward passes of a and b of a ring product) in the same launch, behind the first
    const unsigned total = data2 ? 2u * npoly : npoly;
    for (unsigned it = xcd + 8u * grp; it < total; it += 8u * groups) {  // uniform per workgroup: every lane reaches every barrier
        const bool second = it >= npoly;
        const unsigned poly = second ? it - npoly : it;
        char *pb = reinterpret_cast<char *>((second ? data2 : data) + ((size_t)poly << k));
        const char *ps = reinterpret_cast<const char *>((second ? src2 : src) + ((size_t)poly << k));
        // opaque per iteration: otherwise the 32 per-access offsets are hoisted out of the loop as invariants and live across it
        unsigned offA = offA0, offB = offB0;
        asm volatile("" : "+v"(offA), "+v"(offB));
        if (DIR == 0) {
#pragma unroll
            for (int jj = 0; jj < 16; jj++) x[jj] = ld_stream(reinterpret_cast<const u64 *>(ps + (offA + (unsigned)jj * 16u * leg)));
            cols_stage_fwd<0>(x);
            cols_stage_fwd<1>(x);
            cols_stage_fwd<2>(x);
            cols_stage_fwd<3>(x);
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if ((h & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // keeps the 16 table reads from all being hoisted in front
            }
#pragma unroll
            for (int h = 0; h < 16; h++) lds[CT::idx(16 * h + rg, col)] = x[h];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) x[j] = lds[CT::idx(16 * rg + j, col)];
            dft16_fwd_hot(x);
#pragma unroll
            for (int sg = 0; sg < 16; sg++) st_scratch(reinterpret_cast<u64 *>(pb + (offB + (unsigned)sg * leg)), G::mul(x[sg], tw[sg]));
        } else {
#pragma unroll
            for (int sg = 0; sg < 16; sg++) x[sg] = ld_scratch(reinterpret_cast<const u64 *>(ps + (offB + (unsigned)sg * leg)));
#pragma unroll
            for (int sg = 0; sg < 16; sg++) {
                x[sg] = G::mul(x[sg], tw[sg]);
                if ((sg & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            dft16_inv_hot(x);
#pragma unroll
            for (int j = 0; j < 16; j++) lds[CT::idx(16 * rg + j, col)] = x[j];
            __syncthreads();
#pragma unroll
            for (int h = 0; h < 16; h++) x[h] = lds[CT::idx(16 * h + rg, col)];
#pragma unroll
            for (int h = 0; h < 16; h++) {
                x[h] = G::mul(x[h], wl[h * 16 + rg]);
                if ((h & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            cols_stage_inv<3>(x);
            cols_stage_inv<2>(x);
            cols_stage_inv<1>(x);
            cols_stage_inv<0>(x);
#pragma unroll
            for (int jj = 0; jj < 16; jj++) st_result(reinterpret_cast<u64 *>(pb + (offA + (unsigned)jj * 16u * leg)), x[jj]);
        }
        __syncthreads();  // every lane has read the exchange before the next ring element's writes land
    }
}

// ------------------------------------------------------------------------------------------------
// rows kernel (4096-coefficient tiles), continued: the tile transforms
//   Q = 0, TW = false: D >= 8192, the tile is one 4096-block already twisted by the strided pass: cyclic DFT_4096.
//   TW = true (D = 4096 >> Q <= 4096): the tile holds 2^Q whole ring elements; the stride-256 pass runs the last 4 - Q stages
//   of the radix-16 only; the negacyclic twist psi^(256 row + t) is split into a compile-time shift per register slot
//   ((psi^256)^row, twist_rows) and the column factor psi^t, which commutes with that pass and lives in the w1 table;
//   nvalid (a multiple of D) guards a ragged last tile.
// CANON (plain forward transform): the slots leave the library as they are, so the last network runs canonical butterflies.
// Twisted 4096-point blocks (TW = false, Q = 0) take the lazy DIT networks like the D = 2^16 kernels; whole-ring-element tiles and the
// 512..2048-point blocks of D = 2^17..2^19 keep the DIF ones (their partial networks dft16_fwd_q start in the middle of a DFT_16).
template <int Q, bool TW, int P = kPhased, bool CANON = false>
__device__ __forceinline__ void tile_fwd(const u64 *__restrict__ src, u64 *lds, const int t, const Tables &T, u64 *x,
                                         int nvalid) {
    constexpr bool LZ = !TW && Q == 0;  // Q > 0: 512..2048-point blocks start inside a DFT_16
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int pos = j * 256 + t;
        // ragged last tile (TW only): out-of-range lanes re-read the last valid coefficient; their results belong to ring
        // elements that do not exist and are never stored, and no pass mixes ring elements, so no branch is needed
        x[j] = TW ? ld_stream(src + (pos < nvalid ? pos : nvalid - 1)) : ld_scratch(src + pos);  // TW: operands; else column-pass output
    }
    if (TW) twist_rows<Q, false>(x, std::make_integer_sequence<int, 16>{});
    if constexpr (LZ) dft16_fwd_dit<P, true>(x);
    else dft16_fwd_q<Q, P>(x);
    if (TW) x[0] = G::mul(x[0], T.w1f[t]);  // psi^t: with the twist merged in, slot 0 is no longer multiplied by 1
#pragma unroll
    for (int r = 1; r < 16; r++) x[r] = G::mul(x[r], T.w1f[r * 256 + t]);
#pragma unroll
    for (int r = 0; r < 16; r++) lds[pad(r * 256 + t)] = x[r];
    __syncthreads();
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = lds[pad(base2 + j * 16)];
    if constexpr (LZ) dft16_fwd_dit<P, true>(x);
    else dft16_fwd<P>(x);
#pragma unroll
    for (int s = 1; s < 16; s++) x[s] = G::mul(x[s], T.w2f[s * 16 + i0]);
#pragma unroll
    for (int s = 0; s < 16; s++) lds[pad(base2 + s * 16)] = x[s];  // the very slots this lane just read
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = lds[17 * t + j];
    if constexpr (LZ) dft16_fwd_dit<P, !CANON>(x);
    else dft16_fwd<P>(x);
}

// inverse of tile_fwd; x[] holds positions 16 t .. 16 t + 15 on entry.  w1i is the plain or the fused-product table.
// Without TW the result is 4096 x the cyclic inverse and the strided inverse pass finishes the job.
template <int Q, bool TW, int P = kPhased>
__device__ __forceinline__ void tile_inv(u64 *x, u64 *lds, const int t, const Tables &T, const u64 *w1i,
                                         u64 *__restrict__ dst, int nvalid) {
    constexpr bool LZ = !TW && Q == 0;  // results go on to an inverse column pass that multiplies first
    dft16_inv<P, LZ>(x);
#pragma unroll
    for (int j = 0; j < 16; j++) lds[17 * t + j] = x[j];
    __syncthreads();
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
#pragma unroll
    for (int s = 0; s < 16; s++) x[s] = lds[pad(base2 + s * 16)];
#pragma unroll
    for (int s = 1; s < 16; s++) x[s] = G::mul(x[s], T.w2i[s * 16 + i0]);
    if (LZ) x[0] = G::canon(x[0]);  // see tile256_inv
    dft16_inv<P, LZ>(x);
#pragma unroll
    for (int j = 0; j < 16; j++) lds[pad(base2 + j * 16)] = x[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = lds[pad(r * 256 + t)];
    if (TW) x[0] = G::mul(x[0], w1i[t]);  // psi^-t * D^-1
#pragma unroll
    for (int r = 1; r < 16; r++) x[r] = G::mul(x[r], w1i[r * 256 + t]);
    if (LZ) x[0] = G::canon(x[0]);
    if constexpr (LZ) dft16_inv<P, true>(x);
    else dft16_inv_q<Q, P>(x);
    if (TW) twist_rows<Q, true>(x, std::make_integer_sequence<int, 16>{});
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const int pos = j * 256 + t;
        if (!TW) st_scratch(dst + pos, x[j]);  // goes on to the inverse column pass
        else if (pos < nvalid) st_result(dst + pos, x[j]);
    }
}

// MODE 0: a -> crt(a) in place; 1: a -> icrt(a) in place; 2: out = icrt(crt(a) (.) crt(b)) tile by tile;
// 3: out = icrt(crt(a) (.) b) with b already in CRT/NTT form (the constant-operand product: one transform fewer).
// n_total = flat coefficient count of the batch (only consulted when TW: ragged last tile).
// waves per SIMD: 4; whole-ring-element tiles 3 (2 / 3 / 4: 16.0 / 13.9 / 14.1 ms for the fused D = 4096 product, round 1)
template <int MODE, int Q, bool TW, int PH = (MODE >= 2 ? (TW ? -1 : 0) : kPhased)>
__global__ __launch_bounds__(256, TW ? 3 : 4) void rows_kernel(u64 *a, const u64 *b, u64 *out, Tables T, const u64 *w1i,
                                                                   size_t n_total) {
    __shared__ u64 lds[kLdsElems];
    __shared__ u64 wl[512];  // the two 256-entry table layers w2f, w2i (as in rows256_kernel)
    const int t = threadIdx.x;
    wl[t] = T.w2f[t];
    wl[256 + t] = T.w2i[t];
    __syncthreads();
    T.w2f = wl;
    T.w2i = wl + 256;
    const size_t base = (size_t)blockIdx.x * kTile;
    int nvalid = kTile;
    if (TW) nvalid = n_total - base < (size_t)kTile ? (int)(n_total - base) : kTile;
    // the fused products hold a's transformed tile in registers while b's runs: butterfly by butterfly there (no spills); the
    // stand-alone transforms take the phased stages (PH's default)
    constexpr int P = PH;
    u64 A[16];
    if (MODE == 1) {
        // lane-contiguous global load, then an exchange into the 16-contiguous-per-lane layout of the first pass
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int pos = j * 256 + t;
            lds[pad(pos)] = a[base + ((!TW || pos < nvalid) ? pos : nvalid - 1)];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) A[j] = lds[17 * t + j];
    } else {
        tile_fwd<Q, TW, P, MODE == 0>(a + base, lds, t, T, A, nvalid);
        if (MODE == 0) {
            // results sit 16-contiguous per lane; one more exchange makes the global store lane-contiguous
#pragma unroll
            for (int j = 0; j < 16; j++) lds[17 * t + j] = A[j];  // own pass-3 slots
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int pos = j * 256 + t;
                if (!TW || pos < nvalid) st_slots(a + base + pos, lds[pad(pos)]);
            }
            return;
        }
        u64 B[16];
        __syncthreads();  // everyone has read its pass-3 slots of a before b's pass-1 writes land
        if (MODE == 3) {  // b is in NTT order already: lane-contiguous load, one exchange into 16 consecutive slots per lane
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int pos = j * 256 + t;
                lds[pad(pos)] = b[base + ((!TW || pos < nvalid) ? pos : nvalid - 1)];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) B[j] = lds[17 * t + j];
        } else {
            tile_fwd<Q, TW, P>(b + base, lds, t, T, B, nvalid);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) A[j] = G::mul(A[j], B[j]);
        // no barrier: tile_inv first writes the lane's own slots 17 t + j, which only this lane has just read
    }
    tile_inv<Q, TW, P>(A, lds, t, T, w1i, out + base, nvalid);
}

// ------------------------------------------------------------------------------------------------
// rows256: D = 2^16 behind cols256 -- a tile is 16 twisted blocks of 256 coefficients, each a cyclic DFT_256 = 16 x 16:
// the last two register passes of the 4096-point kernel, one LDS exchange per transform.  Lane (rho, i0) works on block
// rho; global accesses are 128-byte segments (16 lanes x 8 B).  MODE as rows_kernel.
// tools/model_fast_goldilocks.py: rows256_fwd / rows256_inv.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile256_load(const u64 *__restrict__ src, const int t, u64 *x) {
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = ld_scratch(src + base2 + j * 16);
}
// x holds the lane's 16 coefficients (tile256_load) on entry, its 16 consecutive NTT slots on return
// CANON: the slots leave the library as they are (plain forward transform): the last network runs canonical butterflies
template <bool CANON = false>
__device__ __forceinline__ void tile256_fwd_regs(u64 *lds, const int t, const Tables &T, u64 *x) {
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
    dft16_fwd_hot(x);
#pragma unroll
    for (int s = 1; s < 16; s++) x[s] = G::mul(x[s], T.w2f[s * 16 + i0]);
#pragma unroll
    for (int s = 0; s < 16; s++) lds[pad(base2 + s * 16)] = x[s];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; j++) x[j] = lds[17 * t + j];
    dft16_fwd_hot<CANON>(x);
}
template <bool CANON = false>
__device__ __forceinline__ void tile256_fwd(const u64 *__restrict__ src, u64 *lds, const int t, const Tables &T, u64 *x) {
    tile256_load(src, t, x);
    tile256_fwd_regs<CANON>(lds, t, T, x);
}
__device__ __forceinline__ void tile256_inv(u64 *x, u64 *lds, const int t, const Tables &T, u64 *__restrict__ dst) {
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
    dft16_inv_hot(x);
#pragma unroll
    for (int j = 0; j < 16; j++) lds[17 * t + j] = x[j];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) x[s] = lds[pad(base2 + s * 16)];
#pragma unroll
    for (int s = 1; s < 16; s++) x[s] = G::mul(x[s], T.w2i[s * 16 + i0]);
    // x[0] skips the table product (its factor is 1) but is slot i0 of ANOTHER lane's network, a lazy representative: the twiddle-1
    // butterflies of the next network want it canonical
    x[0] = G::canon(x[0]);
    dft16_inv_hot(x);
#pragma unroll
    for (int j = 0; j < 16; j++) st_scratch(dst + base2 + j * 16, x[j]);
}
template <int MODE>
__device__ __forceinline__ void rows256_tile(const unsigned tile, u64 *a, const u64 *b, u64 *out, const Tables &T, u64 *lds) {
    const int t = threadIdx.x;
    const size_t base = (size_t)tile * kTile;
    u64 A[16];
    if (MODE == 1) {
        // the inverse starts from 16 consecutive slots per lane: lane-contiguous load, one exchange
#pragma unroll
        for (int j = 0; j < 16; j++) lds[pad(j * 256 + t)] = a[base + j * 256 + t];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) A[j] = lds[17 * t + j];  // own slots from here on: no barrier before tile256_inv's writes
    } else {
        tile256_fwd<MODE == 0>(a + base, lds, t, T, A);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) lds[17 * t + j] = A[j];  // own slots
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) st_slots(a + base + j * 256 + t, lds[pad(j * 256 + t)]);
            return;
        }
        u64 B[16];
        __syncthreads();  // every lane has read a's exchange before b's lands
        if (MODE == 3) {  // b already in NTT order (see rows_kernel)
#pragma unroll
            for (int j = 0; j < 16; j++) lds[pad(j * 256 + t)] = b[base + j * 256 + t];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) B[j] = lds[17 * t + j];
        } else {
            tile256_fwd(b + base, lds, t, T, B);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) A[j] = G::mul(A[j], B[j]);
    }
    tile256_inv(A, lds, t, T, out + base);
}
// The two 256-entry table layers (w2f, w2i: 45 of the fused product's 93 global loads per lane and tile, 24 GB of L2 reads per
// config-2 batch) are copied into 4 KiB of LDS first: 38 KiB per workgroup, four still fit a CU; -0.9 % on the two-lane step
// (harness, three alternations: 14.97 / 15.04 / 14.97 against 15.10 / 15.16 / 15.24 ms), 97 instead of 112 VGPRs.
template <int MODE>
__global__ __launch_bounds__(256, 4) void rows256_kernel(u64 *a, const u64 *b, u64 *out, Tables T) {
    __shared__ u64 lds[kLdsElems];
    __shared__ u64 wl[512];
    wl[threadIdx.x] = T.w2f[threadIdx.x];
    wl[256 + threadIdx.x] = T.w2i[threadIdx.x];
    __syncthreads();
    T.w2f = wl;
    T.w2i = wl + 256;
    rows256_tile<MODE>(blockIdx.x, a, b, out, T, lds);
}


// ------------------------------------------------------------------------------------------------
// table builder.  pows[j] = psi^(2^j), ipows[j] = psi^-(2^j), j = 0..k  (psi^(2^k) = -1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 pow_from_bits(const u64 *pw, unsigned e, int k) {
    u64 acc = 1;
    for (int j = 0; j <= k; j++)
        if ((e >> j) & 1u) acc = G::mul(acc, pw[j]);
    return acc;
}
// c = number of merged stages the column passes run = log2 of the number of twisted blocks (of N2 = D >> c coefficients)
__global__ void build_tables_kernel(int k, int c, const u64 *pows, const u64 *ipows, u64 dinv, u64 dinv_mul, u64 *twist_f,
                                    u64 *twist_i_plain, u64 *twist_i_mul, u64 *w1f, u64 *w1i, u64 *w1i_mul, u64 *w2f,
                                    u64 *w2i, u64 *wcf, u64 *wci) {
    const int q = 12 - (k - c);          // log2 of (ring elements | twisted blocks) per 4096-coefficient tile
    const size_t d = (size_t)1 << k;
    const size_t n = d > 4096 ? d : 4096;
    const unsigned mask2d = (unsigned)(2 * d - 1);
    for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < d) {
            unsigned b = (unsigned)(idx >> (k - c)), i = (unsigned)(idx & ((d >> c) - 1));
            unsigned e = (unsigned)(((unsigned long long)(2 * bitrev(b, c) + 1) * i) & mask2d);
            twist_f[idx] = pow_from_bits(pows, e, k);
            u64 inv = pow_from_bits(ipows, e, k);
            twist_i_plain[idx] = G::mul(inv, dinv);
            twist_i_mul[idx] = G::mul(inv, dinv_mul);
        }
        if (idx < 4096) {
            // omega_N^(i0 * m0), N = D >> c = the cyclic size the stride-256 pass starts; slot r of a lane belongs to
            // ring element (block) r >> (4 - q) of the tile and carries output index brv_(4-q)(r mod 2^(4-q)) of its sub-DFT
            unsigned r = (unsigned)(idx >> 8), i0 = (unsigned)(idx & 255);
            unsigned m0 = bitrev(r & ((1u << (4 - q)) - 1u), 4 - q);
            unsigned e1 = (unsigned)((((unsigned long long)i0 * m0) << (c + 1)) & mask2d);  // omega_N = psi^(2D/N) = psi^(2^(c+1))
            if (k <= 12) e1 = (e1 + i0) & mask2d;  // column part psi^i0 of the twist, merged (tile_fwd)
            w1f[idx] = pow_from_bits(pows, e1, k);
            const u64 inv1 = pow_from_bits(ipows, e1, k);
            w1i[idx] = k <= 12 ? G::mul(inv1, dinv) : inv1;
            w1i_mul[idx] = k <= 12 ? G::mul(inv1, dinv_mul) : inv1;
        }
        if (idx < 256) {
            unsigned s = (unsigned)(idx >> 4), i0 = (unsigned)(idx & 15);
            unsigned e2 = (unsigned)((((unsigned long long)i0 * bitrev(s, 4)) << (k - 7)) & mask2d);  // omega_256 = psi^(2D/256)
            w2f[idx] = pow_from_bits(pows, e2, k);
            w2i[idx] = pow_from_bits(ipows, e2, k);
            // cols256 W layer: theta^((2 brv4(h) + 1) rg), theta = psi^(D/256)
            unsigned e3 = (unsigned)((((unsigned long long)(2 * bitrev(s, 4) + 1) * i0) << (k - 8)) & mask2d);
            wcf[idx] = pow_from_bits(pows, e3, k);
            wci[idx] = pow_from_bits(ipows, e3, k);
        }
    }
}

}  // namespace gl

// ---- host side ------------------------------------------------------------------------------------
struct GoldilocksFastTables {
    int k = -1;
    int c = 0;               // merged stages run by the column passes; the rows kernels see blocks of D >> c coefficients
    bool cols256 = false;    // c == 8 in one cols256 launch (2^16 <= D <= 2^20)
    bool keep_cols = true;   // lane plans: cols256_keep_kernel (sr_plan flag SR_PLAN_GL_PLAIN_COLS clears it)
    bool split_rows = false; // A/B plan (sr_plan flag SR_PLAN_GL_SPLIT_ROWS): the fused rows kernel of a ring product as two launches --
                             // crt of b's tiles in place in the scratch (rows<0>), then the constant-operand product (rows<3>)
    bool ready = false;
    gl::Tables t{};
    size_t chunk_polys = 0;  // ring products: elements per chunk of launches (0 = as many as the scratch holds)
    // optional per-launch timing hooks (set by capi.hip): tag 0 strided fwd, 1 rows, 2 strided inv
    void (*prof_begin)(void *user, int tag, hipStream_t st) = nullptr;
    void (*prof_end)(void *user, hipStream_t st) = nullptr;
    void *prof_user = nullptr;
};
struct GlProfScope {
    const GoldilocksFastTables &f;
    hipStream_t st;
    GlProfScope(const GoldilocksFastTables &ff, int tag, hipStream_t s) : f(ff), st(s) {
        if (f.prof_begin) f.prof_begin(f.prof_user, tag, st);
    }
    ~GlProfScope() {
        if (f.prof_end) f.prof_end(f.prof_user, st);
    }
};

inline bool gl_fast_supported(const GoldilocksFastTables &f) { return f.ready; }
inline size_t gl_fast_extra_bytes(int k) {
    if (k < 8 || k > 22) return 0;  // D = 256 .. 2^22
    return (((size_t)3 << k) + 3 * 4096 + 4 * 256) * sizeof(uint64_t);
}
// extra = device memory of gl_fast_extra_bytes(k) bytes, placed right behind [tw | itw] in the context's
// twiddle block so that one broadcast ships everything.
inline int gl_fast_init(GoldilocksFastTables &f, int k, const uint64_t *tw, const uint64_t *itw, uint64_t *extra,
                        const uint64_t *host_pows, const uint64_t *host_ipows, uint64_t dinv, uint64_t dinv_mul,
                        bool allow_cols256, size_t chunk_polys, hipStream_t st) {
    f.k = k;
    f.ready = false;
    if (gl_fast_extra_bytes(k) == 0) return 0;
    const size_t d = (size_t)1 << k;
    uint64_t *p = extra;
    uint64_t *twist_f = p;            p += d;
    uint64_t *twist_ip = p;           p += d;
    uint64_t *twist_im = p;           p += d;
    uint64_t *w1f = p;                p += 4096;
    uint64_t *w1i = p;                p += 4096;
    uint64_t *w1im = p;               p += 4096;
    uint64_t *w2f = p;                p += 256;
    uint64_t *w2i = p;                p += 256;
    uint64_t *wcf = p;                p += 256;
    uint64_t *wci = p;
    // plan: 2^16 <= D <= 2^20 runs all its column stages (8) in one shift-only cols256 launch and leaves D / 256-point
    // rows; otherwise the rows are 4096 points (or whole ring elements, D <= 4096).  allow_cols256 = false (sr_plan flag
    // SR_PLAN_GL_NO_COLS256): the older plan.
    f.cols256 = k >= 16 && k <= 20 && allow_cols256;
    f.chunk_polys = chunk_polys;
    f.c = f.cols256 ? 8 : (k > 12 ? k - 12 : 0);
    uint64_t *d_pows = nullptr;
    if (hipMalloc(&d_pows, 2 * (k + 1) * sizeof(uint64_t)) != hipSuccess) return 1;
    bool ok = hipMemcpy(d_pows, host_pows, (k + 1) * 8, hipMemcpyHostToDevice) == hipSuccess &&
              hipMemcpy(d_pows + k + 1, host_ipows, (k + 1) * 8, hipMemcpyHostToDevice) == hipSuccess;
    if (ok) {
        unsigned blocks = (unsigned)(((d > 4096 ? d : 4096) + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gl::build_tables_kernel, dim3(blocks), dim3(256), 0, st, k, f.c, d_pows, d_pows + k + 1, dinv, dinv_mul,
                           twist_f, twist_ip, twist_im, w1f, w1i, w1im, w2f, w2i, wcf, wci);
        ok = hipGetLastError() == hipSuccess;
        ok = hipStreamSynchronize(st) == hipSuccess && ok;  // (also after a failed launch: nothing may still read d_pows)
    }
    (void)hipFree(d_pows);  // on every path
    if (!ok) return 1;
    f.t = gl::Tables{tw, itw, twist_f, twist_ip, twist_im, w1f, w1i, w1im, w2f, w2i, wcf, wci};
    f.ready = true;
    return 0;
}
inline void gl_fast_destroy(GoldilocksFastTables &f) { f.ready = false; }

template <int DIR, bool TWIST>
inline int gl_launch_strided(const GoldilocksFastTables &f, int M, uint64_t *data, const uint64_t *src, int k, int s_lo,
                             size_t npoly, const uint64_t *tw, const uint64_t *twist, hipStream_t st) {
    GlProfScope ps(f, DIR == 0 ? 0 : 2, st);
    const size_t S = (size_t)1 << (k - s_lo - M);
    const size_t blocks = npoly * ((size_t)1 << s_lo) * (S >> 8);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    dim3 g((unsigned)blocks), b(256);
    switch (M) {
        case 0: hipLaunchKernelGGL((gl::strided_kernel<0, DIR, TWIST>), g, b, 0, st, data, src, k, s_lo, tw, twist); break;
        case 1: hipLaunchKernelGGL((gl::strided_kernel<1, DIR, TWIST>), g, b, 0, st, data, src, k, s_lo, tw, twist); break;
        case 2: hipLaunchKernelGGL((gl::strided_kernel<2, DIR, TWIST>), g, b, 0, st, data, src, k, s_lo, tw, twist); break;
        case 3: hipLaunchKernelGGL((gl::strided_kernel<3, DIR, TWIST>), g, b, 0, st, data, src, k, s_lo, tw, twist); break;
        case 4: hipLaunchKernelGGL((gl::strided_kernel<4, DIR, TWIST>), g, b, 0, st, data, src, k, s_lo, tw, twist); break;
        default: return 1;
    }
    return hipGetLastError() != hipSuccess;
}
template <int DIR, bool TWIST>
inline int gl_launch_strided256(const GoldilocksFastTables &f, uint64_t *data, const uint64_t *src, int k, int s_lo, size_t npoly,
                                const uint64_t *tw, const uint64_t *twist, hipStream_t st) {
    GlProfScope ps(f, DIR == 0 ? 0 : 2, st);
    const size_t S = (size_t)1 << (k - s_lo - 8);
    const size_t blocks = npoly * ((size_t)1 << s_lo) * (S >> 4);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((gl::strided256_kernel<DIR, TWIST>), dim3((unsigned)blocks), dim3(256), 0, st, data, src, k, s_lo, tw,
                       twist);
    return hipGetLastError() != hipSuccess;
}
// split the c = k - 12 strided stages into passes: 8 = one 256-leg pass through LDS (D >= 2^20), otherwise
// register passes of at most 4 stages.  The 256-leg pass, when present, is the one next to the rows kernel.
inline int gl_plan(int c, int *ms) {
    int n = 0;
    const bool big = c >= 8;
    if (big) c -= 8;
    while (c > 0) {
        int m = c > 4 ? (c + 1) / 2 : c;
        ms[n++] = m;
        c -= m;
    }
    if (big) ms[n++] = 8;
    return n;  // 0 passes for D <= 4096: the rows kernel applies the twist itself
}
// columns per workgroup (2^LC): measured at D = 2^16, batch 2^14 (tools/ubench/gl_bench.hip): forward 3.86 / 3.75 / 4.38 ms and
// inverse 4.26 / 4.39 / 5.07 ms for LC = 4 / 5 / 6 -- wider segments stream better (strided_pattern.hip) but one or two big
// workgroups per CU overlap their load, exchange and store phases worse than four small ones.  With non-temporal coefficient
// accesses LC = 4 runs 3.70 / 4.09 ms and LC = 5 3.73 / 4.21 ms: LC = 4 both ways.
template <int DIR>
inline int gl_launch_cols256(const GoldilocksFastTables &f, uint64_t *data, const uint64_t *src, size_t npoly,
                             const uint64_t *wc, const uint64_t *twist, hipStream_t st) {
    GlProfScope ps(f, DIR == 0 ? 0 : 2, st);
    constexpr int LC = 4;
    const size_t blocks = npoly << (f.k - 8 - LC);  // N2 / 2^LC per ring element
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    const unsigned grouped = xcd_grouped_tiles(blocks, gl::kColsXcdGroup);
    hipLaunchKernelGGL((gl::cols256_kernel<DIR, LC>), dim3((unsigned)blocks), dim3(16 << LC), 0, st, data, src, f.k, wc, twist,
                       grouped);
    return hipGetLastError() != hipSuccess;
}
// the column pass of a LANE chunk: the workgroup-owns-its-columns kernel when the launch is a whole number of XCD rounds
template <int DIR>
inline int gl_launch_cols256_lane(const GoldilocksFastTables &f, uint64_t *data, const uint64_t *src, size_t npoly, const uint64_t *wc,
                                  const uint64_t *twist, hipStream_t st) {
    const unsigned chunks = 1u << (f.k - 12);                         // N2 / 16 column chunks per ring element
    unsigned groups = chunks >= 128 ? 1u : 128u / chunks;             // 1024 workgroups per launch where the elements allow it
    // worth it only when a workgroup walks over at least two ring elements (one element per workgroup is the plain kernel at three
    // workgroups per CU: config 4's chunks of 8 elements lost 7 % that way)
    // D = 2^16 only: at 2^17 .. 2^19 (32 .. 128 column chunks per element, two elements per workgroup at the default lane chunks) the
    // library-level A/B lost 3-4 % (tools/bench_keep_cols.py)
    if (!f.keep_cols || f.k != 16 || (npoly & 7u) != 0 || npoly < 16u || npoly > 0x7FFFFFFFull)
        return gl_launch_cols256<DIR>(f, data, src, npoly, wc, twist, st);
    if (groups > npoly / 16) groups = (unsigned)(npoly / 16);
    GlProfScope ps(f, DIR == 0 ? 0 : 2, st);
    hipLaunchKernelGGL((gl::cols256_keep_kernel<DIR>), dim3(8u * chunks * groups), dim3(256), 0, st, data, src, (uint64_t *)nullptr,
                       (const uint64_t *)nullptr, f.k, wc, twist, (unsigned)npoly, groups);
    return hipGetLastError() != hipSuccess;
}
// the plain forward column pass over both operands of a ring product in one launch (grid.y picks the operand)
inline int gl_launch_cols256_pair_plain(const GoldilocksFastTables &f, uint64_t *da, const uint64_t *sa, uint64_t *db, const uint64_t *sb,
                                        size_t npoly, hipStream_t st) {
    GlProfScope ps(f, 0, st);
    constexpr int LC = 4;
    const size_t blocks = npoly << (f.k - 8 - LC);
    if (blocks == 0 || blocks > 0x3FFFFFFFull) return 1;
    hipLaunchKernelGGL((gl::cols256_pair_kernel<LC>), dim3((unsigned)(2 * blocks)), dim3(16 << LC), 0, st, da, sa, db, sb, f.k, f.t.wcf,
                       f.t.twist_f, xcd_grouped_tiles(2 * blocks, gl::kColsXcdGroup), (unsigned)blocks);
    return hipGetLastError() != hipSuccess;
}
// the forward column passes of BOTH operands of a ring product in one launch (the same workgroups, their twist factors loaded once,
// walk over the elements of a and then of b); falls back to two launches where the persistent kernel does not apply
inline int gl_launch_cols256_lane_pair(const GoldilocksFastTables &f, uint64_t *da, const uint64_t *sa, uint64_t *db, const uint64_t *sb,
                                       size_t npoly, hipStream_t st) {
    const unsigned chunks = 1u << (f.k - 12);
    unsigned groups = chunks >= 128 ? 1u : 128u / chunks;
    if (!f.keep_cols || f.k != 16 || (npoly & 7u) != 0 || npoly < 16u || npoly > 0x3FFFFFFFull)
        return gl_launch_cols256_pair_plain(f, da, sa, db, sb, npoly, st);
    if (groups > npoly / 16) groups = (unsigned)(npoly / 16);
    GlProfScope ps(f, 0, st);
    hipLaunchKernelGGL((gl::cols256_keep_kernel<0>), dim3(8u * chunks * groups), dim3(256), 0, st, da, sa, db, sb, f.k, f.t.wcf, f.t.twist_f,
                       (unsigned)npoly, groups);
    return hipGetLastError() != hipSuccess;
}
// forward column stages of npoly ring elements: src -> d (src == d: in place).  Only the first pass reads src.
inline int gl_strided_fwd(const GoldilocksFastTables &f, uint64_t *d, const uint64_t *src, size_t npoly, hipStream_t st) {
    if (f.cols256) return gl_launch_cols256<0>(f, d, src, npoly, f.t.wcf, f.t.twist_f, st);
    int ms[8];
    const int n = gl_plan(f.k > 12 ? f.k - 12 : 0, ms);
    int s_lo = 0;
    for (int p = 0; p < n; p++) {
        const bool last = p == n - 1;
        const uint64_t *from = p == 0 ? src : d;
        int rc;
        if (ms[p] == 8)
            rc = last ? gl_launch_strided256<0, true>(f, d, from, f.k, s_lo, npoly, f.t.tw, f.t.twist_f, st)
                      : gl_launch_strided256<0, false>(f, d, from, f.k, s_lo, npoly, f.t.tw, nullptr, st);
        else
            rc = last ? gl_launch_strided<0, true>(f, ms[p], d, from, f.k, s_lo, npoly, f.t.tw, f.t.twist_f, st)
                      : gl_launch_strided<0, false>(f, ms[p], d, from, f.k, s_lo, npoly, f.t.tw, nullptr, st);
        if (rc) return rc;
        s_lo += ms[p];
    }
    return 0;
}
inline int gl_strided_inv(const GoldilocksFastTables &f, uint64_t *d, size_t npoly, bool fused, hipStream_t st) {
    const uint64_t *tw_i = fused ? f.t.twist_i_mul : f.t.twist_i_plain;
    if (f.cols256) return gl_launch_cols256<1>(f, d, d, npoly, f.t.wci, tw_i, st);
    int ms[8];
    const int n = gl_plan(f.k > 12 ? f.k - 12 : 0, ms);
    int s_lo = f.k > 12 ? f.k - 12 : 0;
    for (int p = n - 1; p >= 0; p--) {
        s_lo -= ms[p];
        const bool first = p == n - 1;
        int rc;
        if (ms[p] == 8)
            rc = first ? gl_launch_strided256<1, true>(f, d, d, f.k, s_lo, npoly, f.t.itw, tw_i, st)
                       : gl_launch_strided256<1, false>(f, d, d, f.k, s_lo, npoly, f.t.itw, nullptr, st);
        else
            rc = first ? gl_launch_strided<1, true>(f, ms[p], d, d, f.k, s_lo, npoly, f.t.itw, tw_i, st)
                       : gl_launch_strided<1, false>(f, ms[p], d, d, f.k, s_lo, npoly, f.t.itw, nullptr, st);
        if (rc) return rc;
    }
    return 0;
}
template <int MODE>
inline int gl_launch_rows(const GoldilocksFastTables &f, uint64_t *a, const uint64_t *b, uint64_t *out, size_t npoly,
                          bool fused, hipStream_t st) {
    const size_t n_total = npoly << f.k;
    const size_t tiles = (n_total + gl::kTile - 1) / gl::kTile;
    if (tiles == 0 || tiles > 0x7FFFFFFFull) return 1;
    GlProfScope ps(f, 1, st);
    const uint64_t *w1i = fused ? f.t.w1i_mul : f.t.w1i;
    dim3 g((unsigned)tiles), blk(256);
#define SR_GL_ROWS(QQ, TT) hipLaunchKernelGGL((gl::rows_kernel<MODE, QQ, TT>), g, blk, 0, st, a, b, out, f.t, w1i, n_total)
    const bool tw = f.c == 0;  // whole ring elements per tile: the rows kernel applies the twist itself
    switch (f.k - f.c) {       // log2 of the cyclic size
        case 8:
            if (tw) SR_GL_ROWS(4, true);
            else hipLaunchKernelGGL((gl::rows256_kernel<MODE>), g, blk, 0, st, a, b, out, f.t);
            break;
        case 9: if (tw) SR_GL_ROWS(3, true); else SR_GL_ROWS(3, false); break;
        case 10: if (tw) SR_GL_ROWS(2, true); else SR_GL_ROWS(2, false); break;
        case 11: if (tw) SR_GL_ROWS(1, true); else SR_GL_ROWS(1, false); break;
        default: if (tw) SR_GL_ROWS(0, true); else SR_GL_ROWS(0, false); break;
    }
#undef SR_GL_ROWS
    return hipGetLastError() != hipSuccess;
}
inline int gl_fast_fwd(const GoldilocksFastTables &f, uint64_t *d, size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (gl_strided_fwd(f, d, d, batch, st)) return 1;
    return gl_launch_rows<0>(f, d, nullptr, d, batch, false, st);
}
inline int gl_fast_inv(const GoldilocksFastTables &f, uint64_t *d, size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (gl_launch_rows<1>(f, d, nullptr, d, batch, false, st)) return 1;
    return gl_strided_inv(f, d, batch, false, st);
}
// out = a * b (ring product on in-memory images); a and b are only read (coeff_form.rs:250-258 never mutates an operand):
// a's column stages go straight to out, b's into scratch (scratch_polys ring elements, caller-owned; the batch is cut into
// chunks of that many elements).  out may alias a; b must not alias out.
inline int gl_fast_ring_mul(const GoldilocksFastTables &f, uint64_t *out, const uint64_t *a, const uint64_t *b, uint64_t *scratch,
                            size_t scratch_polys, size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (f.k <= 12)  // whole ring elements per tile: one fused launch, a and b only read
        return gl_launch_rows<2>(f, const_cast<uint64_t *>(a), b, out, batch, true, st);
    if (scratch_polys == 0) return 1;
    size_t chunk = f.chunk_polys ? f.chunk_polys : batch;
    if (chunk > scratch_polys) chunk = scratch_polys;
    const size_t stride = (size_t)1 << f.k;
    for (size_t e = 0; e < batch; e += chunk) {
        const size_t n = batch - e < chunk ? batch - e : chunk;
        uint64_t *o = out + e * stride;
        if (f.cols256) {  // both operands' column passes in one launch (a's output goes straight to out, b's into the scratch)
            if (gl_launch_cols256_pair_plain(f, o, a + e * stride, scratch, b + e * stride, n, st)) return 1;
        } else {
            if (gl_strided_fwd(f, o, a + e * stride, n, st)) return 1;
            if (gl_strided_fwd(f, scratch, b + e * stride, n, st)) return 1;
        }
        if (f.split_rows) {
            if (gl_launch_rows<0>(f, scratch, nullptr, scratch, n, false, st)) return 1;
            if (gl_launch_rows<3>(f, o, scratch, o, n, true, st)) return 1;
        } else if (gl_launch_rows<2>(f, o, scratch, o, n, true, st)) {
            return 1;
        }
        if (gl_strided_inv(f, o, n, true, st)) return 1;
    }
    return 0;
}

// The same product on LANES streams (cols256 plans: 2^16 <= D <= 2^20).  Chunks of `chunk` ring elements are dealt round-robin to
// the lanes; a lane runs a chunk's four launches through ITS OWN scratch pair (sa: a's column-pass output, rows output; sb: b's), so
// that (i) a chunk's intermediates -- 2 x chunk x D x 8 bytes, 128 MB at the default chunk -- are re-read from the Infinity Cache
// instead of HBM (61 against ~100 pJ per byte on a step that runs at the socket's power cap: tools/ubench/l2_power.hip), and (ii)
// the other lane's kernels fill the tail of every small launch (one stream at this chunk size: 21.5 ms per config-2 batch; two:
// 16.3-16.9 ms against 17.3-17.6 ms for the former eight large chunks on one stream).  fork / join: events owned by the caller.
struct GlLanes {
    int n = 0;                       // streams in use (<= 2)
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t fork = nullptr, join[2] = {nullptr, nullptr};
    uint64_t *sa[2] = {nullptr, nullptr}, *sb[2] = {nullptr, nullptr};
    size_t chunk = 0;                // ring elements per chunk (each of sa[i], sb[i] holds that many)
};
inline int gl_fast_ring_mul_lanes(const GoldilocksFastTables &f, uint64_t *out, const uint64_t *a, const uint64_t *b, const GlLanes &L,
                                  size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (!f.cols256 || L.n < 1 || L.chunk == 0) return 1;
    if (hipEventRecord(L.fork, st) != hipSuccess) return 1;
    for (int i = 0; i < L.n; i++)
        if (hipStreamWaitEvent(L.st[i], L.fork, 0) != hipSuccess) return 1;
    const size_t stride = (size_t)1 << f.k;
    int rc = 0;
    size_t c = 0;
    for (size_t e = 0; e < batch && !rc; e += L.chunk, c++) {
        const int i = (int)(c % (size_t)L.n);
        const size_t n = batch - e < L.chunk ? batch - e : L.chunk;
        rc = gl_launch_cols256_lane_pair(f, L.sa[i], a + e * stride, L.sb[i], b + e * stride, n, L.st[i]);
        if (f.split_rows) {
            if (!rc) rc = gl_launch_rows<0>(f, L.sb[i], nullptr, L.sb[i], n, false, L.st[i]);
            if (!rc) rc = gl_launch_rows<3>(f, L.sa[i], L.sb[i], L.sa[i], n, true, L.st[i]);
        } else if (!rc) {
            rc = gl_launch_rows<2>(f, L.sa[i], L.sb[i], L.sa[i], n, true, L.st[i]);
        }
        if (!rc) rc = gl_launch_cols256_lane<1>(f, out + e * stride, L.sa[i], n, f.t.wci, f.t.twist_i_mul, L.st[i]);
    }
    for (int i = 0; i < L.n; i++) {  // join even after a failed launch: the caller's stream must not run ahead of the lanes
        if (hipEventRecord(L.join[i], L.st[i]) != hipSuccess) rc = 1;
        if (hipStreamWaitEvent(st, L.join[i], 0) != hipSuccess) rc = 1;
    }
    return rc;
}

// the stand-alone transforms (elementwise_crt / elementwise_icrt) in the same chunks on the same lanes: in place, no scratch; the chunk
// written by the first launch is re-read by the second from the Infinity Cache
template <int DIR>
inline int gl_fast_transform_lanes(const GoldilocksFastTables &f, uint64_t *d, const GlLanes &L, size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (!f.cols256 || L.n < 1 || L.chunk == 0) return 1;
    if (hipEventRecord(L.fork, st) != hipSuccess) return 1;
    for (int i = 0; i < L.n; i++)
        if (hipStreamWaitEvent(L.st[i], L.fork, 0) != hipSuccess) return 1;
    const size_t stride = (size_t)1 << f.k;
    int rc = 0;
    size_t c = 0;
    for (size_t e = 0; e < batch && !rc; e += L.chunk, c++) {
        const int i = (int)(c % (size_t)L.n);
        const size_t n = batch - e < L.chunk ? batch - e : L.chunk;
        uint64_t *dc = d + e * stride;
        if (DIR == 0) {
            rc = gl_launch_cols256_lane<0>(f, dc, dc, n, f.t.wcf, f.t.twist_f, L.st[i]);
            if (!rc) rc = gl_launch_rows<0>(f, dc, nullptr, dc, n, false, L.st[i]);
        } else {
            rc = gl_launch_rows<1>(f, dc, nullptr, dc, n, false, L.st[i]);
            // the plain inverse pass here: behind the light rows256_kernel<1> of a stand-alone icrt the workgroup-owns-its-columns
            // kernel lost 2.4 % (6.06-6.13 against 5.94-5.95 ms per config-2 batch); the forward one above gains 1.2 %
            if (!rc) rc = gl_launch_cols256<1>(f, dc, dc, n, f.t.wci, f.t.twist_i_plain, L.st[i]);
        }
    }
    for (int i = 0; i < L.n; i++) {
        if (hipEventRecord(L.join[i], L.st[i]) != hipSuccess) rc = 1;
        if (hipStreamWaitEvent(st, L.join[i], 0) != hipSuccess) rc = 1;
    }
    return rc;
}

// the constant-operand product (b already in NTT form) on the same lanes: three launches per chunk, one scratch buffer per lane
inline int gl_fast_ring_mul_rhs_lanes(const GoldilocksFastTables &f, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, const GlLanes &L,
                                      size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (!f.cols256 || L.n < 1 || L.chunk == 0) return 1;
    if (hipEventRecord(L.fork, st) != hipSuccess) return 1;
    for (int i = 0; i < L.n; i++)
        if (hipStreamWaitEvent(L.st[i], L.fork, 0) != hipSuccess) return 1;
    const size_t stride = (size_t)1 << f.k;
    int rc = 0;
    size_t c = 0;
    for (size_t e = 0; e < batch && !rc; e += L.chunk, c++) {
        const int i = (int)(c % (size_t)L.n);
        const size_t n = batch - e < L.chunk ? batch - e : L.chunk;
        rc = gl_launch_cols256_lane<0>(f, L.sa[i], a + e * stride, n, f.t.wcf, f.t.twist_f, L.st[i]);
        if (!rc) rc = gl_launch_rows<3>(f, L.sa[i], b_ntt + e * stride, L.sa[i], n, true, L.st[i]);
        if (!rc) rc = gl_launch_cols256_lane<1>(f, out + e * stride, L.sa[i], n, f.t.wci, f.t.twist_i_mul, L.st[i]);
    }
    for (int i = 0; i < L.n; i++) {
        if (hipEventRecord(L.join[i], L.st[i]) != hipSuccess) rc = 1;
        if (hipStreamWaitEvent(st, L.join[i], 0) != hipSuccess) rc = 1;
    }
    return rc;
}

// out = icrt(crt(a) (.) b_ntt), b_ntt = crt(b) as sr_ntt_fwd leaves it: a's column stages go straight to out, the rows kernel
// transforms a's tile only and reads b's slots in NTT order, the inverse column stages finish in place.  No scratch.
inline int gl_fast_ring_mul_rhs(const GoldilocksFastTables &f, uint64_t *out, const uint64_t *a, const uint64_t *b_ntt, size_t batch,
                                hipStream_t st) {
    if (batch == 0) return 0;
    if (f.k <= 12) return gl_launch_rows<3>(f, const_cast<uint64_t *>(a), b_ntt, out, batch, true, st);
    if (gl_strided_fwd(f, out, a, batch, st)) return 1;
    if (gl_launch_rows<3>(f, out, b_ntt, out, batch, true, st)) return 1;
    return gl_strided_inv(f, out, batch, true, st);
}

}  // namespace sr
