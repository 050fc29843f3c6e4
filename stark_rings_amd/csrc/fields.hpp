// Field arithmetic for the three base primes of the hot path, written for gfx950 VALU
// (32-bit lanes, v_mad_u64_u32 as the only wide multiply) and usable on the host for
// context set-up (twiddle roots, scale constants).
//
// Reference semantics restated: ark-ff 0.4.2 Fp<MontBackend<C,N>,N> (Cargo.lock:59-62), whose
// in-memory image of a is a * 2^(64N) mod p, canonical.  Configured at
//   crates/ring/src/cyclotomic_ring/models/goldilocks/mod.rs:20-24   (Fp64, p = 2^64-2^32+1, g = 7)
//   crates/ring/src/cyclotomic_ring/models/babybear/mod.rs:21-25     (Fp64!, p = 15*2^27+1, g = 31)
//   crates/ring/src/cyclotomic_ring/models/stark_prime/mod.rs:20-24  (Fp256, p = 2^251+17*2^192+1, g = 3)
//
// Device-side convention ("table form"): a transform is linear, so boundary words are treated
// as plain residues and multiplied by twiddles kept in whatever form makes the product cheapest:
//   mul_tw(x, w_tab) = x * w  where  w_tab = w * kappa,  kappa = 1 (Goldilocks, direct 2^64 = 2^32-1
//   folding), 2^32 (BabyBear, 32-bit Montgomery), 2^256 (Stark, 8x32-bit CIOS Montgomery).
// mul_boundary(a, b) = a * b * R_b^-1 with R_b = 2^64 / 2^64 / 2^256 is the reference's Fp product on
// the in-memory images (ntt_form.rs:177-189).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SR_HD __host__ __device__ __forceinline__

// Optimisation barrier for carry chains.  ROCm 7.2's AMDGPU DAG combiner folds
//   addcarry(subcarry(x, b, c0).value, 0, c1)  ->  addcarry(x, -b, c1)
// keeping the VALUE right but not the CARRY-OUT, which breaks `carry of (x - b') + k` tests (seen as results off
// by 2^64 mod p in the Fq3 slot products).  Passing the intermediate limb through an empty asm keeps the two
// carry operations separate; it emits no instruction.
#if defined(__HIP_DEVICE_COMPILE__)
#define SR_OPAQUE(x) asm("" : "+v"(x))  /* not volatile: free to be scheduled, only opaque to the combiner */
#else
#define SR_OPAQUE(x) ((void)0)
#endif

namespace sr {

// Workgroup -> tile order of the column passes.  The hardware hands consecutive workgroups to the eight XCDs in turn, so with
// tile = blockIdx the neighbouring 128-byte column chunks of one leg are fetched through eight different L2s at unrelated moments
// and DRAM sees its pages opened once per chunk.  xcd_tile() gives runs of 2^cl consecutive tiles (column chunks of one ring
// element) to ONE XCD, back to back (run 8 g + x goes to XCD x): the 16-column pattern of D = 2^16 then streams at 6.0 instead
// of 5.4 TB/s (tools/ubench/blocked_pattern.hip).  `grouped` = the launch's tile count rounded down to a multiple of 8 * 2^cl
// (xcd_grouped_tiles); tiles at or beyond it keep tile = blockIdx.
__device__ __forceinline__ unsigned xcd_tile(unsigned bid, int cl, unsigned grouped) {
    if (bid < grouped) {
        const unsigned xcd = bid & 7u, slot = bid >> 3;
        return ((((slot >> cl) << 3) + xcd) << cl) | (slot & ((1u << cl) - 1u));
    }
    return bid;
}
inline unsigned xcd_grouped_tiles(size_t tiles, int cl) { return (unsigned)(tiles & ~(((size_t)8 << cl) - 1)); }

// ------------------------------------------------------------------------------------------
// Representative checks of the tuned Goldilocks path.  The lazy butterflies (Goldilocks::addsub_lazy, ntt_goldilocks.hpp) leave
// 64-bit representatives that differ from the canonical value only when a sum lands in [p, 2^64): 2^-32 per value on uniform
// data, which no uniform-data parity test can see.  The product library compiles these hooks to nothing.  A second build with
// -DSR_GL_CHECK_REPS (libstarkrings_hip_check.so; __graft_entry__.build) makes every kernel count, while the real kernels run:
//   [0] a canonical routine (add, sub, addsub, a phased canonical butterfly) or the canonical leg t of a lazy butterfly that was
//       handed a representative >= p;
//   [1] a lazy sum whose + eps wrapped a second time;   [2] a lazy difference whose + p borrowed a second time;
//   [3] a word >= p leaving the library;                 [4] a routine documented "canonical out" that returned >= p.
// The lazy nine-limb Stark arithmetic (stark_lazy.hpp) has invariants of the same kind -- bounds no parity test can see until they
// break -- and the same build counts them:
//   [5] a limb-wise add / sub whose exact sum left the int32 range the invariants promise (|l| < 2^31 - 16);
//   [6] a Montgomery product whose operands could overflow the 64-bit column accumulator (9 max|a.l| max|w.l| + 2^58 >= 2^63);
//   [7] is not a violation count but the high-water mark of |limb| over every add / sub since the last reset (the tests report the
//       headroom the kernels' reduction schedule leaves below 2^31).
// sr_selftest_rep_counters reads and clears them; tests/test_rep_invariants.py asserts zeros over crafted, structured, edge and
// uniform operands of every tuned plan.
// ------------------------------------------------------------------------------------------
namespace repcheck {
constexpr unsigned long long kP = 0xFFFFFFFF00000001ull;
#if defined(SR_GL_CHECK_REPS)
__device__ unsigned long long g_counters[8];
#endif
#if defined(SR_GL_CHECK_REPS) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void count(int i) { atomicAdd(&g_counters[i], 1ull); }
__device__ __forceinline__ void canonical_in(unsigned long long a, unsigned long long b) {
    if (a >= kP || b >= kP) count(0);
}
__device__ __forceinline__ void lazy_fix(unsigned long long s, unsigned long long d, unsigned long long c1, unsigned long long c2) {
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (((c1 >> lane) & 1ull) && s > ~0xFFFFFFFFull) count(1);   // s + eps would wrap again
    if (((c2 >> lane) & 1ull) && d < 0xFFFFFFFFull) count(2);    // d + p = d - eps would borrow again
}
__device__ __forceinline__ void leaves_library(unsigned long long v) {
    if (v >= kP) count(3);
}
__device__ __forceinline__ void canonical_out(unsigned long long v) {
    if (v >= kP) count(4);
}
__device__ __forceinline__ void stark_limb(long long exact) {
    const unsigned long long m = (unsigned long long)(exact < 0 ? -exact : exact);
    if (m > 2147483647ull - 16) count(5);
    if (m > *(volatile unsigned long long *)&g_counters[7]) atomicMax(&g_counters[7], m);   // high-water mark
}
__device__ __forceinline__ void stark_columns(const int *a, const int *w) {
    unsigned long long ma = 0, mw = 0;
    for (int i = 0; i < 9; i++) {
        const unsigned long long x = (unsigned long long)(a[i] < 0 ? -(long long)a[i] : (long long)a[i]);
        const unsigned long long y = (unsigned long long)(w[i] < 0 ? -(long long)w[i] : (long long)w[i]);
        ma = x > ma ? x : ma;
        mw = y > mw ? y : mw;
    }
    if (ma * mw >= ((1ull << 63) - (1ull << 58)) / 9) count(6);
}
#else
SR_HD void canonical_in(unsigned long long, unsigned long long) {}
SR_HD void lazy_fix(unsigned long long, unsigned long long, unsigned long long, unsigned long long) {}
SR_HD void leaves_library(unsigned long long) {}
SR_HD void canonical_out(unsigned long long) {}
SR_HD void stark_limb(long long) {}
SR_HD void stark_columns(const int *, const int *) {}
#endif
}  // namespace repcheck

// ------------------------------------------------------------------------------------------
// Goldilocks  p = 2^64 - 2^32 + 1
// ------------------------------------------------------------------------------------------
struct Goldilocks {
    using elem = uint64_t;      // register / LDS form, canonical in [0, p)
    using storage = uint64_t;   // global-memory form (reference layout: one u64 limb)
    static constexpr int kStorageWords64 = 1;
    static constexpr int kLdsWords = 2;  // 32-bit words per element in LDS
    static constexpr uint64_t P = 0xFFFFFFFF00000001ull;
    static constexpr uint64_t EPS = 0xFFFFFFFFull;  // 2^64 mod p
    static constexpr uint32_t kGenerator = 7;
    static constexpr int kBoundaryBits = 64;  // R_b = 2^64
    static constexpr int kTwoAdicity = 32;

    SR_HD static elem zero() { return 0; }
    SR_HD static elem load(const storage *p) { return *p; }
    SR_HD static void store(storage *p, elem v) { *p = v; }
    SR_HD static bool valid(elem v) { return v < P; }

    SR_HD static elem neg(elem a) { return a ? P - a : 0; }
    // any 64-bit representative -> the canonical one (representatives >= p exist only for values below 2^32 - 1)
    SR_HD static elem canon(elem a) { return a >= P ? a - P : a; }

    // The arithmetic below exists twice with identical signatures and identical values: hand-scheduled gfx950 code for the device
    // pass, plain C++ for the host pass (context set-up, sr_selftest_field_op, tools/ubench/field_check.hip, which compares the two).
    // On gfx950 nearly every integer VALU instruction this needs (carries, compares, v_mad_u64_u32, 64-bit adds) issues at the same
    // rate (profiles/r01/valu_issue_rates_gfx950.txt), so the device forms are written for the fewest instructions AND the fewest
    // non-VALU issue slots around them:
    //   * a conditional +-p is ONE v_lshl_add_u64 executed under an EXEC mask made from the carry (s_and / s_andn1_saveexec ...
    //     s_mov exec: SALU work, issued beside the other waves' VALU) instead of a second carry chain plus two v_cndmask;
    //   * s_nop: the two wait states gfx950 wants between a VALU write of an SGPR carry and the VALU read of it -- the compiler
    //     cannot see into an asm statement, so every statement pads its own hazards; a statement that writes SCC (s_or_b64,
    //     s_and*_saveexec) declares the clobber (tests/test_inline_asm_clobbers.py);
    //   * multi-step operations are single statements where the compiler would otherwise put a wait state (s_nop 0) between a
    //     statement that defines a VGPR and the next instruction that reads it.
    // SR_GL_CHECK_REPS (the invariant-checking build, tests/test_rep_invariants.py) counts violated preconditions in repcheck::.
#if defined(__HIP_DEVICE_COMPILE__)
    // a + b for canonical a, b: 4 VALU.  t = a + eps cannot overflow; the carry of t + b says a + b >= p (then t + b - 2^64 = a + b - p
    // is the answer) and the other lanes take the eps back (+ p mod 2^64).
    SR_HD static elem add(elem a, elem b) {
        repcheck::canonical_in(a, b);
        uint64_t t, c, sv;
        uint32_t r0, r1;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(a), "s"((uint64_t)EPS));
        asm("v_add_co_u32_e64 %0, %2, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r0), "=&v"(r1), "=&s"(c)
            : "v"((uint32_t)t), "v"((uint32_t)(t >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)));
        uint64_t r = (uint64_t)r0 | ((uint64_t)r1 << 32);
        asm("s_andn1_saveexec_b64 %1, %2\n\t"
            "v_lshl_add_u64 %0, %0, 0, %3\n\t"
            "s_mov_b64 exec, %1"
            : "+v"(r), "=&s"(sv)
            : "s"(c), "s"((uint64_t)P)
            : "scc");
        return r;
    }
    // a - b (+ p on borrow): 3 VALU.  Exact for canonical a, b.
    SR_HD static elem sub(elem a, elem b) {
        repcheck::canonical_in(a, b);
        uint64_t c, sv;
        uint32_t r0, r1;
        asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32_e64 %1, %2, %4, %6, %2"
            : "=&v"(r0), "=&v"(r1), "=&s"(c)
            : "v"((uint32_t)a), "v"((uint32_t)(a >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)));
        uint64_t r = (uint64_t)r0 | ((uint64_t)r1 << 32);
        asm("s_and_saveexec_b64 %1, %2\n\t"
            "v_lshl_add_u64 %0, %0, 0, %3\n\t"
            "s_mov_b64 exec, %1"
            : "+v"(r), "=&s"(sv)
            : "s"(c), "s"((uint64_t)P)
            : "scc");
        return r;
    }
    // s = a + b, d = a - b in one go (every butterfly needs both): the seven VALU of add + sub, but as one statement the two carry
    // chains fill each other's wait states and the EXEC mask is saved and restored once.
    SR_HD static void addsub(elem a, elem b, elem &s, elem &d) {
        repcheck::canonical_in(a, b);
        uint64_t t, c1, c2, sv;
        uint32_t s0, s1, d0, d1;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(a), "s"((uint64_t)EPS));
        asm("v_add_co_u32_e64 %0, %4, %6, %8\n\t"
            "v_sub_co_u32_e64 %2, %5, %10, %8\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %1, %4, %7, %9, %4\n\t"
            "v_subb_co_u32_e64 %3, %5, %11, %9, %5"
            : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1), "=&s"(c1), "=&s"(c2)
            : "v"((uint32_t)t), "v"((uint32_t)(t >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)), "v"((uint32_t)a), "v"((uint32_t)(a >> 32)));
        s = (uint64_t)s0 | ((uint64_t)s1 << 32);
        d = (uint64_t)d0 | ((uint64_t)d1 << 32);
        asm("s_andn1_saveexec_b64 %2, %3\n\t"      // lanes WITHOUT the carry of t + b take the eps back
            "v_lshl_add_u64 %0, %0, 0, %5\n\t"
            "s_and_b64 exec, %2, %4\n\t"           // lanes that borrowed in a - b take + p
            "v_lshl_add_u64 %1, %1, 0, %5\n\t"
            "s_mov_b64 exec, %2"
            : "+v"(s), "+v"(d), "=&s"(sv)
            : "s"(c1), "s"(c2), "s"((uint64_t)P)
            : "scc");
    }
    // The same in three separate steps, for callers that run a whole stage's butterflies phase by phase (ntt_goldilocks.hpp:
    // dif_phased / dit_phased): every step's results are consumed eight statements later, so no post-asm wait state is needed
    // anywhere.  plus_eps: the t = a + eps of the canonical sum.  addsub_chains: the carry chains of s = t + b and d = m - n with
    // (m, n) = (a, b), or (b, a) under SWAP (a twiddle 2^E with E >= 96 is -2^(E-96)).  addsub_fix: the two masked corrections.
    SR_HD static elem plus_eps(elem a) {
        uint64_t t;
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(a), "s"((uint64_t)EPS));
        return t;
    }
    template <bool SWAP>
    SR_HD static void addsub_chains(elem t, elem a, elem b, uint32_t &s0, uint32_t &s1, uint32_t &d0, uint32_t &d1, uint64_t &c1, uint64_t &c2) {
        const elem m = SWAP ? b : a, n = SWAP ? a : b;   // d = m - n
        asm("v_add_co_u32_e64 %0, %4, %6, %8\n\t"
            "v_sub_co_u32_e64 %2, %5, %10, %12\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32_e64 %1, %4, %7, %9, %4\n\t"
            "v_subb_co_u32_e64 %3, %5, %11, %13, %5"
            : "=&v"(s0), "=&v"(s1), "=&v"(d0), "=&v"(d1), "=&s"(c1), "=&s"(c2)
            : "v"((uint32_t)t), "v"((uint32_t)(t >> 32)), "v"((uint32_t)b), "v"((uint32_t)(b >> 32)), "v"((uint32_t)m), "v"((uint32_t)(m >> 32)),
              "v"((uint32_t)n), "v"((uint32_t)(n >> 32)));
    }
    SR_HD static void addsub_fix(elem &s, elem &d, uint64_t c1, uint64_t c2) {
        uint64_t sv;
        asm("s_andn1_saveexec_b64 %2, %3\n\t"
            "v_lshl_add_u64 %0, %0, 0, %5\n\t"
            "s_and_b64 exec, %2, %4\n\t"
            "v_lshl_add_u64 %1, %1, 0, %5\n\t"
            "s_mov_b64 exec, %2"
            : "+v"(s), "+v"(d), "=&s"(sv)
            : "s"(c1), "s"(c2), "s"((uint64_t)P)
            : "scc");
    }
    // LAZY legs of a decimation-in-time butterfly: s = a + t, d = a - t as 64-bit representatives for ANY u64 a and a CANONICAL t
    // (the shift product v 2^E it comes from is).  a + t wraps at most once -- the + eps on carry lands below 2^64 because
    // a + t - 2^64 <= p - 2 -- and a - t borrows at most once (t - a <= p): six VALU instead of seven, no a + eps.  The results are
    // arbitrary representatives; their consumers (another lazy butterfly's a or v, mul, mul_pow2) take those; a canonical butterfly
    // does NOT (ntt_goldilocks.hpp says where each kind sits; tools/model_fast_goldilocks.py: lazy_bf).
    SR_HD static void addsub_lazy_fix(elem &s, elem &d, uint64_t c1, uint64_t c2) {
        repcheck::lazy_fix(s, d, c1, c2);
        uint64_t sv;
        asm("s_and_saveexec_b64 %2, %3\n\t"       // lanes whose a + t carried take + eps (2^64 = eps)
            "v_lshl_add_u64 %0, %0, 0, %6\n\t"
            "s_and_b64 exec, %2, %4\n\t"           // lanes that borrowed in a - t take + p (= - eps mod 2^64)
            "v_lshl_add_u64 %1, %1, 0, %5\n\t"
            "s_mov_b64 exec, %2"
            : "+v"(s), "+v"(d), "=&s"(sv)
            : "s"(c1), "s"(c2), "s"((uint64_t)P), "s"((uint64_t)EPS)
            : "scc");
    }
    // canonical (l2 + hl * eps) for l2 + hl * eps < 2^64 + p: 3 VALU.  v_mad_u64_u32 delivers the 65th bit as its carry-out (C++
    // cannot name it); that bit or t >= p (one 64-bit compare against p in an SGPR pair) selects the lanes that take + eps = - p.
    SR_HD static elem mad_eps_fix(uint64_t l2, uint32_t hl) {
        uint64_t t, cy, c2;
        asm("v_mad_u64_u32 %0, %1, %3, -1, %4\n\t"
            "v_cmp_le_u64_e64 %2, %5, %0\n\t"
            "s_or_b64 %2, %2, %1\n\t"
            "s_and_saveexec_b64 %1, %2\n\t"
            "v_lshl_add_u64 %0, %0, 0, %6\n\t"
            "s_mov_b64 exec, %1"
            : "=&v"(t), "=&s"(cy), "=&s"(c2)
            : "v"(hl), "v"(l2), "s"((uint64_t)P), "s"((uint64_t)EPS)
            : "scc");
        return t;
    }
    // the "+ p on the lanes that borrowed" of a subtraction and the fold behind it as ONE statement: canonical ((r + [bo] p) + hl eps)
    SR_HD static elem fix_fold(uint64_t r, uint64_t bo, uint32_t hl) {
        uint64_t t, cy, c2;
        asm("s_and_saveexec_b64 %1, %4\n\t"
            "v_lshl_add_u64 %3, %3, 0, %6\n\t"
            "s_mov_b64 exec, %1\n\t"
            "v_mad_u64_u32 %0, %1, %5, -1, %3\n\t"
            "v_cmp_le_u64_e64 %2, %6, %0\n\t"
            "s_or_b64 %2, %2, %1\n\t"
            "s_and_saveexec_b64 %1, %2\n\t"
            "v_lshl_add_u64 %0, %0, 0, %7\n\t"
            "s_mov_b64 exec, %1"
            : "=&v"(t), "=&s"(cy), "=&s"(c2), "+v"(r)
            : "s"(bo), "v"(hl), "s"((uint64_t)P), "s"((uint64_t)EPS)
            : "scc");
        return t;
    }
    // (hi * 2^64 + lo) mod p for ANY lo, hi, using 2^64 = eps and 2^96 = -1:  lo - hh + hl * eps
    SR_HD static elem reduce128(uint64_t lo, uint64_t hi) {
        uint64_t c;
        uint32_t r0, r1;
        asm("v_sub_co_u32_e64 %0, %2, %3, %5\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32_e64 %1, %2, %4, 0, %2"
            : "=&v"(r0), "=&v"(r1), "=&s"(c)
            : "v"((uint32_t)lo), "v"((uint32_t)(lo >> 32)), "v"((uint32_t)(hi >> 32)));
        return fix_fold((uint64_t)r0 | ((uint64_t)r1 << 32), c, (uint32_t)hi);
    }
    // a * b mod p for ANY 64-bit representatives a, b; canonical out.  13 VALU.  The compiler's expansion of the 64 x 64 -> 128-bit
    // product zero-extends four 32-bit halves into 64-bit addends of v_mad_u64_u32 (gfx950 wants 64-bit register pairs
    // even-aligned, so {hi(x), 0} always costs a v_mov_b32) and adds the two middle carries with a v_lshl_add_u64: 16 VALU with the
    // reduction.  Here the middle product takes its FULL 64-bit addend, al*bh + (ah*bl + hi(al*bl)) = P2 + c 2^64, and the carry-out
    // c (weight 2^96 = -1 mod p) goes straight into the borrow-in of the reduction's subtraction:
    //   a b = LO + P3 2^64 + c 2^96,  LO = {lo(al bl), lo(P2)},  P3 = ah bh + hi(P2)   ==>   a b = LO - hi(P3) - c + lo(P3) eps  (mod p)
    SR_HD static elem mul(elem a, elem b) {
        const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
        const uint64_t p0 = (uint64_t)al * bl;
        const uint64_t p1 = (uint64_t)ah * bl + (p0 >> 32);  // <= 2^64 - 2^32
        uint64_t p2, c, bo;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(p2), "=s"(c) : "v"(al), "v"(bh), "v"(p1));
        const uint64_t p3 = (uint64_t)ah * bh + (p2 >> 32);  // <= 2^64 - 2^32
        uint32_t r0, r1;
        // gfx950 wants 2 wait states between the VALU write of the SGPR pair c and its use as a borrow-in.  The mad that makes p3
        // (an input of this statement, and itself dependent on p2) is always in between: one of the two; the s_nop 0 is the other,
        // so the distance no longer rests on what the scheduler happens to put there (tests/test_inline_asm_clobbers.py checks it).
        asm("s_nop 0\n\t"
            "v_subb_co_u32_e64 %0, %2, %3, %5, %6\n\t"
            "s_nop 1\n\t"
            "v_subb_co_u32_e64 %1, %2, %4, 0, %2"
            : "=&v"(r0), "=&v"(r1), "=&s"(bo)
            : "v"((uint32_t)p0), "v"((uint32_t)p2), "v"((uint32_t)(p3 >> 32)), "s"(c));
        return fix_fold((uint64_t)r0 | ((uint64_t)r1 << 32), bo, (uint32_t)p3);
    }
#else
    SR_HD static elem add(elem a, elem b) {
        const uint64_t s = a + b;
        return ((s < a) | (s >= P)) ? s + EPS : s;
    }
    SR_HD static elem sub(elem a, elem b) {
        const uint64_t d = a - b;
        return a < b ? d - EPS : d;
    }
    SR_HD static void addsub(elem a, elem b, elem &s, elem &d) {
        s = add(a, b);
        d = sub(a, b);
    }
    SR_HD static elem plus_eps(elem a) { return a + EPS; }
    template <bool SWAP>
    SR_HD static void addsub_chains(elem t, elem a, elem b, uint32_t &s0, uint32_t &s1, uint32_t &d0, uint32_t &d1, uint64_t &c1, uint64_t &c2) {
        const elem m = SWAP ? b : a, n = SWAP ? a : b;
        const uint64_t s = t + b, d = m - n;
        c1 = s < t;   // carry of t + b
        c2 = m < n;   // borrow of m - n
        s0 = (uint32_t)s, s1 = (uint32_t)(s >> 32), d0 = (uint32_t)d, d1 = (uint32_t)(d >> 32);
    }
    SR_HD static void addsub_fix(elem &s, elem &d, uint64_t c1, uint64_t c2) {
        if (!c1) s += P;   // takes the eps of plus_eps back
        if (c2) d += P;
    }
    SR_HD static void addsub_lazy_fix(elem &s, elem &d, uint64_t c1, uint64_t c2) {
        if (c1) s += EPS;
        if (c2) d += P;
    }
    SR_HD static elem mad_eps_fix(uint64_t l2, uint32_t hl) {
        const unsigned __int128 w = (unsigned __int128)l2 + (uint64_t)hl * EPS;
        uint64_t t = (uint64_t)w;
        if ((uint64_t)(w >> 64) != 0 || t >= P) t += EPS;   // on overflow t <= 2^64 - 2^33 and t + eps < p
        return t;
    }
    SR_HD static elem fix_fold(uint64_t r, uint64_t bo, uint32_t hl) { return mad_eps_fix(bo ? r + P : r, hl); }
    SR_HD static elem reduce128(uint64_t lo, uint64_t hi) {
        const uint32_t hh = (uint32_t)(hi >> 32);
        return fix_fold(lo - hh, lo < hh, (uint32_t)hi);   // lo - hh + [borrow] p never overflows: hh < 2^32
    }
    SR_HD static elem mul(elem a, elem b) {
        const unsigned __int128 x = (unsigned __int128)a * b;
        return reduce128((uint64_t)x, (uint64_t)(x >> 64));
    }
#endif
    SR_HD static void addsub_lazy(elem a, elem t, elem &s, elem &d) {
        uint64_t c1, c2;
        uint32_t s0, s1, d0, d1;
        addsub_chains<false>(a, a, t, s0, s1, d0, d1, c1, c2);
        s = (uint64_t)s0 | ((uint64_t)s1 << 32);
        d = (uint64_t)d0 | ((uint64_t)d1 << 32);
        addsub_lazy_fix(s, d, c1, c2);
    }
    SR_HD static elem mul_tw(elem a, elem w) { return mul(a, w); }
    // a * b * 2^-64: 2^-64 = -2^32 (mod p), so (hi, lo) -> hi - lo * 2^32
    SR_HD static elem mul_boundary(elem a, elem b) {
        unsigned __int128 x = (unsigned __int128)a * b;
        uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);  // hi < p because a, b < p
        uint64_t m = reduce128(lo << 32, lo >> 32);
        return sub(hi, m);
    }
    SR_HD static elem tw_from_u64(uint64_t x) { return x % P; }
    SR_HD static elem tw_one() { return 1; }
    // sum-of-products helper: sum_i pre(a_i, b_i), then post() once == sum_i mul_boundary(a_i, b_i)
    SR_HD static elem mul_boundary_pre(elem a, elem b) { return mul_boundary(a, b); }
    SR_HD static elem boundary_post(elem x) { return x; }

    // LDS accessors: array of u64
    SR_HD static elem lds_get(const uint32_t *lds, int idx, int) {
        return *reinterpret_cast<const uint64_t *>(lds + 2 * idx);
    }
    SR_HD static void lds_put(uint32_t *lds, int idx, int, elem v) {
        *reinterpret_cast<uint64_t *>(lds + 2 * idx) = v;
    }
};

// ------------------------------------------------------------------------------------------
// BabyBear  p = 15 * 2^27 + 1 (31 bits).  The reference stores it in a u64 limb (R_b = 2^64);
// on device we keep 32-bit residues and use 32-bit Montgomery (kappa = 2^32) for twiddle products.
// ------------------------------------------------------------------------------------------
struct BabyBear {
    using elem = uint32_t;
    using storage = uint64_t;  // reference layout: Fp64, upper word always zero
    static constexpr int kStorageWords64 = 1;
    static constexpr int kLdsWords = 1;
    static constexpr uint32_t P = 2013265921u;       // 0x78000001
    static constexpr uint32_t PINV = 2013265919u;    // -p^-1 mod 2^32  (p * 0x88000001 = 1 mod 2^32)
    static constexpr uint32_t R2 = 1172168163u;      // 2^64 mod p
    static constexpr uint32_t kGenerator = 31;
    static constexpr int kBoundaryBits = 64;
    static constexpr int kTwoAdicity = 27;

    SR_HD static elem zero() { return 0; }
    SR_HD static elem load(const storage *p) { return (uint32_t)*p; }
    SR_HD static void store(storage *p, elem v) { *p = (uint64_t)v; }
    SR_HD static bool valid(elem v) { return v < P; }

    // conditional corrections as unsigned min(x, x -+ p): one plain add/sub plus one v_min_u32 (p < 2^31, so the
    // wrapped alternative is always the larger of the two when no correction is due)
    SR_HD static uint32_t umin(uint32_t x, uint32_t y) { return x < y ? x : y; }
    SR_HD static elem add(elem a, elem b) {
        uint32_t s = a + b;  // < 2^32
        return umin(s, s - P);
    }
    SR_HD static elem sub(elem a, elem b) {
        uint32_t d = a - b;
        return umin(d, d + P);
    }
    SR_HD static elem neg(elem a) { return a ? P - a : 0; }
    // a * b * 2^-32 mod p
    SR_HD static elem mont32(elem a, elem b) {
        uint64_t t = (uint64_t)a * b;
        uint32_t m = (uint32_t)t * PINV;
        uint32_t u = (uint32_t)((t + (uint64_t)m * P) >> 32);  // < 2p
        return umin(u, u - P);
    }
    SR_HD static elem mul_tw(elem a, elem w) { return mont32(a, w); }
    SR_HD static elem mul_boundary(elem a, elem b) { return mont32(mont32(a, b), 1u); }
    SR_HD static elem tw_from_u64(uint64_t x) { return mont32((uint32_t)(x % P), R2); }
    SR_HD static elem tw_one() { return tw_from_u64(1); }
    SR_HD static elem mul_boundary_pre(elem a, elem b) { return mont32(a, b); }
    SR_HD static elem boundary_post(elem x) { return mont32(x, 1u); }

    SR_HD static elem lds_get(const uint32_t *lds, int idx, int) { return lds[idx]; }
    SR_HD static void lds_put(uint32_t *lds, int idx, int, elem v) { lds[idx] = v; }
};

// ------------------------------------------------------------------------------------------
// Starknet prime  p = 2^251 + 17 * 2^192 + 1, eight 32-bit limbs, CIOS Montgomery R = 2^256.
// p = 1 (mod 2^32) so -p^-1 mod 2^32 = 0xFFFFFFFF and the reduction word is m = -t0.
// ------------------------------------------------------------------------------------------
struct U256 {
    uint32_t l[8];
};
struct alignas(16) U256Storage {
    uint64_t q[4];  // reference layout: 4 little-endian u64 limbs
};

struct Stark {
    using elem = U256;
    using storage = U256Storage;
    static constexpr int kStorageWords64 = 4;
    static constexpr int kLdsWords = 8;
    static constexpr uint32_t kGenerator = 3;
    static constexpr int kBoundaryBits = 256;
    static constexpr int kTwoAdicity = 192;

    SR_HD static uint32_t pl(int i) {  // modulus limbs
        return i == 0 ? 1u : (i == 6 ? 0x11u : (i == 7 ? 0x08000000u : 0u));
    }
    SR_HD static elem zero() {
        elem z;
#pragma unroll
        for (int i = 0; i < 8; i++) z.l[i] = 0;
        return z;
    }
    SR_HD static elem load(const storage *p) {
        elem e;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t q = p->q[i];
            e.l[2 * i] = (uint32_t)q;
            e.l[2 * i + 1] = (uint32_t)(q >> 32);
        }
        return e;
    }
    SR_HD static void store(storage *p, const elem &e) {
#pragma unroll
        for (int i = 0; i < 4; i++) p->q[i] = (uint64_t)e.l[2 * i] | ((uint64_t)e.l[2 * i + 1] << 32);
    }
    SR_HD static bool geq_p(const elem &a) {
        // a >= p ?  compare from the top limb
        bool gt = false, lt = false;
#pragma unroll
        for (int i = 7; i >= 0; i--) {
            uint32_t pi = pl(i);
            bool undecided = !(gt | lt);
            gt |= undecided & (a.l[i] > pi);
            lt |= undecided & (a.l[i] < pi);
        }
        return !lt;
    }
    SR_HD static bool valid(const elem &a) { return !geq_p(a); }
    SR_HD static uint32_t add_raw(elem &r, const elem &a, const elem &b) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)a.l[i] + b.l[i];
            r.l[i] = (uint32_t)c;
            c >>= 32;
        }
        return (uint32_t)c;
    }
    SR_HD static uint32_t sub_raw(elem &r, const elem &a, const elem &b) {
        uint32_t borrow = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t d = (uint64_t)a.l[i] - b.l[i] - borrow;
            r.l[i] = (uint32_t)d;
            borrow = (uint32_t)(d >> 32) & 1u;
        }
        return borrow;
    }
    SR_HD static elem modulus() {
        elem p;
#pragma unroll
        for (int i = 0; i < 8; i++) p.l[i] = pl(i);
        return p;
    }
    SR_HD static elem add(const elem &a, const elem &b) {
        elem s, t;
        add_raw(s, a, b);  // a + b < 2p < 2^253: no carry out
        uint32_t borrow = sub_raw(t, s, modulus());
        return borrow ? s : t;
    }
    SR_HD static elem sub(const elem &a, const elem &b) {
        elem d, t;
        uint32_t borrow = sub_raw(d, a, b);
        add_raw(t, d, modulus());
        return borrow ? t : d;
    }
    SR_HD static elem neg(const elem &a) { return sub(zero(), a); }
    // a * b * 2^-256 mod p.  Host build: textbook CIOS.  Device build: product scanning (FIPS) with a 96-bit column
    // accumulator (acc: one aligned VGPR pair, t2: the overflow word); every partial product is one v_mad_u64_u32 whose
    // carry-out feeds one v_addc_co_u32.  The carry-out is not expressible in C++, so the multiply-accumulate pair is
    // a small inline-asm statement (two MACs per statement so each carry sits its 2 wait states before its reader);
    // everything else stays C++.  The modulus is sparse (limbs 1, 0, 0, 0, 0, 0, 0x11, 2^27) and -p^-1 = -1 mod 2^32:
    // m_k = -acc_lo, and only m_i * 0x11 and m_i * 2^27 products exist.  267 VALU instructions vs 707 for the
    // compiler's CIOS (tools/ubench/stark_montmul_check.hip checks this build against the host build).
#if defined(__HIP_DEVICE_COMPILE__)
    static __device__ __forceinline__ void mac2(uint64_t &acc, uint32_t &t2, uint32_t x0, uint32_t y0, uint32_t x1,
                                                uint32_t y1) {
        uint64_t c1;
        asm("v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t"
            "v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc\n\t"
            "s_nop 0\n\t"
            "v_addc_co_u32 %1, %2, 0, %1, %2"
            : "+v"(acc), "+v"(t2), "=&s"(c1)
            : "v"(x0), "v"(y0), "v"(x1), "v"(y1)
            : "vcc");
    }
    static __device__ __forceinline__ void mac1(uint64_t &acc, uint32_t &t2, uint32_t x0, uint32_t y0) {
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "+v"(acc), "+v"(t2)
            : "v"(x0), "v"(y0)
            : "vcc");
    }
    static __device__ __forceinline__ elem mont_mul(const elem &a, const elem &b) {
        uint64_t acc = 0;
        uint32_t t2 = 0, m[8];
        elem r;
        const uint32_t p6 = 0x11u, p7 = 0x08000000u, one = 1u;
#pragma unroll
        for (int k = 0; k < 15; k++) {
            uint32_t px[12], py[12];
            int n = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int j = k - i;
                if (j >= 0 && j < 8) {
                    px[n] = a.l[i];
                    py[n] = b.l[j];
                    n++;
                }
            }
#pragma unroll
            for (int i = 0; i < 8; i++) {  // m_i exists once column i is closed: i < k
                const int j = k - i;
                if (i < k && j == 6) {
                    px[n] = m[i];
                    py[n] = p6;
                    n++;
                }
                if (i < k && j == 7) {
                    px[n] = m[i];
                    py[n] = p7;
                    n++;
                }
            }
#pragma unroll
            for (int q = 0; q + 1 < n; q += 2) mac2(acc, t2, px[q], py[q], px[q + 1], py[q + 1]);
            if (n & 1) mac1(acc, t2, px[n - 1], py[n - 1]);
            if (k < 8) {
                m[k] = 0u - (uint32_t)acc;
                mac1(acc, t2, m[k], one);  // + m_k * p_0 clears the low word
            } else {
                r.l[k - 8] = (uint32_t)acc;
            }
            acc = (acc >> 32) | ((uint64_t)t2 << 32);
            t2 = 0;
        }
        r.l[7] = (uint32_t)acc;
        const uint32_t top = (uint32_t)(acc >> 32);
        elem u;
        const uint32_t borrow = sub_raw(u, r, modulus());
        return (top | !borrow) ? u : r;
    }
#else
    static elem mont_mul(const elem &a, const elem &b) {
        uint32_t t[10];
        for (int i = 0; i < 10; i++) t[i] = 0;
        for (int i = 0; i < 8; i++) {
            uint64_t c = 0;
            for (int j = 0; j < 8; j++) {
                c += (uint64_t)a.l[j] * b.l[i] + t[j];
                t[j] = (uint32_t)c;
                c >>= 32;
            }
            c += t[8];
            t[8] = (uint32_t)c;
            t[9] = (uint32_t)(c >> 32);
            uint32_t m = 0u - t[0];
            c = ((uint64_t)m * pl(0) + t[0]) >> 32;
            for (int j = 1; j < 8; j++) {
                c += (uint64_t)m * pl(j) + t[j];
                t[j - 1] = (uint32_t)c;
                c >>= 32;
            }
            c += t[8];
            t[7] = (uint32_t)c;
            t[8] = t[9] + (uint32_t)(c >> 32);
        }
        elem s, u;
        for (int i = 0; i < 8; i++) s.l[i] = t[i];
        uint32_t borrow = sub_raw(u, s, modulus());
        return (t[8] | !borrow) ? u : s;
    }
#endif
    SR_HD static elem mul_tw(const elem &a, const elem &w) { return mont_mul(a, w); }
    SR_HD static elem mul_boundary(const elem &a, const elem &b) { return mont_mul(a, b); }
    SR_HD static elem mul_boundary_pre(const elem &a, const elem &b) { return mont_mul(a, b); }
    SR_HD static elem boundary_post(const elem &x) { return x; }
    SR_HD static elem r2() {  // 2^512 mod p
        elem e;
        e.l[0] = 0x7E000401u; e.l[1] = 0xFFFFFD73u; e.l[2] = 0x330FFFFFu; e.l[3] = 0x00000001u;
        e.l[4] = 0xFF6F8000u; e.l[5] = 0xFFFFFFFFu; e.l[6] = 0x5E008810u; e.l[7] = 0x07FFD4ABu;
        return e;
    }
    SR_HD static elem tw_from_u64(uint64_t x) {
        elem e = zero();
        e.l[0] = (uint32_t)x;
        e.l[1] = (uint32_t)(x >> 32);
        return mont_mul(e, r2());
    }
    SR_HD static elem tw_one() { return tw_from_u64(1); }

    // LDS accessors: limb-major (SoA) so that consecutive lanes hit consecutive banks
    SR_HD static elem lds_get(const uint32_t *lds, int idx, int n) {
        elem e;
#pragma unroll
        for (int i = 0; i < 8; i++) e.l[i] = lds[i * n + idx];
        return e;
    }
    SR_HD static void lds_put(uint32_t *lds, int idx, int n, const elem &v) {
#pragma unroll
        for (int i = 0; i < 8; i++) lds[i * n + idx] = v.l[i];
    }
};

// ------------------------------------------------------------------------------------------
// helpers shared by host set-up code and device table builders
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Frog prime p = 15912092521325583641 (frog_ring/mod.rs:19-25, "next" row 4): a generic 64-bit Montgomery field,
// R = 2^64; p > 2^63, so sums carry out of 64 bits.  Data and table constants are both Montgomery images.
// ------------------------------------------------------------------------------------------
struct Frog {
    using elem = uint64_t;
    using storage = uint64_t;
    static constexpr int kStorageWords64 = 1;
    static constexpr int kLdsWords = 2;
    static constexpr uint64_t P = 0xDCD31BD79EC2DD19ull;
    static constexpr uint64_t PINV = 0xA3AE2AD7EED4D0D7ull;  // -p^-1 mod 2^64
    static constexpr uint64_t R1 = 0x232CE428613D22E7ull;    // 2^64 mod p
    static constexpr uint64_t R2 = 0x39662481D8836F74ull;    // 2^128 mod p
    static constexpr uint32_t kGenerator = 3;
    static constexpr int kBoundaryBits = 64;
    static constexpr int kTwoAdicity = 3;

    SR_HD static elem zero() { return 0; }
    SR_HD static elem load(const storage *p) { return *p; }
    SR_HD static void store(storage *p, elem v) { *p = v; }
    SR_HD static bool valid(elem v) { return v < P; }
    SR_HD static elem add(elem a, elem b) {
        uint64_t s = a + b;
        return (s < a || s >= P) ? s - P : s;
    }
    SR_HD static elem sub(elem a, elem b) { return a >= b ? a - b : a + (P - b); }
    SR_HD static elem neg(elem a) { return a ? P - a : 0; }
    SR_HD static elem mont_mul(elem a, elem b) {  // a * b * 2^-64 mod p
        unsigned __int128 t = (unsigned __int128)a * b;
        uint64_t m = (uint64_t)t * PINV;
        unsigned __int128 u = t + (unsigned __int128)m * P;  // low 64 bits vanish; may carry out of 128 bits
        uint64_t hi = (uint64_t)(u >> 64);
        return (u < t || hi >= P) ? hi - P : hi;
    }
    SR_HD static elem mul_tw(elem a, elem w) { return mont_mul(a, w); }
    SR_HD static elem mul_boundary(elem a, elem b) { return mont_mul(a, b); }
    SR_HD static elem tw_from_u64(uint64_t x) { return mont_mul(x % P, R2); }
    SR_HD static elem tw_one() { return R1; }
};

template <class F>
SR_HD typename F::elem pow_tw(typename F::elem base, const uint64_t *e, int e_words) {
    typename F::elem acc = F::tw_one();
    for (int i = 0; i < e_words; i++) {
        uint64_t w = e[i];
        for (int b = 0; b < 64; b++) {
            if ((w >> b) & 1) acc = F::mul_tw(acc, base);
            base = F::mul_tw(base, base);
        }
    }
    return acc;
}

SR_HD uint32_t bitrev(uint32_t x, int bits) { return bits ? (__builtin_bitreverse32(x) >> (32 - bits)) : 0u; }

}  // namespace sr
