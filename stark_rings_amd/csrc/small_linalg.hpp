// Linear algebra over the reference's OWN RqNTT types (finished in round 2): the only concrete
// CyclotomicPolyRingNTTGeneral the reference ships have extension-field slots -- Goldilocks-24: 8 x Fq3
// (goldilocks/mod.rs:69-119), BabyBear-72: 8 x Fq9 (babybear/mod.rs:81-131), Frog-16: 4 x Fq4 (frog_ring/mod.rs:62-107) -- and
// Matrix<R>::checked_mul_vec / checked_mul_mat (crates/linear_algebra/src/matrix.rs:148-178) and
// SparseMatrix<R>::checked_mul_vec (sparse_matrix.rs:201-212) are generic over R.  Here a lane owns ONE slot (W consecutive
// coefficients of one ring element): it multiplies slots with the ring's slot product (ntt_form.rs:159-189 through
// small_slot_mul / frog_fq4_mul, the very code the fused ring products use) and sums them coefficient-wise; a workgroup owns
// one output element and splits the inner dimension over its lanes.
#pragma once
#include "frog_ring.hpp"
#include "small_rings.hpp"

namespace sr {

struct SlotG24 {
    using F = Goldilocks;
    using K = SmallRingConsts;
    static constexpr int W = 3, D = 24;
    static __device__ __forceinline__ void mul(F::elem *x, const F::elem *y, const K &k) { small_slot_mul<F, W>(x, y, k); }
};
struct SlotB72 {
    using F = BabyBear;
    using K = SmallRingConsts;
    static constexpr int W = 9, D = 72;
    static __device__ __forceinline__ void mul(F::elem *x, const F::elem *y, const K &k) { small_slot_mul<F, W>(x, y, k); }
};
struct SlotFrog {
    using F = Frog;
    using K = FrogConsts;
    static constexpr int W = 4, D = 16;
    static __device__ __forceinline__ void mul(F::elem *x, const F::elem *y, const K &k) { frog_fq4_mul(x, y, k); }
};

template <class SL>
__device__ __forceinline__ void slot_load(typename SL::F::elem *x, const uint64_t *p) {
#pragma unroll
    for (int i = 0; i < SL::W; i++) x[i] = SL::F::load(p + i);
}
// acc += x * y
template <class SL>
__device__ __forceinline__ void slot_fma(typename SL::F::elem *acc, typename SL::F::elem *x, const typename SL::F::elem *y,
                                         const typename SL::K &k) {
    SL::mul(x, y, k);
#pragma unroll
    for (int i = 0; i < SL::W; i++) acc[i] = SL::F::add(acc[i], x[i]);
}
template <class SL>
__device__ __forceinline__ void slot_store(uint64_t *p, const typename SL::F::elem *acc) {
#pragma unroll
    for (int i = 0; i < SL::W; i++) SL::F::store(p + i, acc[i]);
}

// ---- sums of slot products without a reduction per term (round 3) --------------------------------------------------------------
// A matrix product is a SUM of slot products, and all three slot fields are Fq[X]/(X^W - NR) in a flat basis (memory index m holds
// the coefficient of X^PERM[m]: Fq3 as is; Fq9 = Fq3[v]/(v^3 - u) with v = X, u = X^3; Fq4 = Fq2[v]/(v^2 - u), u^2 = r, with v = X).
// So sum_t x_t y_t = fold(P_0 .. P_{2W-2}) with P_e = sum_t sum_{i+j=e} x_t[i] y_t[j], a plain integer sum of products of memory
// images, and the modular work -- the Montgomery step that removes one factor R = 2^64 and the fold P_e + NR P_{e+W} -- happens
// ONCE per output instead of once per term.  The integer sums live in 96-bit accumulators (a 64-bit word plus a count of its
// carries): one v_mad_u64_u32 with carry-out plus one v_addc_co_u32 per 32 x 32-bit partial product.  Per slot product and term:
// Frog-16 16 x 8 = 128 VALU against ~700 for frog_fq4_mul + adds (21 64-bit Montgomery products), BabyBear-72 81 x 2 = 162
// against ~300, Goldilocks-24 9 x 8 = 72 against ~160.  Results are the same field elements bit for bit (everything is canonical
// at the end); tests/test_gpu_parity.py::test_linear_algebra_over_the_reference_rings and ::test_long_sums_* pin them.
struct Acc96 {
    uint64_t lo;
    uint32_t hi;
};
// (c00, cm, c11) += (al bl, al bh + ah bl, ah bh) for a = ah 2^32 + al, b = bh 2^32 + bl: eight VALU, the carry of each multiply-add
// (an SGPR pair) consumed at least two instructions after it is produced (gfx950: 2 wait states between a VALU write of an SGPR and
// a VALU read of it)
__device__ __forceinline__ void mac3x96(Acc96 &c00, Acc96 &cm, Acc96 &c11, uint64_t a, uint64_t b) {
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t s0, s1, s2, s3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %2, %7, %10, %13, %2\n\t"
        "v_mad_u64_u32 %4, %8, %11, %13, %4\n\t"
        "v_addc_co_u32_e64 %1, %6, 0, %1, %6\n\t"
        "v_mad_u64_u32 %2, %9, %11, %12, %2\n\t"
        "v_addc_co_u32_e64 %3, %7, 0, %3, %7\n\t"
        "v_addc_co_u32_e64 %5, %8, 0, %5, %8\n\t"
        "v_addc_co_u32_e64 %3, %9, 0, %3, %9"
        : "+v"(c00.lo), "+v"(c00.hi), "+v"(cm.lo), "+v"(cm.hi), "+v"(c11.lo), "+v"(c11.hi), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh));
}
// acc += a b for 32-bit factors (BabyBear images): two VALU
__device__ __forceinline__ void mac96(Acc96 &c, uint32_t a, uint32_t b) {
    uint64_t s0;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %1, %2, 0, %1, %2"
        : "+v"(c.lo), "+v"(c.hi), "=&s"(s0)
        : "v"(a), "v"(b));
}

// two independent accumulators at once: the second multiply-add and one s_nop 0 are the wait states of the first carry
__device__ __forceinline__ void mac96x2(Acc96 &c, uint32_t a, uint32_t b, Acc96 &d, uint32_t e, uint32_t f) {
    uint64_t s0, s1;
    asm("v_mad_u64_u32 %0, %4, %6, %7, %0\n\t"
        "v_mad_u64_u32 %2, %5, %8, %9, %2\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32_e64 %1, %4, 0, %1, %4\n\t"
        "v_addc_co_u32_e64 %3, %5, 0, %3, %5"
        : "+v"(c.lo), "+v"(c.hi), "+v"(d.lo), "+v"(d.hi), "=&s"(s0), "=&s"(s1)
        : "v"(a), "v"(b), "v"(e), "v"(f));
}

template <class SL> struct SlotDot;
// Goldilocks-24: five exponents x three classes.  finish: S00 2^-64 + Smid 2^-32 + S11 (SumOfProducts<Goldilocks>, ntt_generic.hpp)
template <> struct SlotDot<SlotG24> {
    Acc96 a[5][3];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int e = 0; e < 5; e++)
#pragma unroll
            for (int c = 0; c < 3; c++) a[e][c] = Acc96{0, 0};
    }
    __device__ __forceinline__ void fma(const uint64_t *x, const uint64_t *y) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) mac3x96(a[i + j][0], a[i + j][1], a[i + j][2], x[i], y[j]);
    }
    __device__ __forceinline__ void finish(uint64_t *out, const SmallRingConsts &k) const {
        using G = Goldilocks;
        const uint64_t inv32 = 0xFFFFFFFE00000002ull;  // 2^-32 mod p
        uint64_t t[5];
#pragma unroll
        for (int e = 0; e < 5; e++) {
            const uint64_t s00 = G::reduce128(a[e][0].lo, a[e][0].hi), sm = G::reduce128(a[e][1].lo, a[e][1].hi),
                           s11 = G::reduce128(a[e][2].lo, a[e][2].hi);
            t[e] = G::add(G::add(G::mul_boundary(s00, 1), G::mul(sm, inv32)), s11);
        }
        const uint64_t nr = sc_get<G>(k.R[1]);
#pragma unroll
        for (int i = 0; i < 2; i++) t[i] = G::add(t[i], G::mul_tw(t[i + 3], nr));
#pragma unroll
        for (int m = 0; m < 3; m++) out[m] = t[HomoTables<3>::PERM[m]];
    }
};
// BabyBear-72: seventeen exponents, raw 62-bit products.  finish: two Montgomery word steps take V < 2^95 to V 2^-64 mod p
template <> struct SlotDot<SlotB72> {
    Acc96 a[17];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int e = 0; e < 17; e++) a[e] = Acc96{0, 0};
    }
    __device__ __forceinline__ void fma(const uint32_t *x, const uint32_t *y) {
        using T = HomoTables<9>;
        uint32_t px[9], py[9];
#pragma unroll
        for (int m = 0; m < 9; m++) {
            px[T::PERM[m]] = x[m];
            py[T::PERM[m]] = y[m];
        }
#pragma unroll
        for (int i = 0; i < 9; i++) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) mac96x2(a[i + j], px[i], py[j], a[i + j + 1], px[i], py[j + 1]);
            mac96(a[i + 8], px[i], py[8]);
        }
    }
    // V = lo + hi 2^64, hi < 2^31 (fewer than 2^29 terms): V 2^-64 mod p, canonical
    static __device__ __forceinline__ uint32_t redc(const Acc96 &v) {
        using B = BabyBear;
        const uint32_t m1 = (uint32_t)v.lo * B::PINV;
        const unsigned __int128 w1 = ((unsigned __int128)v.hi << 64 | v.lo) + (uint64_t)m1 * B::P;
        const uint64_t v1 = (uint64_t)(w1 >> 32);                       // < 2^63 + 2^31
        const uint32_t m2 = (uint32_t)v1 * B::PINV;
        uint64_t v2 = (uint64_t)(((unsigned __int128)v1 + (uint64_t)m2 * B::P) >> 32);  // <= 2^32
        if (v2 >= B::P) v2 -= B::P;
        if (v2 >= B::P) v2 -= B::P;
        return (uint32_t)v2;
    }
    __device__ __forceinline__ void finish(uint32_t *out, const SmallRingConsts &k) const {
        using B = BabyBear;
        uint32_t t[17];
#pragma unroll
        for (int e = 0; e < 17; e++) t[e] = redc(a[e]);
        const uint32_t nr = sc_get<B>(k.R[1]);  // table form (times 2^32): mul_tw(t, nr) = t NR
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = B::add(t[i], B::mul_tw(t[i + 9], nr));
#pragma unroll
        for (int m = 0; m < 9; m++) out[m] = t[HomoTables<9>::PERM[m]];
    }
};
// Frog-16: memory (c0.c0, c0.c1, c1.c0, c1.c1) = coefficients of (1, u, v, u v) = X^(0, 2, 1, 3) with v = X, u = X^2, X^4 = r
// (frog_ring/mod.rs:36-60).  Seven exponents x three classes.  finish: V = S00 + Smid 2^32 + S11 2^64 < 2^192, one Montgomery word
// step (the sum carries R^2, the image wants R), the rest reduced with 2^64 = R (mod p).
template <> struct SlotDot<SlotFrog> {
    Acc96 a[7][3];
    static constexpr int PERM[4] = {0, 2, 1, 3};
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int e = 0; e < 7; e++)
#pragma unroll
            for (int c = 0; c < 3; c++) a[e][c] = Acc96{0, 0};
    }
    __device__ __forceinline__ void fma(const uint64_t *x, const uint64_t *y) {
        uint64_t px[4], py[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            px[PERM[m]] = x[m];
            py[PERM[m]] = y[m];
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) mac3x96(a[i + j][0], a[i + j][1], a[i + j][2], px[i], py[j]);
    }
    static __device__ __forceinline__ uint64_t redc(const Acc96 *c) {
        using F = Frog;
        typedef unsigned __int128 u128;
        // V = w0 + w1 2^64 + w2 2^128
        u128 lo = (u128)c[0].lo + ((u128)c[1].lo << 32);                                   // < 2^97
        u128 hi = (u128)c[0].hi + ((u128)c[1].hi << 32) + c[2].lo + (lo >> 64);            // weight 2^64, < 2^66
        const uint64_t w0 = (uint64_t)lo, w1 = (uint64_t)hi;
        const uint64_t w2 = (uint64_t)(hi >> 64) + c[2].hi;                                // weight 2^128, < 2^34
        // one word step: (V + m p) / 2^64 with m = -w0 / p mod 2^64
        const uint64_t m = w0 * F::PINV;
        const u128 mp = (u128)m * F::P;
        const u128 t = (u128)w0 + (uint64_t)mp;                                            // low word vanishes; carry 0 or 1
        const u128 q = (u128)w1 + (uint64_t)(mp >> 64) + (uint64_t)(t >> 64);              // weight 1, < 2^66
        const uint64_t q0 = (uint64_t)q, q1 = (uint64_t)(q >> 64) + w2;                    // q1 < 2^35: weight 2^64 = R (mod p)
        const uint64_t r0 = q0 >= F::P ? q0 - F::P : q0;
        return F::add(r0, F::mont_mul(q1, F::R2));                                         // q1 R2 / R = q1 R
    }
    __device__ __forceinline__ void finish(uint64_t *out, const FrogConsts &k) const {
        using F = Frog;
        uint64_t t[7];
#pragma unroll
        for (int e = 0; e < 7; e++) t[e] = redc(a[e]);
#pragma unroll
        for (int i = 0; i < 3; i++) t[i] = F::add(t[i], F::mont_mul(k.R[1], t[i + 4]));
#pragma unroll
        for (int m = 0; m < 4; m++) out[m] = t[PERM[m]];
    }
};

// One workgroup (256 lanes) per output ring element (mat-vec, sparse mat-vec) or per RB x CB block of them (mat-mat).  Lane t =
// (term group g, slot sl) with S = D / W slots per element: consecutive lanes read consecutive slots, i.e. whole ring elements, so
// the streams of M (or the CSR values) are read with full cache lines; a lane sums every G-th term of the inner dimension in its
// SlotDot, finishes it once, and the partial sums of a slot meet in LDS.
template <class SL>
__device__ __forceinline__ void slot_reduce_store(typename SL::F::elem *lds, const typename SL::F::elem *acc, uint64_t *dst) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 256 / S;
    const int t = threadIdx.x, g = t / S, s = t % S;
#pragma unroll
    for (int i = 0; i < SL::W; i++) lds[(g * S + s) * SL::W + i] = acc[i];
    __syncthreads();
    if (t < SL::D) {  // one lane per coefficient of the output element (slot t / W, coefficient t % W)
        E sum = lds[t];
        for (int gg = 1; gg < G; gg++) sum = SL::F::add(sum, lds[gg * SL::D + t]);
        SL::F::store(dst + t, sum);
    }
}
// y[r] = sum_c M[r][c] * v[c].  nsplit > 1 (few rows): workgroup (r, sp) sums columns [sp * cpw, (sp + 1) * cpw) into
// part[(r * nsplit + sp)] and slot_sum_kernel adds the parts -- so that a short, wide matrix still fills the chip.
template <class SL>
__global__ __launch_bounds__(256) void slot_matvec_kernel(typename SL::K k, uint64_t *y, const uint64_t *m, const uint64_t *v,
                                                          size_t nrows, size_t ncols, unsigned nsplit, size_t cpw) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 256 / S;
    static_assert(256 % S == 0 && SL::D <= 256, "slot layout");
    __shared__ E lds[256 * SL::W];
    const size_t r = blockIdx.x / nsplit;
    const unsigned sp = blockIdx.x % nsplit;
    const int g = threadIdx.x / S, s = threadIdx.x % S;
    const size_t c0 = (size_t)sp * cpw, c1 = c0 + cpw < ncols ? c0 + cpw : ncols;
    E x[SL::W], z[SL::W], res[SL::W];
    SlotDot<SL> acc;
    acc.init();
    for (size_t c = c0 + g; c < c1; c += G) {
        slot_load<SL>(x, m + (r * ncols + c) * SL::D + s * SL::W);
        slot_load<SL>(z, v + c * SL::D + s * SL::W);
        acc.fma(x, z);
    }
    acc.finish(res, k);
    slot_reduce_store<SL>(lds, res, y + (r * nsplit + sp) * SL::D);
}
// y[r] = sum_sp part[r * nsplit + sp], coefficient-wise
template <class SL>
__global__ __launch_bounds__(256) void slot_sum_kernel(uint64_t *y, const uint64_t *part, size_t nrows, unsigned nsplit) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= nrows * SL::D) return;
    const size_t r = i / SL::D, cidx = i % SL::D;
    typename SL::F::elem sum = SL::F::load(part + (r * nsplit) * SL::D + cidx);
    for (unsigned sp = 1; sp < nsplit; sp++) sum = SL::F::add(sum, SL::F::load(part + (r * nsplit + sp) * SL::D + cidx));
    SL::F::store(y + i, sum);
}
// CSR sparse matrix times vector; an entry with column >= ncols is skipped and counted once (the reference panics there)
template <class SL>
__global__ __launch_bounds__(256) void slot_spmv_kernel(typename SL::K k, uint64_t *y, const uint64_t *vals, const uint32_t *cols,
                                                        const uint64_t *row_ptr, const uint64_t *v, size_t nrows, size_t ncols,
                                                        unsigned long long *bad) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 256 / S;
    __shared__ E lds[256 * SL::W];
    const size_t r = blockIdx.x;
    const int g = threadIdx.x / S, s = threadIdx.x % S;
    E x[SL::W], z[SL::W], res[SL::W];
    SlotDot<SL> acc;
    acc.init();
    const uint64_t j1 = row_ptr[r + 1];
    for (uint64_t j = row_ptr[r] + g; j < j1; j += G) {
        const uint32_t c = cols[j];
        if (c >= ncols) {
            if (s == 0) atomicAdd(bad, 1ull);
            continue;
        }
        slot_load<SL>(x, vals + j * SL::D + s * SL::W);
        slot_load<SL>(z, v + (size_t)c * SL::D + s * SL::W);
        acc.fma(x, z);
    }
    acc.finish(res, k);
    slot_reduce_store<SL>(lds, res, y + r * SL::D);
}
// Y (n x p) = A (n x m) * B (m x p).  A workgroup owns a 2 x 2 block of outputs: lane t = (output o = t / 64, term group g, slot sl),
// G = 64 / S term groups.  Per step of G inner indices the 2 x G elements of A and the G x 2 elements of B the block needs go
// through LDS ONCE (lane-contiguous global reads of whole ring elements) and serve both outputs of their row / column: half the
// operand traffic of one output per workgroup, and every lane works on one of four independent sums.
template <class SL>
__global__ __launch_bounds__(256) void slot_matmul_kernel(typename SL::K k, uint64_t *y, const uint64_t *a, const uint64_t *b,
                                                          size_t n, size_t m, size_t p) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 64 / S, D = SL::D, W = SL::W;
    static_assert(64 % S == 0, "slot layout");
    __shared__ E lds[4 * G * D > 256 * W ? 4 * G * D : 256 * W];  // [A: 2 x G elements | B: G x 2 elements], then the partial sums
    const size_t pb = (p + 1) / 2;
    const size_t i0 = (blockIdx.x / pb) * 2, j0 = (blockIdx.x % pb) * 2;
    const int t = threadIdx.x, o = t >> 6, u = t & 63, g = u / S, s = u % S;
    const int oi = o >> 1, oj = o & 1;
    E x[W], z[W], res[W];
    SlotDot<SL> acc;
    acc.init();
    for (size_t t0 = 0; t0 < m; t0 += G) {
        __syncthreads();  // the previous step's reads are done
        for (int idx = t; idx < 4 * G * D; idx += 256) {
            const int el = idx / D, w = idx % D;          // el < 2 G: A[i0 + el / G][t0 + el % G]; else B[t0 + (el - 2G) / 2][j0 + (el - 2G) % 2]
            E val = SL::F::zero();
            if (el < 2 * G) {
                const size_t ii = i0 + el / G, tt = t0 + el % G;
                if (ii < n && tt < m) val = SL::F::load(a + (ii * m + tt) * D + w);
            } else {
                const size_t tt = t0 + (el - 2 * G) / 2, jj = j0 + (el - 2 * G) % 2;
                if (tt < m && jj < p) val = SL::F::load(b + (tt * p + jj) * D + w);
            }
            lds[idx] = val;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < W; q++) {
            x[q] = lds[(oi * G + g) * D + s * W + q];
            z[q] = lds[(2 * G + g * 2 + oj) * D + s * W + q];
        }
        acc.fma(x, z);  // terms beyond m were loaded as zero
    }
    acc.finish(res, k);
    __syncthreads();
    // partial sums: [o][g][slot][W]; then lane (o, coefficient c) adds the G groups
#pragma unroll
    for (int q = 0; q < W; q++) lds[((o * G + g) * S + s) * W + q] = res[q];
    __syncthreads();
    if (u < D || (D > 64)) {
        for (int c = u; c < D; c += 64) {
            E sum = lds[(o * G) * D + c];
            for (int gg = 1; gg < G; gg++) sum = SL::F::add(sum, lds[(o * G + gg) * D + c]);
            const size_t ii = i0 + oi, jj = j0 + oj;
            if (ii < n && jj < p) SL::F::store(y + (ii * p + jj) * D + c, sum);
        }
    }
}

// part: device scratch of nrows * nsplit ring elements when the launcher splits rows (may be null when it does not); returns the
// number of parts a row is cut into for (nrows, ncols): 1 when there are enough rows to fill the chip
inline unsigned slot_matvec_splits(size_t nrows, size_t ncols, int groups) {
    if (nrows == 0 || nrows >= 1024) return 1;                      // >= 4 workgroups per CU already
    size_t want = (2048 + nrows - 1) / nrows;
    const size_t max_by_cols = ncols / ((size_t)groups * 4);        // at least four terms per lane and part
    if (want > max_by_cols) want = max_by_cols;
    return (unsigned)(want < 1 ? 1 : (want > 64 ? 64 : want));
}
template <class SL>
inline int slot_matvec(const typename SL::K &k, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols,
                       uint64_t *part, unsigned nsplit, hipStream_t st) {
    if (nrows == 0) return 0;
    if (!part) nsplit = 1;
    const size_t blocks = nrows * nsplit;
    if (blocks > 0x7FFFFFFFull) return 1;
    const size_t cpw = (ncols + nsplit - 1) / nsplit;
    hipLaunchKernelGGL((slot_matvec_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, nsplit > 1 ? part : y, m, v, nrows, ncols,
                       nsplit, cpw);
    if (hipGetLastError() != hipSuccess) return 1;
    if (nsplit > 1) {
        hipLaunchKernelGGL((slot_sum_kernel<SL>), dim3((unsigned)((nrows * SL::D + 255) / 256)), dim3(256), 0, st, y, part, nrows, nsplit);
        if (hipGetLastError() != hipSuccess) return 1;
    }
    return 0;
}
template <class SL>
inline int slot_spmv(const typename SL::K &k, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr,
                     const uint64_t *v, size_t nrows, size_t ncols, unsigned long long *bad, hipStream_t st) {
    const size_t blocks = nrows;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_spmv_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, vals, cols, row_ptr, v, nrows, ncols, bad);
    return hipGetLastError() != hipSuccess;
}
template <class SL>
inline int slot_matmul(const typename SL::K &k, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p,
                       hipStream_t st) {
    const size_t blocks = ((n + 1) / 2) * ((p + 1) / 2);
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_matmul_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, a, b, n, m, p);
    return hipGetLastError() != hipSuccess;
}

}  // namespace sr
