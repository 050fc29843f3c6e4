// The reference's frog ring, batched: Fq[X]/(X^16 + 1) -> 4 x Fq4 over p = 15912092521325583641 -- SURVEY 8f #4.
//   crt / icrt        crates/ring/src/cyclotomic_ring/models/frog_ring/ntt.rs:114-151, 163-200
//   homogenize maps   ntt.rs:206-291 (each residue ring Fq[X]/(X^4 - r^e) onto Fq4 = Fq2[v]/(v^2 - u), Fq2 = Fq[u]/(u^2 - r))
//   slot product      Fp4 over Fp2 tower, frog_ring/mod.rs:36-60 (memory (c0.c0, c0.c1, c1.c0, c1.c1))
//   reduce            frog_ring/mod.rs:72-79
// One ring element per lane: its 16 coefficients are one 128-byte line, read and written once.
#pragma once
#include "fields.hpp"

namespace sr {

enum { FROG_CRT = 0, FROG_ICRT = 1, FROG_MUL = 2, FROG_RINGMUL = 3, FROG_REDUCE = 4, FROG_MULB = 5 };  // MULB: slot product with ONE element b

struct FrogConsts {
    uint64_t R[8];  // ROOTS_OF_UNITY_8[k] = w^k, w = 3^((p-1)/8), Montgomery form (ntt.rs:15-24)
    uint64_t inv4;  // FOUR_INV (ntt.rs:27)
};

// (source index, root index) per output index of a block; root -1 = copy, -2 = negate (ntt.rs:222-291)
struct FrogMaps {
    static constexpr signed char HS[4][4] = {{0, 2, 1, 3}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 2, 3, 1}};
    static constexpr signed char HR[4][4] = {{-1, -1, -1, -1}, {-1, 2, 1, 3}, {-1, 1, 6, -2}, {-1, 3, 5, 1}};
    static constexpr signed char DS[4][4] = {{0, 2, 1, 3}, {0, 2, 1, 3}, {0, 3, 1, 2}, {0, 3, 1, 2}};
    static constexpr signed char DR[4][4] = {{-1, -1, -1, -1}, {-1, 7, 6, 5}, {-1, -2, 7, 2}, {-1, 7, 5, 3}};
};

template <bool DEHOMO>
__device__ __forceinline__ void frog_maps(uint64_t *c, const FrogConsts &k) {
    using F = Frog;
#pragma unroll
    for (int blk = 0; blk < 4; blk++) {
        uint64_t old[4];
#pragma unroll
        for (int i = 0; i < 4; i++) old[i] = c[4 * blk + i];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t v = old[DEHOMO ? FrogMaps::DS[blk][i] : FrogMaps::HS[blk][i]];
            const int r = DEHOMO ? FrogMaps::DR[blk][i] : FrogMaps::HR[blk][i];
            c[4 * blk + i] = r == -1 ? v : (r == -2 ? F::neg(v) : F::mul_tw(v, k.R[r]));
        }
    }
}
__device__ __forceinline__ void frog_fwd(uint64_t *a, const FrogConsts &k) {
    using F = Frog;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = a[i], z = F::mul_tw(a[8 + i], k.R[2]);
        a[i] = F::add(x, z);
        a[8 + i] = F::sub(x, z);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t x = a[i], z = F::mul_tw(a[4 + i], k.R[1]);
        a[i] = F::add(x, z);
        a[4 + i] = F::sub(x, z);
        x = a[8 + i];
        z = F::mul_tw(a[12 + i], k.R[3]);
        a[8 + i] = F::add(x, z);
        a[12 + i] = F::sub(x, z);
    }
    frog_maps<false>(a, k);
}
__device__ __forceinline__ void frog_inv(uint64_t *a, const FrogConsts &k) {
    using F = Frog;
    frog_maps<true>(a, k);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t x = a[i], y = a[4 + i];
        a[i] = F::add(x, y);
        a[4 + i] = F::mul_tw(F::sub(x, y), k.R[7]);
        x = a[8 + i];
        y = a[12 + i];
        a[8 + i] = F::add(x, y);
        a[12 + i] = F::mul_tw(F::sub(x, y), k.R[5]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = a[i], y = a[8 + i];
        a[i] = F::mul_tw(F::add(x, y), k.inv4);
        a[8 + i] = F::mul_tw(F::mul_tw(F::sub(x, y), k.R[6]), k.inv4);
    }
}
// Fq2 product (a0 + a1 u)(b0 + b1 u), u^2 = NONRESIDUE = ROOTS[1]
__device__ __forceinline__ void frog_fq2_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const FrogConsts &k) {
    using F = Frog;
    const uint64_t r0 = F::add(F::mont_mul(a[0], b[0]), F::mont_mul(k.R[1], F::mont_mul(a[1], b[1])));
    const uint64_t r1 = F::add(F::mont_mul(a[0], b[1]), F::mont_mul(a[1], b[0]));
    r[0] = r0;
    r[1] = r1;
}
// x <- x * y in Fq4 = Fq2[v]/(v^2 - u): c0 = a0 b0 + u a1 b1, c1 = a0 b1 + a1 b0, with u (t0 + t1 u) = NR t1 + t0 u
__device__ __forceinline__ void frog_fq4_mul(uint64_t *x, const uint64_t *y, const FrogConsts &k) {
    using F = Frog;
    uint64_t p00[2], p11[2], p01[2], p10[2];
    frog_fq2_mul(p00, x, y, k);
    frog_fq2_mul(p11, x + 2, y + 2, k);
    frog_fq2_mul(p01, x, y + 2, k);
    frog_fq2_mul(p10, x + 2, y, k);
    x[0] = F::add(p00[0], F::mont_mul(k.R[1], p11[1]));
    x[1] = F::add(p00[1], p11[0]);
    x[2] = F::add(p01[0], p10[0]);
    x[3] = F::add(p01[1], p10[1]);
}

template <int OP>
__global__ __launch_bounds__(64) void frog16_kernel(FrogConsts k, const uint64_t *a, const uint64_t *b, uint64_t *out, size_t batch) {
    const size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (e >= batch) return;
    uint64_t x[16], y[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = a[e * 16 + i];
    if (OP == FROG_CRT) frog_fwd(x, k);
    if (OP == FROG_ICRT) frog_inv(x, k);
    if (OP == FROG_MUL || OP == FROG_RINGMUL || OP == FROG_MULB) {
#pragma unroll
        for (int i = 0; i < 16; i++) y[i] = b[(OP == FROG_MULB ? 0 : e * 16) + i];
        if (OP == FROG_RINGMUL) {
            frog_fwd(x, k);
            frog_fwd(y, k);
        }
#pragma unroll
        for (int s = 0; s < 4; s++) frog_fq4_mul(x + 4 * s, y + 4 * s, k);
        if (OP == FROG_RINGMUL) frog_inv(x, k);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[e * 16 + i] = x[i];
}
__global__ void frog16_reduce_kernel(const uint64_t *in, size_t in_len, uint64_t *out, size_t batch) {
    const size_t n = batch * 16;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / 16, i = t % 16;
        const uint64_t *src = in + e * in_len;
        const uint64_t lo = i < in_len ? src[i] : 0, hi = 16 + i < in_len ? src[16 + i] : 0;
        out[t] = Frog::sub(lo, hi);
    }
}

inline void frog_init(FrogConsts &c) {
    using F = Frog;
    const uint64_t e[1] = {(F::P - 1) / 8}, pm2[1] = {F::P - 2};
    const uint64_t w = pow_tw<F>(F::tw_from_u64(F::kGenerator), e, 1);
    uint64_t r = F::tw_one();
    for (int i = 0; i < 8; i++) {
        c.R[i] = r;
        r = F::mul_tw(r, w);
    }
    c.inv4 = pow_tw<F>(F::tw_from_u64(4), pm2, 1);
}
// op: FROG_*; returns non-zero on launch failure
inline int frog_launch(const FrogConsts &c, int op, const uint64_t *a, const uint64_t *b, size_t in_len, uint64_t *out,
                       size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (op == FROG_REDUCE) {
        size_t blocks = (batch * 16 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(frog16_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a, in_len, out, batch);
        return hipGetLastError() != hipSuccess;
    }
    const size_t blocks = (batch + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return 1;
    dim3 g((unsigned)blocks), t(64);
    switch (op) {
        case FROG_CRT: hipLaunchKernelGGL(frog16_kernel<FROG_CRT>, g, t, 0, st, c, a, b, out, batch); break;
        case FROG_ICRT: hipLaunchKernelGGL(frog16_kernel<FROG_ICRT>, g, t, 0, st, c, a, b, out, batch); break;
        case FROG_MUL: hipLaunchKernelGGL(frog16_kernel<FROG_MUL>, g, t, 0, st, c, a, b, out, batch); break;
        case FROG_RINGMUL: hipLaunchKernelGGL(frog16_kernel<FROG_RINGMUL>, g, t, 0, st, c, a, b, out, batch); break;
        case FROG_MULB: hipLaunchKernelGGL(frog16_kernel<FROG_MULB>, g, t, 0, st, c, a, b, out, batch); break;
        default: return 1;
    }
    return hipGetLastError() != hipSuccess;
}

}  // namespace sr
