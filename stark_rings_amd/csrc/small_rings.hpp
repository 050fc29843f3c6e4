// Reference-native partially-splitting rings, batched: one ring element per lane.
//   Goldilocks  Fq[X]/(X^24 - X^12 + 1) -> 8 x Fq3   crates/ring/src/cyclotomic_ring/models/goldilocks/ntt.rs:135-437
//   BabyBear    Fq[X]/(X^72 - X^36 + 1) -> 8 x Fq9   crates/ring/src/cyclotomic_ring/models/babybear/ntt.rs:143-588
// Both share one shape with h = D/2, q = D/4, e = D/8 (goldilocks/ntt.rs:146-225 == babybear/ntt.rs:154-233):
//   stage Z : z = zeta a[h+i]; (a[i], a[h+i]) <- (a[i] + z, a[i] + a[h+i] - z), zeta = ROOTS[4]
//   stage 1 : radix-2 blocks at 0, h with ROOTS[2], ROOTS[10]
//   stage 2 : radix-2 blocks at 0, q, h, 3q with ROOTS[1], [7], [5], [11]
//   homogenize: per residue block a signed/twisted permutation into the canonical Fq3 / Fq9
// Slot products are the reference's CubicExtField products on the in-memory (Montgomery) images
// (ntt_form.rs:177-189 with BaseCRTField = Fq3 / Fq9; goldilocks/mod.rs:34-54, babybear/mod.rs:33-66, fq9.rs:7-58).
// Data is treated as plain residues by the linear maps (see fields.hpp); ROOTS are kept in table form.
#pragma once
#include <type_traits>
#include "fields.hpp"

namespace sr {

enum {
    SMALL_G24_CRT, SMALL_G24_ICRT, SMALL_G24_MUL, SMALL_G24_RINGMUL, SMALL_G24_REDUCE,
    SMALL_B72_CRT, SMALL_B72_ICRT, SMALL_B72_MUL, SMALL_B72_RINGMUL, SMALL_B72_REDUCE
};
enum { SOP_CRT = 0, SOP_ICRT = 1, SOP_MUL = 2, SOP_RINGMUL = 3, SOP_MULB = 5 };  // MULB: slot product with ONE element b for the whole batch

// ROOTS_OF_UNITY_24[k] = omega^k, omega = g^((p-1)/24) (goldilocks/ntt.rs:15-40, babybear/ntt.rs:16-41),
// KAPPA = (2 zeta - 1)^-1, 1/8, 1/4 (goldilocks/ntt.rs:42-47, babybear/ntt.rs:136-141); table form, 64-bit slots.
struct SmallRingConsts {
    uint64_t R[24];
    uint64_t kappa, inv8, inv4;
    int is_goldilocks;
};

template <class F>
SR_HD typename F::elem sc_get(uint64_t v) { return (typename F::elem)v; }

// ---- homogenize tables.  For block b = 1..7 (residues e = 13,7,19,5,17,11,23): dst[i] = +-src[SRC[i]] * R[ROOT[i]],
// ROOT -1: copy, -2: negate.  Goldilocks goldilocks/ntt.rs:350-437; BabyBear babybear/ntt.rs:351-578 (maps act
// before the (1,3),(2,6),(5,7) swap in homogenize and after it in dehomogenize, :364,:369,...).
template <int W> struct HomoTables;
template <> struct HomoTables<3> {
    static constexpr signed char HS[7][3] = {{0,1,2},{0,1,2},{0,1,2},{0,2,1},{0,2,1},{0,2,1},{0,2,1}};
    static constexpr signed char HR[7][3] = {{-1,-2,-1},{-1,2,4},{-1,6,12},{-1,3,1},{-1,11,5},{-1,7,3},{-1,15,7}};
    static constexpr signed char DS[7][3] = {{0,1,2},{0,1,2},{0,1,2},{0,2,1},{0,2,1},{0,2,1},{0,2,1}};
    static constexpr signed char DR[7][3] = {{-1,-2,-1},{-1,22,20},{-1,18,12},{-1,23,21},{-1,19,13},{-1,21,17},{-1,17,9}};
    static constexpr signed char PERM[3] = {0, 1, 2};
};
template <> struct HomoTables<9> {
    static constexpr signed char HS[7][9] = {
        {0,7,5,3,1,8,6,4,2}, {0,4,8,3,7,2,6,1,5}, {0,1,2,3,4,5,6,7,8}, {0,2,4,6,8,1,3,5,7},
        {0,8,7,6,5,4,3,2,1}, {0,5,1,6,2,7,3,8,4}, {0,2,4,6,8,1,3,5,7}};
    static constexpr signed char HR[7][9] = {
        {-1,10,7,4,1,11,8,5,2}, {-1,3,6,2,5,1,4,-1,3}, {-1,2,4,6,8,10,-2,14,16}, {-1,1,2,3,4,-1,1,2,3},
        {-1,15,13,11,9,7,5,3,1}, {-1,6,1,7,2,8,3,9,4}, {-1,5,10,15,20,2,7,-2,17}};
    static constexpr signed char DS[7][9] = {
        {0,4,8,3,7,2,6,1,5}, {0,7,5,3,1,8,6,4,2}, {0,1,2,3,4,5,6,7,8}, {0,5,1,6,2,7,3,8,4},
        {0,8,7,6,5,4,3,2,1}, {0,2,4,6,8,1,3,5,7}, {0,5,1,6,2,7,3,8,4}};
    static constexpr signed char DR[7][9] = {
        {-1,23,22,20,19,17,16,14,13}, {-1,-1,23,22,21,21,20,19,18}, {-1,22,20,18,16,14,-2,10,8}, {-1,-1,23,23,22,22,21,21,20},
        {-1,23,21,19,17,15,13,11,9}, {-1,23,22,21,20,18,17,16,15}, {-1,22,19,17,14,-2,9,7,4}};
    static constexpr signed char PERM[9] = {0, 3, 6, 1, 4, 7, 2, 5, 8};  // involution (1,3)(2,6)(5,7): babybear/ntt.rs:580-588
};

template <class F, int W>
__device__ __forceinline__ void small_homogenize(typename F::elem *c, const SmallRingConsts &k) {
    using E = typename F::elem;
    using T = HomoTables<W>;
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        E old[W], mid[W];
#pragma unroll
        for (int i = 0; i < W; i++) old[i] = c[blk * W + i];
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (blk == 0) {
                mid[i] = old[i];
            } else {
                E v = old[T::HS[blk - 1][i]];
                int r = T::HR[blk - 1][i];
                mid[i] = r == -1 ? v : (r == -2 ? F::neg(v) : F::mul_tw(v, sc_get<F>(k.R[r])));
            }
        }
#pragma unroll
        for (int i = 0; i < W; i++) c[blk * W + i] = mid[T::PERM[i]];
    }
}
template <class F, int W>
__device__ __forceinline__ void small_dehomogenize(typename F::elem *c, const SmallRingConsts &k) {
    using E = typename F::elem;
    using T = HomoTables<W>;
#pragma unroll
    for (int blk = 0; blk < 8; blk++) {
        E old[W];
#pragma unroll
        for (int i = 0; i < W; i++) old[i] = c[blk * W + T::PERM[i]];
#pragma unroll
        for (int i = 0; i < W; i++) {
            if (blk == 0) {
                c[blk * W + i] = old[i];
            } else {
                E v = old[T::DS[blk - 1][i]];
                int r = T::DR[blk - 1][i];
                c[blk * W + i] = r == -1 ? v : (r == -2 ? F::neg(v) : F::mul_tw(v, sc_get<F>(k.R[r])));
            }
        }
    }
}

template <class F, int D>
__device__ __forceinline__ void small_fwd3(typename F::elem *a, const SmallRingConsts &k) {
    using E = typename F::elem;
    constexpr int h = D / 2, q = D / 4, e = D / 8;
    const E zeta = sc_get<F>(k.R[4]);
#pragma unroll
    for (int i = 0; i < h; i++) {
        E ci = a[i], cj = a[h + i];
        E z = F::mul_tw(cj, zeta);
        a[i] = F::add(ci, z);
        a[h + i] = F::sub(F::add(ci, cj), z);
    }
    constexpr int r1[2] = {2, 10};
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const E w = sc_get<F>(k.R[r1[t]]);
#pragma unroll
        for (int i = 0; i < q; i++) {
            E u = a[t * h + i], v = F::mul_tw(a[t * h + q + i], w);
            a[t * h + i] = F::add(u, v);
            a[t * h + q + i] = F::sub(u, v);
        }
    }
    constexpr int r2[4] = {1, 7, 5, 11};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const E w = sc_get<F>(k.R[r2[t]]);
#pragma unroll
        for (int i = 0; i < e; i++) {
            E u = a[t * q + i], v = F::mul_tw(a[t * q + e + i], w);
            a[t * q + i] = F::add(u, v);
            a[t * q + e + i] = F::sub(u, v);
        }
    }
}
template <class F, int D>
__device__ __forceinline__ void small_inv3(typename F::elem *a, const SmallRingConsts &k) {
    using E = typename F::elem;
    constexpr int h = D / 2, q = D / 4, e = D / 8;
    constexpr int r2[4] = {23, 17, 19, 13};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const E w = sc_get<F>(k.R[r2[t]]);
#pragma unroll
        for (int i = 0; i < e; i++) {
            E u = a[t * q + i], v = a[t * q + e + i];
            a[t * q + i] = F::add(u, v);
            a[t * q + e + i] = F::mul_tw(F::sub(u, v), w);
        }
    }
    constexpr int r1[2] = {22, 14};
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const E w = sc_get<F>(k.R[r1[t]]);
#pragma unroll
        for (int i = 0; i < q; i++) {
            E u = a[t * h + i], v = a[t * h + q + i];
            a[t * h + i] = F::add(u, v);
            a[t * h + q + i] = F::mul_tw(F::sub(u, v), w);
        }
    }
    const E kappa = sc_get<F>(k.kappa), i8 = sc_get<F>(k.inv8), i4 = sc_get<F>(k.inv4);
#pragma unroll
    for (int i = 0; i < h; i++) {
        E ci = a[i], cj = a[h + i];
        E kd = F::mul_tw(F::sub(ci, cj), kappa);
        a[i] = F::mul_tw(F::sub(F::add(ci, cj), kd), i8);
        a[h + i] = F::mul_tw(kd, i4);
    }
}

// x <- x * y in Fq[X]/(X^W - NONRESIDUE) on in-memory images; memory index m <-> X^PERM[m]
template <class F, int W>
__device__ __forceinline__ void small_slot_mul(typename F::elem *x, const typename F::elem *y, const SmallRingConsts &k) {
    using E = typename F::elem;
    using T = HomoTables<W>;
    E px[W], py[W], t[2 * W - 1];
#pragma unroll
    for (int m = 0; m < W; m++) {
        px[T::PERM[m]] = x[m];
        py[T::PERM[m]] = y[m];
    }
    if constexpr (std::is_same<F, BabyBear>::value) {
        // Lazy column sums: four raw 62-bit products fit a u64 (4 (p-1)^2 < 2^64), so a column of c terms costs c multiply-adds
        // and ceil(c / 4) Montgomery reductions instead of c of each: 81 + 27 x 6 + 30 VALU for the Fq9 schoolbook, against
        // 81 x 8.  A group sum T < 2 p 2^32 first loses p 2^32 if it can (high word only), then reduces as usual.
#pragma unroll
        for (int kk = 0; kk < 2 * W - 1; kk++) {
            E col = 0;
            const int lo = kk < W ? 0 : kk - W + 1, hi_i = kk < W ? kk : W - 1;
#pragma unroll
            for (int g = lo; g <= hi_i; g += 4) {
                uint64_t acc = 0;
#pragma unroll
                for (int i = g; i < g + 4; i++)
                    if (i <= hi_i) acc += (uint64_t)px[i] * py[kk - i];
                uint32_t hi = (uint32_t)(acc >> 32);
                hi = BabyBear::umin(hi, hi - BabyBear::P);
                const uint64_t tt = ((uint64_t)hi << 32) | (uint32_t)acc;
                const uint32_t m = (uint32_t)tt * BabyBear::PINV;
                const uint32_t u = (uint32_t)((tt + (uint64_t)m * BabyBear::P) >> 32);
                const E r = BabyBear::umin(u, u - BabyBear::P);
                col = g == lo ? r : F::add(col, r);
            }
            t[kk] = col;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2 * W - 1; i++) t[i] = F::zero();
#pragma unroll
        for (int i = 0; i < W; i++)
#pragma unroll
            for (int j = 0; j < W; j++) t[i + j] = F::add(t[i + j], F::mul_boundary_pre(px[i], py[j]));
    }
    const E nr = sc_get<F>(k.R[1]);
#pragma unroll
    for (int i = 0; i < W - 1; i++) t[i] = F::add(t[i], F::mul_tw(t[i + W], nr));
#pragma unroll
    for (int m = 0; m < W; m++) x[m] = F::boundary_post(t[T::PERM[m]]);
}

// A workgroup is ONE wave and owns 64 consecutive ring elements.  A lane works on its own element, whose D words are
// D * 8 bytes apart from its neighbour's: reading them lane by lane would touch 64 cache lines per load instruction.  The
// block of 64 * D words is therefore moved between HBM and LDS with lane-contiguous accesses (512 B per instruction) and
// each lane picks its element out of LDS (element stride D + 1 words: conflict-free).  A ragged last block clamps its reads.
// The block goes through LDS in kRounds<D> rounds of 64 / kRounds elements (D = 72: two rounds, 9.3 KB per wave instead of 18.7,
// so that the register file and not the LDS sets the number of resident waves).
template <int D>
constexpr int small_rounds() { return D >= 24 ? 2 : 1; }
template <class F, int D>
__device__ __forceinline__ void small_block_load(typename F::elem *lds, const uint64_t *src, size_t first, size_t batch,
                                                 typename F::elem *x) {
    constexpr int R = small_rounds<D>(), PER = 64 / R;
    const int t = threadIdx.x;
    const size_t n_valid = (batch - first < 64 ? batch - first : 64) * D;
#pragma unroll
    for (int r = 0; r < R; r++) {
#pragma unroll 4
        for (int idx = t; idx < PER * D; idx += 64) {
            const int el = idx / D, i = idx - el * D;
            const size_t g = (size_t)r * PER * D + idx;
            lds[el * (D + 1) + i] = F::load(src + first * D + (g < n_valid ? g : 0));
        }
        __syncthreads();
        if (t / PER == r) {
#pragma unroll
            for (int i = 0; i < D; i++) x[i] = lds[(t - r * PER) * (D + 1) + i];
        }
        __syncthreads();
    }
}
template <class F, int D>
__device__ __forceinline__ void small_block_store(typename F::elem *lds, uint64_t *dst, size_t first, size_t batch,
                                                  const typename F::elem *x) {
    constexpr int R = small_rounds<D>(), PER = 64 / R;
    const int t = threadIdx.x;
    const size_t n_valid = (batch - first < 64 ? batch - first : 64) * D;
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (t / PER == r) {
#pragma unroll
            for (int i = 0; i < D; i++) lds[(t - r * PER) * (D + 1) + i] = x[i];
        }
        __syncthreads();
#pragma unroll 4
        for (int idx = t; idx < PER * D; idx += 64) {
            const int el = idx / D, i = idx - el * D;
            const size_t g = (size_t)r * PER * D + idx;
            if (g < n_valid) F::store(dst + first * D + g, lds[el * (D + 1) + i]);
        }
        if (r + 1 < R) __syncthreads();
    }
}

// STAGED: through LDS as above; otherwise each lane reads and writes its own element directly.  Which one wins was
// measured per ring and operation over 2^22 elements (tools/bench_small_rings.py): staging gains 25-70 % everywhere
// except Goldilocks-24's fused ring product (-13 %, 96 data VGPRs plus the exchanges), which stays direct (its ICRT was direct
// too until the block went through LDS in two rounds: staged 4.0 TB/s, direct 3.6-3.8).
template <class F, int D, int W, int OP>
constexpr bool small_staged() {
    return !(D == 24 && OP == SOP_RINGMUL);
}
template <class F, int D, bool STAGED>
__device__ __forceinline__ void small_get(typename F::elem *lds, const uint64_t *src, size_t first, size_t batch,
                                          typename F::elem *x) {
    if (STAGED) {
        small_block_load<F, D>(lds, src, first, batch, x);
    } else {
        const size_t e = first + threadIdx.x < batch ? first + threadIdx.x : batch - 1;
#pragma unroll
        for (int i = 0; i < D; i++) x[i] = F::load(src + e * D + i);
    }
}
template <class F, int D, int W, int OP>
__global__ __launch_bounds__(64, (small_staged<F, D, W, OP>() && D == 24 ? 4 : 3)) void small_ring_kernel(SmallRingConsts k, const uint64_t *a, const uint64_t *b,
                                                        uint64_t *out, size_t batch) {
    using E = typename F::elem;
    constexpr bool STAGED = small_staged<F, D, W, OP>();
    __shared__ E lds[STAGED ? (64 / small_rounds<D>()) * (D + 1) : 1];
    const size_t first = blockIdx.x * (size_t)64;
    E x[D];
    small_get<F, D, STAGED>(lds, a, first, batch, x);
    if (OP == SOP_CRT) {
        small_fwd3<F, D>(x, k);
        small_homogenize<F, W>(x, k);
    } else if (OP == SOP_ICRT) {
        small_dehomogenize<F, W>(x, k);
        small_inv3<F, D>(x, k);
    } else {
        E y[D];
        if (OP == SOP_MULB) {
#pragma unroll
            for (int i = 0; i < D; i++) y[i] = F::load(b + i);
        } else {
            small_get<F, D, STAGED>(lds, b, first, batch, y);
        }
        if (OP == SOP_RINGMUL) {
            small_fwd3<F, D>(x, k);
            small_homogenize<F, W>(x, k);
            small_fwd3<F, D>(y, k);
            small_homogenize<F, W>(y, k);
        }
#pragma unroll
        for (int s = 0; s < 8; s++) small_slot_mul<F, W>(x + s * W, y + s * W, k);
        if (OP == SOP_RINGMUL) {
            small_dehomogenize<F, W>(x, k);
            small_inv3<F, D>(x, k);
        }
    }
    if (STAGED) {
        small_block_store<F, D>(lds, out, first, batch, x);
    } else if (first + threadIdx.x < batch) {
        const size_t e = first + threadIdx.x;
#pragma unroll
        for (int i = 0; i < D; i++) F::store(out + e * D + i, x[i]);
    }
}

// goldilocks/mod.rs:75-98, babybear/mod.rs:87-110
template <class F, int D>
__global__ void small_reduce_kernel(const uint64_t *in, size_t in_len, uint64_t *out, size_t batch) {
    size_t n = batch * D;
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        size_t e = t / D;
        int i = (int)(t % D);
        const uint64_t *src = in + e * in_len;
        auto get = [&](size_t j) { return j < in_len ? F::load(src + j) : F::zero(); };
        typename F::elem v;
        if (i < D / 2)
            v = F::sub(F::sub(get(i), get(D + i)), get(D + D / 2 + i));
        else
            v = F::add(get(i), get(D / 2 + i));
        F::store(out + t, v);
    }
}

template <class F>
inline void small_consts_for(SmallRingConsts &c) {
    using E = typename F::elem;
    uint64_t p = (uint64_t)F::P;
    uint64_t e[1] = {(p - 1) / 24};
    E w = pow_tw<F>(F::tw_from_u64(F::kGenerator), e, 1);
    E r = F::tw_one();
    E roots[24];
    for (int i = 0; i < 24; i++) {
        roots[i] = r;
        c.R[i] = (uint64_t)r;
        r = F::mul_tw(r, w);
    }
    uint64_t pm2[1] = {p - 2};
    E two_z_m1 = F::sub(F::add(roots[4], roots[4]), F::tw_one());
    c.kappa = (uint64_t)pow_tw<F>(two_z_m1, pm2, 1);
    c.inv8 = (uint64_t)pow_tw<F>(F::tw_from_u64(8), pm2, 1);
    c.inv4 = (uint64_t)pow_tw<F>(F::tw_from_u64(4), pm2, 1);
}
inline int small_init(SmallRingConsts &c, bool goldilocks) {
    c.is_goldilocks = goldilocks ? 1 : 0;
    if (goldilocks)
        small_consts_for<Goldilocks>(c);
    else
        small_consts_for<BabyBear>(c);
    return 0;
}
inline void small_destroy(SmallRingConsts &) {}

template <class F, int D, int W>
inline int small_dispatch(const SmallRingConsts &c, int op, const uint64_t *a, const uint64_t *b, size_t in_len,
                          uint64_t *out, size_t batch, hipStream_t st) {
    if (batch == 0) return 0;
    if (op == 4) {
        size_t n = batch * D, blocks = (n + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL((small_reduce_kernel<F, D>), dim3((unsigned)blocks), dim3(256), 0, st, a, in_len, out, batch);
        return hipGetLastError() != hipSuccess;
    }
    size_t blocks = (batch + 63) / 64;
    if (blocks > 0x7FFFFFFFull) return 1;
    dim3 g((unsigned)blocks), t(64);
    switch (op) {
        case SOP_CRT: hipLaunchKernelGGL((small_ring_kernel<F, D, W, SOP_CRT>), g, t, 0, st, c, a, b, out, batch); break;
        case SOP_ICRT: hipLaunchKernelGGL((small_ring_kernel<F, D, W, SOP_ICRT>), g, t, 0, st, c, a, b, out, batch); break;
        case SOP_MUL: hipLaunchKernelGGL((small_ring_kernel<F, D, W, SOP_MUL>), g, t, 0, st, c, a, b, out, batch); break;
        case SOP_RINGMUL: hipLaunchKernelGGL((small_ring_kernel<F, D, W, SOP_RINGMUL>), g, t, 0, st, c, a, b, out, batch); break;
        case SOP_MULB: hipLaunchKernelGGL((small_ring_kernel<F, D, W, SOP_MULB>), g, t, 0, st, c, a, b, out, batch); break;
        default: return 1;
    }
    return hipGetLastError() != hipSuccess;
}
// op is one of the SMALL_* ids; returns non-zero on launch failure
inline int small_launch(const SmallRingConsts &c, int op, const uint64_t *a, const uint64_t *b, size_t in_len,
                        uint64_t *out, size_t batch, hipStream_t st) {
    if (op <= SMALL_G24_REDUCE) return small_dispatch<Goldilocks, 24, 3>(c, op - SMALL_G24_CRT, a, b, in_len, out, batch, st);
    return small_dispatch<BabyBear, 72, 9>(c, op - SMALL_B72_CRT, a, b, in_len, out, batch, st);
}
// a[e] *= b[0] slot-wise for every element e of the batch (Matrix<R> *= &R, matrix.rs:207-211)
inline int small_launch_mul_bcast(const SmallRingConsts &c, bool b72, uint64_t *a, const uint64_t *b, size_t batch, hipStream_t st) {
    return b72 ? small_dispatch<BabyBear, 72, 9>(c, SOP_MULB, a, b, 0, a, batch, st) : small_dispatch<Goldilocks, 24, 3>(c, SOP_MULB, a, b, 0, a, batch, st);
}

}  // namespace sr
