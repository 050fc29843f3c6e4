// Tuned negacyclic transforms for the Stark rings Fp[X]/(X^D+1), D = 2^k, 9 <= k <= 20, on StarkL arithmetic (stark_lazy.hpp).
// Same algorithm, twiddle tables and slot order as the generic kernels (ntt_generic.hpp; reference
// crates/ring/src/cyclotomic_ring/models/stark_prime/ntt.rs:121-235 forward, :245-346 inverse, generalised to 2^k): forward
// Cooley-Tukey stages s = 0..k-1 with w = tw[2^s + block], inverse Gentleman-Sande stages k-1..0 with D^-1 in stage 0.
//
// What is different is where the data lives between stages: a lane keeps several coefficients in registers and runs two or
// three stages on them before anything is exchanged (the table indices of consecutive stages are 2 t0 + {0, 1}, 4 t0 + {0..3}):
//   rows512_kernel  the last nine stages: a 128-lane workgroup owns a tile of 512 consecutive coefficients, FOUR per lane (36
//                   VGPRs) -- five register passes (2, 2, 2, 2, 1 stages) with four LDS transposes in between.  MODE_MUL keeps
//                   fwd(a) in registers while b goes through the same LDS tile, multiplies the slots in registers and runs the
//                   inverse passes in the mirrored order; global loads and stores use the lane-contiguous layout (coefficient
//                   t + 128 j); the first pass' twiddles are wave-uniform.  Four per lane, not eight: with eight the kernel
//                   needed 256+ VGPRs, one wave per SIMD, and lost more to latency than the saved transposes gained.
//   cols_kernel<M>  the first k - 9 stages, M <= 3 at a time (2^M coefficients per lane), straight from and to global memory
//                   (legs 2^(k - s0 - M) >= 512 coefficients apart: every leg is a lane-contiguous 32-byte stream), no LDS.
// Lazy-carry bookkeeping (stark_lazy.hpp): a forward stage adds at most 2^28 per limb, so one weak reduction after the sixth
// rows stage keeps every limb below 2^31; an inverse group reduces its sum legs weakly at its end (a three-stage group also
// relaxes its twice-summed legs before the third stage).  Loads take canonical memory images, stores canonicalise.
#pragma once
#include "ntt_generic.hpp"
#include "stark_lazy.hpp"

namespace sr {
namespace st {

using F = StarkL;
using E = S9;
using S = U256Storage;
using P = NttParams<StarkL>;

constexpr int kTileLog = 9, kTile = 512;

__device__ __forceinline__ void ct(E &u, E &v, const E &w) {
    const E t = F::mul_tw(v, w);
    v = F::sub(u, t);
    u = F::add(u, t);
}
__device__ __forceinline__ void gs(E &u, E &v, const E &w) {
    const E d = F::sub(u, v);
    u = F::add(u, v);
    v = F::mul_tw(d, w);
}

// M forward stages on 2^M register legs; stage 0 of the group pairs legs 2^(M-1) apart.  t0 = table index of the first stage's
// block (2^s0 + block); the following stages use 2 t0 + {0, 1} and 4 t0 + {0..3}.
template <int M>
__device__ __forceinline__ void fwd_group(E *x, const E *tw, uint32_t t0) {
    if constexpr (M == 3) {
        const E w = tw[t0];
#pragma unroll
        for (int j = 0; j < 4; j++) ct(x[j], x[j + 4], w);
        t0 *= 2;
    }
    if constexpr (M >= 2) {
        constexpr int G = M == 3 ? 2 : 1;  // groups of four legs
#pragma unroll
        for (int g = 0; g < G; g++) {
            const E w = tw[t0 + g];
#pragma unroll
            for (int j = 0; j < 2; j++) ct(x[4 * g + j], x[4 * g + j + 2], w);
        }
        t0 *= 2;
    }
    constexpr int G2 = 1 << (M - 1);
#pragma unroll
    for (int g = 0; g < G2; g++) ct(x[2 * g], x[2 * g + 1], tw[t0 + g]);
}

// The mirrored inverse stages.  LAST: the group ends with stage 0 of the whole transform, whose legs are scaled by
// scale0 / scale1 (D^-1 folded in) instead of a twiddle.
template <int M, bool LAST>
__device__ __forceinline__ void inv_group(E *x, const P &p, uint32_t t0) {
    constexpr int G2 = 1 << (M - 1);
    const uint32_t t_last = t0 << (M - 1);
    if constexpr (M == 1 && LAST) {
        const E d = F::sub(x[0], x[1]);
        x[0] = F::mul_tw(F::add(x[0], x[1]), p.scale0);
        x[1] = F::mul_tw(d, p.scale1);
        return;
    }
#pragma unroll
    for (int g = 0; g < G2; g++) gs(x[2 * g], x[2 * g + 1], p.itw[t_last + g]);
    if constexpr (M >= 2) {
        constexpr int G = M == 3 ? 2 : 1;
        const uint32_t t_mid = t0 << (M - 2);
#pragma unroll
        for (int g = 0; g < G; g++) {
            if constexpr (M == 2 && LAST) {
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    const E d = F::sub(x[j], x[j + 2]);
                    x[j] = F::mul_tw(F::add(x[j], x[j + 2]), p.scale0);
                    x[j + 2] = F::mul_tw(d, p.scale1);
                }
            } else {
                const E w = p.itw[t_mid + g];
#pragma unroll
                for (int j = 0; j < 2; j++) gs(x[4 * g + j], x[4 * g + j + 2], w);
            }
        }
    }
    if constexpr (M == 3) {
        // legs 0 and 4 have been summed twice (limbs up to 4 * 2^28): relax them before the third sum
        x[0] = F::relax(x[0]);
        x[4] = F::relax(x[4]);
        if constexpr (LAST) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const E d = F::sub(x[j], x[j + 4]);
                x[j] = F::mul_tw(F::add(x[j], x[j + 4]), p.scale0);
                x[j + 4] = F::mul_tw(d, p.scale1);
            }
        } else {
            const E w = p.itw[t0];
#pragma unroll
            for (int j = 0; j < 4; j++) gs(x[j], x[j + 4], w);
        }
    }
    if constexpr (!LAST) {  // the legs that end as sums leave weakly reduced (the others are fresh products)
#pragma unroll
        for (int j = 0; j < (1 << (M - 1)); j++) x[j] = F::weak_reduce(x[j]);
    }
}

// ---- strided passes: stages [s0, s0 + M) on legs D >> (s0 + M) apart, straight from / to global memory ----------------------
// one lane per (element, block of stage s0, offset inside the leg): grid.x * 256 >= batch << (k - M)
// LAST (inverse only): the pass ends with stage 0 of the transform (s0 == 0)
template <int M, int DIR, bool LAST>
__global__ __launch_bounds__(256, 2) void cols_kernel(S *data, size_t batch, int s0, P p) {
    const size_t gid = blockIdx.x * (size_t)256 + threadIdx.x;
    const int lq = p.k - M;                // log2 lanes per element
    if (gid >= (batch << lq)) return;
    const size_t poly = gid >> lq;
    const uint32_t q = (uint32_t)(gid & (((size_t)1 << lq) - 1));
    const int ls = p.k - s0 - M;           // log2 leg stride
    const uint32_t blk = q >> ls, r = q & ((1u << ls) - 1u);
    S *base = data + (poly << p.k) + ((size_t)blk << (ls + M)) + r;
    E x[1 << M];
#pragma unroll
    for (int j = 0; j < (1 << M); j++) x[j] = F::load(base + ((size_t)j << ls));
    const uint32_t t0 = (1u << s0) + blk;
    if constexpr (DIR == MODE_FWD) {
        fwd_group<M>(x, p.tw, t0);
    } else {
        inv_group<M, LAST>(x, p, t0);
    }
#pragma unroll
    for (int j = 0; j < (1 << M); j++) F::store(base + ((size_t)j << ls), x[j]);
}

// ---- the last nine stages: a 128-lane workgroup (two waves) per 512-coefficient tile, four coefficients per lane ------------
// register layouts of lane t, leg j (two stages per pass pair legs 2 and 1 apart; the fifth pass is a single stage):
//   L1: e = t + 128 j                      halves 256, 128   block = tile                (wave-uniform twiddles)
//   L2: e = 128 (t >> 5) + 32 j + (t & 31) halves  64,  32   block = t >> 5
//   L3: e =  32 (t >> 3) +  8 j + (t & 7)  halves  16,   8   block = t >> 3
//   L4: e =   8 (t >> 1) +  2 j + (t & 1)  halves   4,   2   block = t >> 1
//   L5: e = 4 t + j                        half     1        blocks 2 t, 2 t + 1
// LDS: limb-major rows of pad(511) + 1 words, pad(e) = e + 5 (e >> 5): conflict-free for L1, L2, L5, two-way for L3, L4
// (tools search in DESIGN.md 5.3).
constexpr int kLanes = 128;
__device__ __forceinline__ int pad(int e) { return e + 5 * (e >> 5); }
constexpr int kLdsRow = 511 + 5 * 15 + 1;
constexpr int kLdsWords = 9 * kLdsRow;
template <int L>
__device__ __forceinline__ int pos(int t, int j) {
    if constexpr (L == 1) return t + 128 * j;
    else if constexpr (L == 2) return ((t >> 5) << 7) + 32 * j + (t & 31);
    else if constexpr (L == 3) return ((t >> 3) << 5) + 8 * j + (t & 7);
    else if constexpr (L == 4) return ((t >> 1) << 3) + 2 * j + (t & 1);
    else return 4 * t + j;
}
template <int FROM, int TO>
__device__ __forceinline__ void exchange(E *x, uint32_t *lds, int t) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = pad(pos<FROM>(t, j));
#pragma unroll
        for (int i = 0; i < 9; i++) lds[i * kLdsRow + a] = (uint32_t)x[j].l[i];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int a = pad(pos<TO>(t, j));
#pragma unroll
        for (int i = 0; i < 9; i++) x[j].l[i] = (int32_t)lds[i * kLdsRow + a];
    }
    __syncthreads();
}

// forward stages k-9 .. k-1 of one tile: canonical memory images in (layout L1), L5 registers out.  tix = 2^(k-9) + tile index
// inside its ring element = the table index of the tile's block at stage k-9.
__device__ __forceinline__ void tile_fwd(const S *src, E *x, uint32_t *lds, int t, const P &p, uint32_t tix) {
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = F::load(src + pos<1>(t, j));
    fwd_group<2>(x, p.tw, tix);
    exchange<1, 2>(x, lds, t);
    fwd_group<2>(x, p.tw, tix * 4 + (t >> 5));
    exchange<2, 3>(x, lds, t);
    fwd_group<2>(x, p.tw, tix * 16 + (t >> 3));
#pragma unroll
    for (int j = 0; j < 4; j++) x[j] = F::weak_reduce(x[j]);  // six stages of uncarried sums: limbs up to 7 * 2^28
    exchange<3, 4>(x, lds, t);
    fwd_group<2>(x, p.tw, tix * 64 + (t >> 1));
    exchange<4, 5>(x, lds, t);
    ct(x[0], x[1], p.tw[tix * 256 + 2 * t]);
    ct(x[2], x[3], p.tw[tix * 256 + 2 * t + 1]);
}
// inverse stages k-1 .. k-9: L5 registers in, canonical memory images out (layout L1); for k == 9 the last group ends with
// stage 0 of the transform
__device__ __forceinline__ void tile_inv(E *x, S *dst, uint32_t *lds, int t, const P &p, uint32_t tix) {
    gs(x[0], x[1], p.itw[tix * 256 + 2 * t]);
    gs(x[2], x[3], p.itw[tix * 256 + 2 * t + 1]);
    x[0] = F::weak_reduce(x[0]);
    x[2] = F::weak_reduce(x[2]);
    exchange<5, 4>(x, lds, t);
    inv_group<2, false>(x, p, tix * 64 + (t >> 1));
    exchange<4, 3>(x, lds, t);
    inv_group<2, false>(x, p, tix * 16 + (t >> 3));
    exchange<3, 2>(x, lds, t);
    inv_group<2, false>(x, p, tix * 4 + (t >> 5));
    exchange<2, 1>(x, lds, t);
    if (p.k == kTileLog) inv_group<2, true>(x, p, tix);
    else inv_group<2, false>(x, p, tix);
#pragma unroll
    for (int j = 0; j < 4; j++) F::store(dst + pos<1>(t, j), x[j]);
}

// grid.x = batch * D / 512, 128 lanes.  a, b, out: flat batches (out may be a).
#ifndef SR_ST_WAVES
#define SR_ST_WAVES 3  /* measured at D = 2^12, batch 2^12: 1.72 ms with 2 waves per SIMD, 1.64 with 3, 1.79 with 4 */
#endif
template <int MODE>
__global__ __launch_bounds__(kLanes, SR_ST_WAVES) void rows512_kernel(S *a, const S *b, S *out, P p) {
    __shared__ uint32_t lds[kLdsWords];
    const int t = threadIdx.x;
    const size_t tile = blockIdx.x;
    const size_t off = tile << kTileLog;
    const uint32_t tix = (1u << (p.k - kTileLog)) + (uint32_t)(tile & (((size_t)1 << (p.k - kTileLog)) - 1));
    E x[4];
    if constexpr (MODE == MODE_FWD) {
        tile_fwd(a + off, x, lds, t, p, tix);
        exchange<5, 1>(x, lds, t);
#pragma unroll
        for (int j = 0; j < 4; j++) F::store(out + off + pos<1>(t, j), x[j]);
    } else if constexpr (MODE == MODE_INV) {
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = F::load(a + off + pos<1>(t, j));
        exchange<1, 5>(x, lds, t);
        tile_inv(x, out + off, lds, t, p, tix);
    } else {
        E y[4];
        tile_fwd(a + off, y, lds, t, p, tix);
        tile_fwd(b + off, x, lds, t, p, tix);
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = F::mul_data(x[j], y[j]);
        tile_inv(x, out + off, lds, t, p, tix);
    }
}

inline bool supported(int k) { return k >= kTileLog && k <= 20; }

template <int DIR>
inline int launch_cols(S *d, size_t batch, int s0, int m, const P &p, hipStream_t st) {
    const size_t lanes = batch << (p.k - m);
    const size_t blocks = (lanes + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return 1;
    const dim3 g((unsigned)blocks), b(256);
    const bool last = DIR == MODE_INV && s0 == 0;
    if (m == 3) {
        if (last) hipLaunchKernelGGL((cols_kernel<3, DIR, true>), g, b, 0, st, d, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<3, DIR, false>), g, b, 0, st, d, batch, s0, p);
    } else if (m == 2) {
        if (last) hipLaunchKernelGGL((cols_kernel<2, DIR, true>), g, b, 0, st, d, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<2, DIR, false>), g, b, 0, st, d, batch, s0, p);
    } else {
        if (last) hipLaunchKernelGGL((cols_kernel<1, DIR, true>), g, b, 0, st, d, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<1, DIR, false>), g, b, 0, st, d, batch, s0, p);
    }
    return hipGetLastError() != hipSuccess;
}
// the strided stages 0 .. k-10 as passes of (c mod 3), 3, 3, ... stages; begin / end bracket each launch for the profiler
template <class Hook>
inline int fwd_cols(S *d, size_t batch, const P &p, hipStream_t st, Hook &&hook) {
    int c = p.k - kTileLog, s0 = 0;
    while (c > 0) {
        const int m = c % 3 ? c % 3 : 3;
        hook(true);
        const int rc = launch_cols<MODE_FWD>(d, batch, s0, m, p, st);
        hook(false);
        if (rc) return rc;
        s0 += m;
        c -= m;
    }
    return 0;
}
template <class Hook>
inline int inv_cols(S *d, size_t batch, const P &p, hipStream_t st, Hook &&hook) {
    const int c = p.k - kTileLog;
    int s_hi = c;  // stages [0, s_hi) remain
    // mirror of fwd_cols: its passes were (c mod 3 or 3), 3, 3, ...; undo them last to first
    while (s_hi > 0) {
        const int first = c % 3 ? c % 3 : 3;
        const int m = s_hi > first ? 3 : first;
        hook(true);
        const int rc = launch_cols<MODE_INV>(d, batch, s_hi - m, m, p, st);
        hook(false);
        if (rc) return rc;
        s_hi -= m;
    }
    return 0;
}
template <int MODE>
inline int launch_rows(S *a, const S *b, S *out, size_t batch, const P &p, hipStream_t st) {
    const size_t tiles = batch << (p.k - kTileLog);
    if (tiles > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((rows512_kernel<MODE>), dim3((unsigned)tiles), dim3(kLanes), 0, st, a, b, out, p);
    return hipGetLastError() != hipSuccess;
}

}  // namespace st
}  // namespace sr
