// Tuned negacyclic transforms for the Stark rings Fp[X]/(X^D+1), D = 2^k, 4 <= k <= 20, on StarkL arithmetic (stark_lazy.hpp).
// Same algorithm, twiddle tables and slot order as the generic kernels (ntt_generic.hpp; reference
// crates/ring/src/cyclotomic_ring/models/stark_prime/ntt.rs:121-235 forward, :245-346 inverse, generalised to 2^k): forward
// Cooley-Tukey stages s = 0..k-1 with w = tw[2^s + block], inverse Gentleman-Sande stages k-1..0 with D^-1 in stage 0.
//
// What is different is where the data lives between stages: a lane keeps several coefficients in registers and runs two or
// three stages on them before anything is exchanged (the table indices of consecutive stages are 2 t0 + {0, 1}, 4 t0 + {0..3}):
//   tile_kernel     the last LOGT stages: a workgroup of 2^(LOGT-2) lanes owns a tile of 2^LOGT consecutive coefficients, FOUR per
//                   lane (36 VGPRs) -- register passes of two stages with LDS transposes in between.  MODE_MUL keeps fwd(a) in
//                   registers while b goes through the same LDS tile, multiplies the slots in registers and runs the inverse
//                   passes in the mirrored order; global loads and stores use the lane-contiguous layout; the first pass'
//                   twiddles are wave-uniform.  For 512 <= D <= 4096 the tile is the ring element (LOGT = k): the whole ring
//                   product is one launch and one trip over HBM.  Above, LOGT = 9 after the strided passes.  Four per lane, not
//                   eight: with eight the kernel needed 256+ VGPRs, one wave per SIMD, and lost more to latency than the saved
//                   transposes gained.
//   cols_kernel<M>  the first k - 9 stages, M <= 3 at a time (2^M coefficients per lane), straight from and to global memory
//                   (legs 2^(k - s0 - M) >= 512 coefficients apart: every leg is a lane-contiguous 32-byte stream), no LDS.
// Lazy-carry bookkeeping (stark_lazy.hpp): a forward stage adds at most 2^28 per limb, so one weak reduction after the sixth
// rows stage keeps every limb below 2^31; an inverse group reduces its sum legs weakly at its end (a three-stage group also
// relaxes its twice-summed legs before the third stage).  Loads take canonical memory images, stores canonicalise.
#pragma once
#include <atomic>
#include <cstdlib>

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
__global__ __launch_bounds__(256, 2) void cols_kernel(S *data, const S *src, size_t batch, int s0, P p) {
    const size_t gid = blockIdx.x * (size_t)256 + threadIdx.x;
    const int lq = p.k - M;                // log2 lanes per element
    if (gid >= (batch << lq)) return;
    const size_t poly = gid >> lq;
    const uint32_t q = (uint32_t)(gid & (((size_t)1 << lq) - 1));
    const int ls = p.k - s0 - M;           // log2 leg stride
    const uint32_t blk = q >> ls, r = q & ((1u << ls) - 1u);
    const size_t off = (poly << p.k) + ((size_t)blk << (ls + M)) + r;
    S *base = data + off;
    const S *sbase = src + off;  // src == data: in place; otherwise the pass reads src and leaves it intact
    E x[1 << M];
#pragma unroll
    for (int j = 0; j < (1 << M); j++) x[j] = F::load(sbase + ((size_t)j << ls));
    const uint32_t t0 = (1u << s0) + blk;
    if constexpr (DIR == MODE_FWD) {
        fwd_group<M>(x, p.tw, t0);
    } else {
        inv_group<M, LAST>(x, p, t0);
    }
#pragma unroll
    for (int j = 0; j < (1 << M); j++) F::store(base + ((size_t)j << ls), x[j]);
}

// ---- the last LOGT stages: a workgroup of 2^(LOGT-2) lanes per tile of 2^LOGT consecutive coefficients, four per lane ---------
// LOGT = 9 for D > 4096 (after the strided passes); LOGT = k for 512 <= D <= 4096: the whole transform -- and in MODE_MUL the
// whole ring product -- is ONE launch and one trip over HBM (D = 4096: 1024 lanes, 157 KB of the CU's 160 KB LDS).
// Pass q = 0 .. LOGT/2 - 1 runs stages 2q, 2q + 1 of the tile on the register layout
//     e = ((t >> ls) << (ls + 2)) + (j << ls) + (t & (2^ls - 1)),   ls = LOGT - 2q - 2      (legs 2^ls apart, block t >> ls)
// and an odd LOGT ends with the single stage of half 1 on e = 4 t + j (blocks 2 t, 2 t + 1).  Pass 0's twiddles are uniform.
// LDS: limb-major rows, pad(e) = e + PM (e >> 5): conflict-free or two-way for every layout (PM = 5 at LOGT = 9, else 3;
// tools/stark_lds_padding_search.py).
// TPW tiles share a workgroup (and an LDS row) when a tile has fewer than 128 lanes: D = 16 .. 256, 256 lanes per workgroup;
// pad() is additive over tile bases (multiples of 16), so a tile addresses its slice through an offset pointer.
template <int LOGT, int TPW = 1>
struct Tile {
    static constexpr int kLanes = 1 << (LOGT - 2);   // per tile
    static constexpr int kPM = LOGT == 9 ? 5 : 3;
    static constexpr int kRow = ((TPW << LOGT) - 1) + kPM * (((TPW << LOGT) - 1) >> 5) + 1;
    static constexpr int kWords = 9 * kRow;
    static constexpr int kPasses = LOGT / 2;     // two-stage passes
    static constexpr bool kOdd = LOGT & 1;
    static constexpr int kLayouts = kPasses + (kOdd ? 1 : 0);
    __device__ static __forceinline__ int pad(int e) { return e + kPM * (e >> 5); }
    template <int Q>
    __device__ static __forceinline__ int pos(int t, int j) {
        if constexpr (Q >= kPasses) {
            return 4 * t + j;
        } else {
            constexpr int ls = LOGT - 2 * Q - 2;
            return ((t >> ls) << (ls + 2)) + (j << ls) + (t & ((1 << ls) - 1));
        }
    }
    template <int Q>
    __device__ static __forceinline__ uint32_t block(int t) {
        constexpr int ls = LOGT - 2 * Q - 2;
        return (uint32_t)(t >> ls);
    }
    template <int FROM, int TO>
    __device__ static __forceinline__ void exchange(E *x, uint32_t *lds, int t) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int a = pad(pos<FROM>(t, j));
#pragma unroll
            for (int i = 0; i < 9; i++) lds[i * kRow + a] = (uint32_t)x[j].l[i];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int a = pad(pos<TO>(t, j));
#pragma unroll
            for (int i = 0; i < 9; i++) x[j].l[i] = (int32_t)lds[i * kRow + a];
        }
        __syncthreads();
    }
    // forward passes Q .. : registers in layout Q -> registers in the last layout.  tix = table index of the tile's block at its
    // first stage (2^(k - LOGT) + tile index inside the ring element).  Every third pass ends with a weak reduction (six
    // stages of uncarried sums reach 7 * 2^28 per limb).
    template <int Q>
    __device__ static __forceinline__ void fwd_from(E *x, uint32_t *lds, int t, const P &p, uint32_t tix) {
        if constexpr (Q < kPasses) {
            fwd_group<2>(x, p.tw, (tix << (2 * Q)) + block<Q>(t));
            if constexpr (Q % 3 == 2 && Q + 1 < kLayouts) {
#pragma unroll
                for (int j = 0; j < 4; j++) x[j] = F::weak_reduce(x[j]);
            }
            if constexpr (Q + 1 < kLayouts) {
                exchange<Q, Q + 1>(x, lds, t);
                fwd_from<Q + 1>(x, lds, t, p, tix);
            }
        } else {  // the single last stage of an odd LOGT
            ct(x[0], x[1], p.tw[(tix << (LOGT - 1)) + 2 * t]);
            ct(x[2], x[3], p.tw[(tix << (LOGT - 1)) + 2 * t + 1]);
        }
    }
    // inverse passes Q .. 0: registers in layout Q -> registers in layout 0 after the tile's first stage.  WHOLE: the tile's
    // first stage is stage 0 of the transform (scaled legs).
    template <int Q, bool WHOLE>
    __device__ static __forceinline__ void inv_from(E *x, uint32_t *lds, int t, const P &p, uint32_t tix) {
        if constexpr (Q >= kPasses) {
            gs(x[0], x[1], p.itw[(tix << (LOGT - 1)) + 2 * t]);
            gs(x[2], x[3], p.itw[(tix << (LOGT - 1)) + 2 * t + 1]);
            x[0] = F::weak_reduce(x[0]);
            x[2] = F::weak_reduce(x[2]);
        } else if constexpr (Q == 0) {
            inv_group<2, WHOLE>(x, p, tix);
        } else {
            inv_group<2, false>(x, p, (tix << (2 * Q)) + block<Q>(t));
        }
        if constexpr (Q > 0) {
            exchange<Q, Q - 1>(x, lds, t);
            inv_from<Q - 1, WHOLE>(x, lds, t, p, tix);
        }
    }
    __device__ static __forceinline__ void load(const S *src, E *x, int t) {
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = F::load(src + pos<0>(t, j));
    }
    __device__ static __forceinline__ void store(S *dst, const E *x, int t) {
#pragma unroll
        for (int j = 0; j < 4; j++) F::store(dst + pos<0>(t, j), x[j]);
    }
};

constexpr int kTileWaves = 3;  // waves per SIMD of the tiles up to 1024 points; 512-tiles at D = 2^12, batch 2^12: 1.72 / 1.64 / 1.79 ms with 2 / 3 / 4
// D = 1024 tiles (256 lanes, 40 KB): 3 waves per SIMD without spills beat 4 with (1.84 against 1.96 ms per 2^24 coefficients);
// D = 2048 (512 lanes, 80 KB) needs the 128-register budget for its second workgroup per CU (2.13 against 2.55 ms).
constexpr int tiles_per_wg(int logt) { return logt >= 9 ? 1 : 256 >> (logt - 2); }
// grid.x = ceil(n_tiles / TPW), n_tiles = batch * D / 2^LOGT.  a, b, out: flat batches (out may be a).  WHOLE: LOGT == k.
// Lanes of a tile past the end (ragged last workgroup, TPW > 1) compute on zeros and store nothing: every lane reaches every
// barrier.
template <int LOGT, int MODE, bool WHOLE>
__global__ __launch_bounds__((1 << (LOGT - 2)) * tiles_per_wg(LOGT), (LOGT <= 10 ? kTileWaves : 4)) void tile_kernel(S *a, const S *b, S *out,
                                                                                                                   size_t n_tiles, P p) {
    constexpr int TPW = tiles_per_wg(LOGT);
    using T = Tile<LOGT, TPW>;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_all[];
    const int t = threadIdx.x & (T::kLanes - 1);
    const int tile_in_wg = threadIdx.x >> (LOGT - 2);
    uint32_t *lds = lds_all + T::pad(tile_in_wg << LOGT);
    const size_t tile = blockIdx.x * (size_t)TPW + tile_in_wg;
    const bool live = TPW == 1 || tile < n_tiles;
    const size_t off = (live ? tile : 0) << LOGT;
    const uint32_t tix = WHOLE ? 1u : (1u << (p.k - LOGT)) + (uint32_t)(tile & (((size_t)1 << (p.k - LOGT)) - 1));
    constexpr int LAST = T::kLayouts - 1;
    E x[4];
    if constexpr (MODE == MODE_FWD) {
        T::load(a + off, x, t);
        T::template fwd_from<0>(x, lds, t, p, tix);
        T::template exchange<LAST, 0>(x, lds, t);
        if (live) T::store(out + off, x, t);
    } else if constexpr (MODE == MODE_INV) {
        T::load(a + off, x, t);
        T::template exchange<0, LAST>(x, lds, t);
        T::template inv_from<LAST, WHOLE>(x, lds, t, p, tix);
        if (live) T::store(out + off, x, t);
    } else {
        E y[4];
        T::load(a + off, y, t);
        T::template fwd_from<0>(y, lds, t, p, tix);
        T::load(b + off, x, t);
        T::template fwd_from<0>(x, lds, t, p, tix);
#pragma unroll
        for (int j = 0; j < 4; j++) x[j] = F::mul_data(x[j], y[j]);
        T::template inv_from<LAST, WHOLE>(x, lds, t, p, tix);
        if (live) T::store(out + off, x, t);
    }
}

constexpr int kMinLog = 4;  // D = 16, the reference's own Stark ring (stark_prime/mod.rs:34-68), is the smallest tile
inline bool supported(int k) { return k >= kMinLog && k <= 20; }
// 512 <= D <= 4096: the tile kernel takes the whole transform; above, the last nine stages after the strided passes
// Largest log2 D taken as one tile.  Measured over 2^24 coefficients (ring product, ms): D = 1024 one tile 1.96 / strided + 512-tiles
// 2.48; D = 2048 2.22 / 2.46; D = 4096 2.86 / 2.57 (1024 lanes and 157 KB of LDS leave one workgroup per CU and every transpose
// stalls all sixteen waves).  SR_ST_WHOLE_MAX overrides (9..12); the context reads it once, when it is created.
inline int whole_max(int requested) {  // sr_plan.stark_whole_max: 0 = default
    const int x = requested ? requested : 11;
    return x < 9 ? 9 : (x > 12 ? 12 : x);
}
inline bool whole(int k, int wmax) { return k >= kMinLog && k <= wmax; }

template <int DIR>
inline int launch_cols(S *d, const S *src, size_t batch, int s0, int m, const P &p, hipStream_t st) {
    const size_t lanes = batch << (p.k - m);
    const size_t blocks = (lanes + 255) / 256;
    if (blocks > 0x7FFFFFFFull) return 1;
    const dim3 g((unsigned)blocks), b(256);
    const bool last = DIR == MODE_INV && s0 == 0;
    if (m == 3) {
        if (last) hipLaunchKernelGGL((cols_kernel<3, DIR, true>), g, b, 0, st, d, src, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<3, DIR, false>), g, b, 0, st, d, src, batch, s0, p);
    } else if (m == 2) {
        if (last) hipLaunchKernelGGL((cols_kernel<2, DIR, true>), g, b, 0, st, d, src, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<2, DIR, false>), g, b, 0, st, d, src, batch, s0, p);
    } else {
        if (last) hipLaunchKernelGGL((cols_kernel<1, DIR, true>), g, b, 0, st, d, src, batch, s0, p);
        else hipLaunchKernelGGL((cols_kernel<1, DIR, false>), g, b, 0, st, d, src, batch, s0, p);
    }
    return hipGetLastError() != hipSuccess;
}
// the strided stages 0 .. k-10 as passes of (c mod 3), 3, 3, ... stages; begin / end bracket each launch for the profiler
template <class Hook>
inline int fwd_cols(S *d, const S *src, size_t batch, const P &p, bool one_tile, hipStream_t st, Hook &&hook) {
    int c = one_tile ? 0 : p.k - kTileLog, s0 = 0;
    while (c > 0) {
        const int m = c % 3 ? c % 3 : 3;
        hook(true);
        const int rc = launch_cols<MODE_FWD>(d, s0 == 0 ? src : d, batch, s0, m, p, st);  // only the first pass reads src
        hook(false);
        if (rc) return rc;
        s0 += m;
        c -= m;
    }
    return 0;
}
template <class Hook>
inline int inv_cols(S *d, size_t batch, const P &p, bool one_tile, hipStream_t st, Hook &&hook) {
    const int c = one_tile ? 0 : p.k - kTileLog;
    int s_hi = c;  // stages [0, s_hi) remain
    // mirror of fwd_cols: its passes were (c mod 3 or 3), 3, 3, ...; undo them last to first
    while (s_hi > 0) {
        const int first = c % 3 ? c % 3 : 3;
        const int m = s_hi > first ? 3 : first;
        hook(true);
        const int rc = launch_cols<MODE_INV>(d, d, batch, s_hi - m, m, p, st);
        hook(false);
        if (rc) return rc;
        s_hi -= m;
    }
    return 0;
}
template <int LOGT, int MODE, bool WHOLE>
inline int launch_tile(S *a, const S *b, S *out, size_t tiles, const P &p, hipStream_t st) {
    using T = Tile<LOGT, tiles_per_wg(LOGT)>;
    constexpr size_t bytes = (size_t)T::kWords * 4;
    const size_t wgs = (tiles + tiles_per_wg(LOGT) - 1) / tiles_per_wg(LOGT);
    if constexpr (bytes > 65536) {  // more than 64 KB of dynamic LDS has to be allowed once per kernel and device
        static std::atomic<bool> attr_done[64];  // per kernel instantiation; contexts on several threads may race here
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 1;
        if (dev < 0 || dev >= 64 || !attr_done[dev].load(std::memory_order_acquire)) {
            if (hipFuncSetAttribute((const void *)tile_kernel<LOGT, MODE, WHOLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
                hipSuccess)
                return 1;  // setting it twice from two threads is harmless; the flag only saves the repeated call
            if (dev >= 0 && dev < 64) attr_done[dev].store(true, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL((tile_kernel<LOGT, MODE, WHOLE>), dim3((unsigned)wgs), dim3(T::kLanes * tiles_per_wg(LOGT)), bytes, st, a, b, out,
                       tiles, p);
    return hipGetLastError() != hipSuccess;
}
template <int MODE>
inline int launch_rows(S *a, const S *b, S *out, size_t batch, const P &p, bool one_tile, hipStream_t st) {
    const int logt = one_tile ? p.k : kTileLog;
    const size_t tiles = batch << (p.k - logt);
    if (tiles > 0x7FFFFFFFull) return 1;
    if (!one_tile) return launch_tile<9, MODE, false>(a, b, out, tiles, p, st);
    switch (p.k) {
        case 4: return launch_tile<4, MODE, true>(a, b, out, tiles, p, st);
        case 5: return launch_tile<5, MODE, true>(a, b, out, tiles, p, st);
        case 6: return launch_tile<6, MODE, true>(a, b, out, tiles, p, st);
        case 7: return launch_tile<7, MODE, true>(a, b, out, tiles, p, st);
        case 8: return launch_tile<8, MODE, true>(a, b, out, tiles, p, st);
        case 9: return launch_tile<9, MODE, true>(a, b, out, tiles, p, st);
        case 10: return launch_tile<10, MODE, true>(a, b, out, tiles, p, st);
        case 11: return launch_tile<11, MODE, true>(a, b, out, tiles, p, st);
        case 12: return launch_tile<12, MODE, true>(a, b, out, tiles, p, st);
        default: return 1;
    }
}

}  // namespace st
}  // namespace sr
