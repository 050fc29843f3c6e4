// Register-tiled negacyclic NTT for word-sized fields (BabyBear here; any F with a 1- or 2-word element):
// the reference's merged radix-2 stages (crates/ring/src/cyclotomic_ring/models/stark_prime/ntt.rs:121-346
// generalised, twiddle of stage s block b = psi^brv_k(2^s + b)) executed FOUR STAGES AT A TIME on 16
// coefficients held in registers, instead of one LDS round trip per stage (ntt_generic.hpp).
//
//   D = 2^c * 4096:  strided passes run stages 0..c-1 (2^M legs per lane, wave-uniform twiddles from scalar
//   loads, coalesced stride-4096 accesses); the rows kernel owns one 4096-coefficient tile per workgroup
//   (256 lanes x 16 coefficients) and runs stages c..c+11 as three radix-16 register passes with two trips
//   through a padded LDS tile (pad = pos + pos/16: conflict-free for all three access patterns).
//   MODE_MUL transforms a and b side by side (one twiddle fetch serves both), multiplies slot-wise
//   (ntt_form.rs:177-189) and runs the inverse from registers: a, b read once, c written once.
//   BabyBear arithmetic is 32-bit Montgomery (fields.hpp): 5 instructions per twiddle product, 3 per add/sub.
//   Intermediates between kernels live in library-owned scratch as bare 32-bit words (the reference's
//   BabyBear limb is an 8-byte Fp64 with an always-zero upper half), so the caller's a and b are only read
//   and a ring product moves 48 instead of 72 bytes per coefficient through HBM.
#pragma once
#include <cstdlib>

#include "fields.hpp"

namespace sr {
namespace rt {

constexpr int kTile = 4096;
constexpr int kLds = kTile + kTile / 16;

template <class F>
struct Params {
    int k;   // log2 D
    int c;   // k - 12
    const typename F::elem *tw, *itw;
    typename F::elem scale0, scale1;  // inverse stage 0: sum leg, diff leg (table form)
    bool no_cols256 = false;          // plan: 4-stage column passes + 12-stage rows instead of the 8 + 8 split
};

__device__ __forceinline__ int pad(int pos) { return pos + (pos >> 4); }

// Memory views.  Boundary = the reference's in-memory words (F::storage: for BabyBear an 8-byte Fp64 word whose
// upper half is zero); Packed = bare F::elem words in library-owned scratch.  Intermediates between the strided
// passes and the rows kernel travel Packed, which for BabyBear halves their HBM traffic.
struct Boundary {};
struct Packed {};
template <class F, class V> struct View;
// Boundary words are read once and written once per launch: non-temporal (the same hint took the Goldilocks column passes from
// 3.90 to 3.75 ms; tables keep the default policy and stay in L2).
template <class F> struct View<F, Boundary> {
    typedef typename F::storage T;
    static __device__ __forceinline__ typename F::elem ld(const T *p) {
        const T v = __builtin_nontemporal_load(p);
        return F::load(&v);
    }
    static __device__ __forceinline__ void st(T *p, const typename F::elem &v) {
        T w;
        F::store(&w, v);
        __builtin_nontemporal_store(w, p);
    }
};
// The packed scratch holds a chunk's intermediates; without the hint they allocate in the Infinity Cache, which is what the two-lane
// ring product wants (capi.hip: rt_ring_mul).
template <class F> struct View<F, Packed> {
    typedef typename F::elem T;
    static __device__ __forceinline__ typename F::elem ld(const T *p) { return *p; }
    static __device__ __forceinline__ void st(T *p, const typename F::elem &v) { *p = v; }
};

// Packed words at the BOUNDARY (the opt-in packed-u32 entry points, sr_*_packed32_*: the caller's operands and results are bare
// F::elem words -- for BabyBear the low half of the reference's Fp64 limb, babybear/mod.rs:18-26, whose upper half is always zero):
// same width as the scratch view, but read once / written once per call, hence the non-temporal hint of the boundary view.
struct PackedStream {};
template <class F> struct View<F, PackedStream> {
    typedef typename F::elem T;
    static __device__ __forceinline__ typename F::elem ld(const T *p) { return __builtin_nontemporal_load(p); }
    static __device__ __forceinline__ void st(T *p, const typename F::elem &v) { __builtin_nontemporal_store(v, p); }
};

// the 15 twiddles of four merged stages starting at global stage s0 for a lane working in block blk0 of stage s0:
// w[(1 << u) - 1 + i] = table[2^(s0+u) + (blk0 << u) + i], u = 0..3, i < 2^u.  The 2^u entries of stage s0 + u are
// contiguous and 2^u-aligned, so they are fetched as ONE vector per 16 bytes (4-byte elements: 1 + 1 + 1 + 2 loads instead
// of 15 scalar ones; consecutive lanes then read consecutive vectors, whole cache lines per instruction, where the scalar
// form walked 2^u-word strides and touched up to 16 lines per load).  The tables are 256-byte aligned (hipMalloc).
template <class F>
__device__ __forceinline__ void load_tw16(typename F::elem *w, const typename F::elem *table, int s0, unsigned blk0) {
    using E = typename F::elem;
    if constexpr (sizeof(E) == 4) {
        w[0] = table[(1u << s0) + blk0];
        const uint2 v1 = *reinterpret_cast<const uint2 *>(table + (2u << s0) + (blk0 << 1));
        w[1] = v1.x;
        w[2] = v1.y;
        const uint4 v2 = *reinterpret_cast<const uint4 *>(table + (4u << s0) + (blk0 << 2));
        w[3] = v2.x;
        w[4] = v2.y;
        w[5] = v2.z;
        w[6] = v2.w;
        const uint4 *p3 = reinterpret_cast<const uint4 *>(table + (8u << s0) + (blk0 << 3));
        const uint4 a = p3[0], b = p3[1];
        w[7] = a.x;
        w[8] = a.y;
        w[9] = a.z;
        w[10] = a.w;
        w[11] = b.x;
        w[12] = b.y;
        w[13] = b.z;
        w[14] = b.w;
    } else if constexpr (sizeof(E) == 8) {
        w[0] = table[(1u << s0) + blk0];
#pragma unroll
        for (int u = 1; u < 4; u++) {
            const ulonglong2 *pv = reinterpret_cast<const ulonglong2 *>(table + (1u << (s0 + u)) + (blk0 << u));
#pragma unroll
            for (int i = 0; i < (1 << u) / 2; i++) {
                const ulonglong2 v = pv[i];
                w[(1 << u) - 1 + 2 * i] = v.x;
                w[(1 << u) - 1 + 2 * i + 1] = v.y;
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < (1 << u); i++) w[(1 << u) - 1 + i] = table[(1u << (s0 + u)) + (blk0 << u) + (unsigned)i];
    }
}
// four merged forward stages on x[0..15]: stage u pairs (j, j + (8 >> u)) with twiddle w[(1 << u) - 1 + (j >> (4 - u))]
template <class F>
__device__ __forceinline__ void fwd16(typename F::elem *x, const typename F::elem *w) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j & half) continue;
            const typename F::elem a = x[j], v = F::mul_tw(x[j + half], w[(1 << u) - 1 + (j >> (4 - u))]);
            x[j] = F::add(a, v);
            x[j + half] = F::sub(a, v);
        }
    }
}
// four merged inverse (Gentleman-Sande) stages, last stage first; with HAS_STAGE0 the final one is global stage 0
// and applies the D^-1 constants instead of a twiddle
template <class F, bool HAS_STAGE0>
__device__ __forceinline__ void inv16(typename F::elem *x, const typename F::elem *w, const Params<F> &p) {
#pragma unroll
    for (int u = 3; u >= 0; u--) {
        const int half = 8 >> u;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j & half) continue;
            const typename F::elem a = x[j], b = x[j + half];
            const typename F::elem sum = F::add(a, b), dif = F::sub(a, b);
            if (HAS_STAGE0 && u == 0) {
                x[j] = F::mul_tw(sum, p.scale0);
                x[j + half] = F::mul_tw(dif, p.scale1);
            } else {
                x[j] = sum;
                x[j + half] = F::mul_tw(dif, w[(1 << u) - 1 + (j >> (4 - u))]);
            }
        }
    }
}

// MODE 0: forward in place; 1: inverse in place; 2: out = icrt(crt(a) (.) crt(b)) for this tile
// C0: D = 4096 (c = 0), the tile is a whole ring element and its last inverse pass contains global stage 0
// waves per SIMD the kernels are compiled for: four; the fused product of a two-word field (Goldilocks on this path: the cross-check
// plan SR_PLAN_GL_REGTILE) holds two 16-coefficient tiles of 8-byte words plus their twiddles and gets the registers of two
constexpr int rows_waves(int elem_bytes, int mode) { return (mode == 2 && elem_bytes > 4) ? 2 : 4; }
template <class F, int MODE, class VI, class VO, bool C0>
__global__ __launch_bounds__(256, rows_waves(sizeof(typename F::elem), MODE)) void rows_kernel(const typename View<F, VI>::T *a, const typename View<F, VI>::T *b,
                                                      typename View<F, VO>::T *out, Params<F> p) {
    using E = typename F::elem;
    using In = View<F, VI>;
    using Out = View<F, VO>;
    __shared__ E lds[(MODE == 2 ? 2 : 1) * kLds];
    E *la = lds, *lb = lds + kLds;
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kTile;
    const unsigned tile_blk = blockIdx.x & ((1u << p.c) - 1u);
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
    E x[16], y[16], w[15], wn[15];
    const unsigned blk2 = (tile_blk << 4) + (unsigned)(t >> 4), blk3 = (tile_blk << 8) + (unsigned)t;

    if (MODE != 1) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x[j] = In::ld(a + base + j * 256 + t);
            if (MODE == 2) y[j] = In::ld(b + base + j * 256 + t);
        }
        // pass 1: tile-local stages 0..3, twiddles uniform over the workgroup (scalar loads); the per-lane
        // twiddles of pass 2 are requested now so that they arrive behind the LDS exchange
        load_tw16<F>(w, p.tw, p.c, tile_blk);
        load_tw16<F>(wn, p.tw, p.c + 4, blk2);
        fwd16<F>(x, w);
        if (MODE == 2) fwd16<F>(y, w);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            la[pad(r * 256 + t)] = x[r];
            if (MODE == 2) lb[pad(r * 256 + t)] = y[r];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x[j] = la[pad(base2 + j * 16)];
            if (MODE == 2) y[j] = lb[pad(base2 + j * 16)];
        }
        // pass 2: stages 4..7; after stage 3 the tile holds 16 blocks of 256, this lane works in block t >> 4
        load_tw16<F>(w, p.tw, p.c + 8, blk3);  // pass-3 twiddles in flight during pass 2
        fwd16<F>(x, wn);
        if (MODE == 2) fwd16<F>(y, wn);
#pragma unroll
        for (int s = 0; s < 16; s++) {  // the very slots this lane just read
            la[pad(base2 + s * 16)] = x[s];
            if (MODE == 2) lb[pad(base2 + s * 16)] = y[s];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x[j] = la[17 * t + j];
            if (MODE == 2) y[j] = lb[17 * t + j];
        }
        // pass 3: stages 8..11 inside this lane's own 16 coefficients
        if (MODE == 2) load_tw16<F>(wn, p.itw, p.c + 8, blk3);  // first inverse pass
        fwd16<F>(x, w);
        if (MODE == 0) {
            // results sit 16-contiguous per lane; one more exchange makes the global store lane-contiguous
#pragma unroll
            for (int j = 0; j < 16; j++) la[17 * t + j] = x[j];  // own pass-3 slots
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) Out::st(out + base + j * 256 + t, la[pad(j * 256 + t)]);
            return;
        }
        fwd16<F>(y, w);
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = F::mul_tw(x[j], y[j]);
    } else {
        // lane-contiguous global load, then an exchange into the 16-contiguous-per-lane layout of the first pass
#pragma unroll
        for (int j = 0; j < 16; j++) la[pad(j * 256 + t)] = In::ld(a + base + j * 256 + t);
        load_tw16<F>(wn, p.itw, p.c + 8, blk3);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = la[17 * t + j];
    }

    // inverse: stages c+11 .. c
    load_tw16<F>(w, p.itw, p.c + 4, blk2);
    inv16<F, false>(x, wn, p);
#pragma unroll
    for (int j = 0; j < 16; j++) la[17 * t + j] = x[j];  // own slots (pass-3 reads of la by this lane are done)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) x[s] = la[pad(base2 + s * 16)];
    load_tw16<F>(wn, p.itw, p.c, tile_blk);
    inv16<F, false>(x, w, p);
#pragma unroll
    for (int j = 0; j < 16; j++) la[pad(base2 + j * 16)] = x[j];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; r++) x[r] = la[pad(r * 256 + t)];
    inv16<F, C0>(x, wn, p);
#pragma unroll
    for (int j = 0; j < 16; j++) Out::st(out + base + j * 256 + t, x[j]);
}

// S0: this pass starts at global stage 0 (s_lo == 0)
template <class F, int M, int DIR, class VI, class VO, bool S0>
__global__ __launch_bounds__(256) void strided_kernel(const typename View<F, VI>::T *src, typename View<F, VO>::T *dst,
                                                      int s_lo, Params<F> p) {
    using E = typename F::elem;
    constexpr int R = 1 << M;
    const int ls = p.k - s_lo - M;
    const unsigned chunks = 1u << (ls - 8);
    const unsigned ci = blockIdx.x & (chunks - 1u);
    const unsigned rest = blockIdx.x >> (ls - 8);
    const unsigned h = rest & ((1u << s_lo) - 1u);
    const size_t poly = rest >> s_lo;
    const size_t off = (poly << p.k) + ((size_t)h << (p.k - s_lo)) + ci * 256u + threadIdx.x;
    E x[R];
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = View<F, VI>::ld(src + off + ((size_t)j << ls));
    if (DIR == 0) {
#pragma unroll
        for (int u = 0; u < M; u++) {
            const int half = 1 << (M - 1 - u);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (j & half) continue;
                const E w = p.tw[(1u << (s_lo + u)) + (h << u) + (unsigned)(j >> (M - u))];
                const E a = x[j], v = F::mul_tw(x[j + half], w);
                x[j] = F::add(a, v);
                x[j + half] = F::sub(a, v);
            }
        }
    } else {
#pragma unroll
        for (int u = M - 1; u >= 0; u--) {
            const int half = 1 << (M - 1 - u);
#pragma unroll
            for (int j = 0; j < R; j++) {
                if (j & half) continue;
                const E a = x[j], b = x[j + half];
                const E sum = F::add(a, b), dif = F::sub(a, b);
                if (S0 && u == 0) {
                    x[j] = F::mul_tw(sum, p.scale0);
                    x[j + half] = F::mul_tw(dif, p.scale1);
                } else {
                    x[j] = sum;
                    x[j + half] = F::mul_tw(dif, p.itw[(1u << (s_lo + u)) + (h << u) + (unsigned)(j >> (M - u))]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < R; j++) View<F, VO>::st(dst + off + ((size_t)j << ls), x[j]);
}

// ------------------------------------------------------------------------------------------------------------------
// D = 2^16 and 2^20: stages 0..7 in ONE column pass (cols256_kernel) instead of two 4-stage passes / instead of leaving
// twelve stages to the VALU-bound rows kernel.  A workgroup owns 256 legs (stride N2 = D / 256) x 16 consecutive columns:
// pass A (stages 0..3, twiddles uniform over the grid) on legs rg + 16 jj, one exchange through the padded LDS tile,
// pass B (stages 4..7, the lane's block is rg') on legs 16 rg' + j.  The pass streams one operand once and has VALU to
// spare, so the four stages it takes over from the rows kernel are free.  grid.x = npoly * N2 / 16.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kXcdGroup = 3;  // 8 consecutive tiles per XCD turn: 1 KiB (4-byte words, 32 columns) or 1 KiB (8-byte, 16 columns) of every leg
template <class F, int DIR, class VI, class VO, int COLS>
__global__ __launch_bounds__(16 * COLS) void cols256_kernel(const typename View<F, VI>::T *src, typename View<F, VO>::T *dst,
                                                            const typename View<F, VI>::T *src2, typename View<F, VO>::T *dst2,
                                                            Params<F> p, unsigned grouped) {
    using E = typename F::elem;
    if (blockIdx.y) {  // grid.y = 2: the forward passes of both operands of a ring product in one launch
        src = src2;
        dst = dst2;
    }
    // [leg][column] words; COLS = 32 for 4-byte elements so that a leg's segment is a whole 128-byte line on the packed side
    __shared__ E lds[256 * COLS + 16 * COLS];
    constexpr int LC = COLS == 32 ? 5 : 4;
    const int t = threadIdx.x;
    const int ls = p.k - 8;  // log2 N2
    const unsigned tile = xcd_tile(blockIdx.x, kXcdGroup, grouped);  // runs of column chunks share an XCD (fields.hpp)
    const unsigned ci = tile & ((1u << (ls - LC)) - 1u);
    const size_t poly = tile >> (ls - LC);
    const int col = t & (COLS - 1), rg = t >> LC;
    const size_t off = (poly << p.k) + ci * (unsigned)COLS + (unsigned)col;
    auto at = [](int leg, int c) { const int pos = leg * COLS + c; return pos + (pos >> 8) * (COLS == 32 ? 0 : 16); };
    E x[16], w[15];
    if (DIR == 0) {
#pragma unroll
        for (int jj = 0; jj < 16; jj++) x[jj] = View<F, VI>::ld(src + off + ((size_t)(rg + 16 * jj) << ls));
        load_tw16<F>(w, p.tw, 0, 0u);
        fwd16<F>(x, w);
        load_tw16<F>(w, p.tw, 4, (unsigned)rg);  // pass-B twiddles in flight across the exchange
#pragma unroll
        for (int h = 0; h < 16; h++) lds[at(16 * h + rg, col)] = x[h];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = lds[at(16 * rg + j, col)];  // block rg, leg j
        fwd16<F>(x, w);
#pragma unroll
        for (int j = 0; j < 16; j++) View<F, VO>::st(dst + off + ((size_t)(16 * rg + j) << ls), x[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = View<F, VI>::ld(src + off + ((size_t)(16 * rg + j) << ls));
        load_tw16<F>(w, p.itw, 4, (unsigned)rg);
        inv16<F, false>(x, w, p);
        load_tw16<F>(w, p.itw, 0, 0u);
#pragma unroll
        for (int j = 0; j < 16; j++) lds[at(16 * rg + j, col)] = x[j];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 16; h++) x[h] = lds[at(16 * h + rg, col)];
        inv16<F, true>(x, w, p);  // contains global stage 0: the D^-1 constants
#pragma unroll
        for (int jj = 0; jj < 16; jj++) View<F, VO>::st(dst + off + ((size_t)(rg + 16 * jj) << ls), x[jj]);
    }
}

// (A variant of this pass for the packed-u32 boundary in which a lane owns two neighbouring columns -- 256-byte segments, 8-byte
// accesses, 512-lane workgroups -- measured 2 x 1.77 / 1.85 ms against 2 x 1.68 / 1.77 ms for this kernel on the packed BabyBear
// workload and is not kept: two workgroups of eight waves per CU overlap their load, exchange and store phases worse than four do.)

// D = 2^16 behind cols256: a 4096-coefficient tile is 16 blocks of 256 coefficients that only need stages 8..15:
// the last two register passes of rows_kernel (Params::c = k - 12 as usual, so the twiddle indices are the same),
// one LDS exchange per transform.  Lane (rho, i0) starts in block rho of the tile.
template <class F, int MODE, class VI, class VO>
__global__ __launch_bounds__(256, rows_waves(sizeof(typename F::elem), MODE)) void rows256_kernel(const typename View<F, VI>::T *a, const typename View<F, VI>::T *b,
                                                         typename View<F, VO>::T *out, Params<F> p) {
    using E = typename F::elem;
    using In = View<F, VI>;
    using Out = View<F, VO>;
    // (one shared tile for a and b -- more workgroups per CU, 98 VGPRs -- and 5 or 6 waves per SIMD were measured: 4.58-5.14 ms
    // against 4.59 ms for this form; the kernel is bound by instruction issue)
    __shared__ E lds[(MODE == 2 ? 2 : 1) * kLds];
    E *la = lds, *lb = lds + kLds;
    const int t = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * kTile;
    const unsigned tile_blk = blockIdx.x & ((1u << p.c) - 1u);
    const int i0 = t & 15, base2 = (t >> 4) * 256 + i0;
    E x[16], y[16], w[15], wn[15];
    const unsigned blk2 = (tile_blk << 4) + (unsigned)(t >> 4), blk3 = (tile_blk << 8) + (unsigned)t;

    // (Sending the tile through LDS so that global accesses are lane-contiguous instead of 64-byte segments measured 11.1 ms against
    // 4.8 ms on BabyBear D = 2^16: the extra exchanges and the registers they hold cost more than the half-line accesses.)
    if (MODE != 1) {
        load_tw16<F>(wn, p.tw, p.c + 4, blk2);
        load_tw16<F>(w, p.tw, p.c + 8, blk3);
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x[j] = In::ld(a + base + base2 + j * 16);
            if (MODE == 2) y[j] = In::ld(b + base + base2 + j * 16);
        }
        fwd16<F>(x, wn);
        if (MODE == 2) fwd16<F>(y, wn);
#pragma unroll
        for (int s = 0; s < 16; s++) {
            la[pad(base2 + s * 16)] = x[s];
            if (MODE == 2) lb[pad(base2 + s * 16)] = y[s];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            x[j] = la[17 * t + j];
            if (MODE == 2) y[j] = lb[17 * t + j];
        }
        if (MODE == 2) load_tw16<F>(wn, p.itw, p.c + 8, blk3);
        fwd16<F>(x, w);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 16; j++) la[17 * t + j] = x[j];  // own slots
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; j++) Out::st(out + base + j * 256 + t, la[pad(j * 256 + t)]);
            return;
        }
        fwd16<F>(y, w);
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = F::mul_tw(x[j], y[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 16; j++) la[pad(j * 256 + t)] = In::ld(a + base + j * 256 + t);
        load_tw16<F>(wn, p.itw, p.c + 8, blk3);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = la[17 * t + j];
    }
    load_tw16<F>(w, p.itw, p.c + 4, blk2);
    inv16<F, false>(x, wn, p);
#pragma unroll
    for (int j = 0; j < 16; j++) la[17 * t + j] = x[j];  // own slots
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; s++) x[s] = la[pad(base2 + s * 16)];
    inv16<F, false>(x, w, p);
#pragma unroll
    for (int j = 0; j < 16; j++) Out::st(out + base + base2 + j * 16, x[j]);
}

// ---- host-side launchers ---------------------------------------------------------------------------
inline int plan(int c, int *ms) {  // split c strided stages into register passes of at most 4
    int n = 0;
    while (c > 0) {
        int m = c > 4 ? (c >= 8 ? 4 : (c + 1) / 2) : c;
        ms[n++] = m;
        c -= m;
    }
    return n;
}
struct Hooks {  // optional per-launch timing (tags: 0 strided fwd, 1 rows, 2 strided inv)
    void (*begin)(void *user, int tag, hipStream_t st) = nullptr;
    void (*end)(void *user, hipStream_t st) = nullptr;
    void *user = nullptr;
};
struct Scope {
    const Hooks &h;
    hipStream_t st;
    Scope(const Hooks &hh, int tag, hipStream_t s) : h(hh), st(s) {
        if (h.begin) h.begin(h.user, tag, st);
    }
    ~Scope() {
        if (h.end) h.end(h.user, st);
    }
};
template <class F, int DIR, class VI, class VO>
inline int launch_strided(const Hooks &hk, int M, const typename View<F, VI>::T *src, typename View<F, VO>::T *dst, int s_lo,
                          size_t npoly, const Params<F> &p, hipStream_t st) {
    const size_t blocks = (npoly << s_lo) << (p.k - s_lo - M - 8);
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    Scope sc(hk, DIR == 0 ? 0 : 2, st);
    dim3 g((unsigned)blocks), b(256);
#define SR_RT_STRIDED(MM)                                                                                          \
    case MM:                                                                                                       \
        if (s_lo == 0) hipLaunchKernelGGL((strided_kernel<F, MM, DIR, VI, VO, true>), g, b, 0, st, src, dst, s_lo, p); \
        else hipLaunchKernelGGL((strided_kernel<F, MM, DIR, VI, VO, false>), g, b, 0, st, src, dst, s_lo, p);          \
        break;
    switch (M) {
        SR_RT_STRIDED(1) SR_RT_STRIDED(2) SR_RT_STRIDED(3) SR_RT_STRIDED(4)
        default: return 1;
    }
#undef SR_RT_STRIDED
    return hipGetLastError() != hipSuccess;
}
// 2^16 and 2^20 run stages 0..7 as one cols256 launch (SR_RT_COLS256=0: the 4-stage register passes, kept for A/B tests)
template <class F>
inline bool use_cols256(const Params<F> &p) {
    return !p.no_cols256 && (p.k == 16 || p.k == 20);  // no_cols256: sr_plan flag SR_PLAN_RT_NO_COLS256, fixed per context
}
template <class F, int DIR, class VI, class VO>
inline int launch_cols256(const Hooks &hk, const typename View<F, VI>::T *src, typename View<F, VO>::T *dst, size_t npoly,
                          const Params<F> &p, hipStream_t st) {
    constexpr int COLS = sizeof(typename F::elem) == 4 ? 32 : 16;
    const size_t blocks = (npoly << (p.k - 8)) / COLS;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    Scope sc(hk, DIR == 0 ? 0 : 2, st);
    hipLaunchKernelGGL((cols256_kernel<F, DIR, VI, VO, COLS>), dim3((unsigned)blocks), dim3(16 * COLS), 0, st, src, dst,
                       (const typename View<F, VI>::T *)nullptr, (typename View<F, VO>::T *)nullptr, p, xcd_grouped_tiles(blocks, kXcdGroup));
    return hipGetLastError() != hipSuccess;
}
// the forward column passes of both operands of a ring product in ONE launch (grid.y = 2)
template <class F, class VB>
inline int launch_cols256_pair(const Hooks &hk, const typename View<F, VB>::T *a, typename F::elem *da, const typename View<F, VB>::T *b,
                               typename F::elem *db, size_t npoly, const Params<F> &p, hipStream_t st) {
    constexpr int COLS = sizeof(typename F::elem) == 4 ? 32 : 16;
    const size_t blocks = (npoly << (p.k - 8)) / COLS;
    if (blocks == 0 || blocks > 0x7FFFFFFFull) return 1;
    Scope sc(hk, 0, st);
    hipLaunchKernelGGL((cols256_kernel<F, 0, VB, Packed, COLS>), dim3((unsigned)blocks, 2), dim3(16 * COLS), 0, st, a, da, b, db, p,
                       xcd_grouped_tiles(blocks, kXcdGroup));
    return hipGetLastError() != hipSuccess;
}
// VB = the view of the caller's operands and results: Boundary (the reference's 8-byte words; every sr_* entry point) or
// PackedStream (bare F::elem words; the sr_*_packed32_* entry points).
// forward strided stages: boundary words at `src` -> packed words at `dst` (c >= 1)
template <class F, class VB = Boundary>
inline int strided_fwd(const Hooks &hk, const typename View<F, VB>::T *src, typename F::elem *dst, size_t npoly,
                       const Params<F> &p, hipStream_t st) {
    if (use_cols256(p)) return launch_cols256<F, 0, VB, Packed>(hk, src, dst, npoly, p, st);
    int ms[8], s_lo = 0;
    const int n = plan(p.c, ms);
    for (int i = 0; i < n; i++) {
        int rc = i == 0 ? launch_strided<F, 0, VB, Packed>(hk, ms[i], src, dst, s_lo, npoly, p, st)
                        : launch_strided<F, 0, Packed, Packed>(hk, ms[i], dst, dst, s_lo, npoly, p, st);
        if (rc) return rc;
        s_lo += ms[i];
    }
    return 0;
}
// inverse strided stages: packed words at `src` (clobbered) -> boundary words at `dst`
template <class F, class VB = Boundary>
inline int strided_inv(const Hooks &hk, typename F::elem *src, typename View<F, VB>::T *dst, size_t npoly, const Params<F> &p,
                       hipStream_t st) {
    if (use_cols256(p)) return launch_cols256<F, 1, Packed, VB>(hk, src, dst, npoly, p, st);
    int ms[8], s_lo = p.c;
    const int n = plan(p.c, ms);
    for (int i = n - 1; i >= 0; i--) {
        s_lo -= ms[i];
        int rc = i == 0 ? launch_strided<F, 1, Packed, VB>(hk, ms[i], src, dst, s_lo, npoly, p, st)
                        : launch_strided<F, 1, Packed, Packed>(hk, ms[i], src, src, s_lo, npoly, p, st);
        if (rc) return rc;
    }
    return 0;
}
template <class F, int MODE, class VI, class VO>
inline int launch_rows(const Hooks &hk, const typename View<F, VI>::T *a, const typename View<F, VI>::T *b,
                       typename View<F, VO>::T *out, size_t npoly, const Params<F> &p, hipStream_t st) {
    const size_t tiles = npoly << p.c;
    if (tiles == 0 || tiles > 0x7FFFFFFFull) return 1;
    Scope sc(hk, 1, st);
    if (p.k == 16 && use_cols256(p))  // stages 0..7 are done (or come last) in cols256: only 8..15 here
        hipLaunchKernelGGL((rows256_kernel<F, MODE, VI, VO>), dim3((unsigned)tiles), dim3(256), 0, st, a, b, out, p);
    else if (p.c == 0)
        hipLaunchKernelGGL((rows_kernel<F, MODE, VI, VO, true>), dim3((unsigned)tiles), dim3(256), 0, st, a, b, out, p);
    else
        hipLaunchKernelGGL((rows_kernel<F, MODE, VI, VO, false>), dim3((unsigned)tiles), dim3(256), 0, st, a, b, out, p);
    return hipGetLastError() != hipSuccess;
}
// scratch0 / scratch1: library-owned device buffers of batch * D elems each (only touched when c > 0)
template <class F, class VB = Boundary>
inline int fwd(const Hooks &hk, typename View<F, VB>::T *d, size_t batch, const Params<F> &p, typename F::elem *scratch0,
               hipStream_t st) {
    if (batch == 0) return 0;
    if (p.c == 0) return launch_rows<F, 0, VB, VB>(hk, d, nullptr, d, batch, p, st);
    if (strided_fwd<F, VB>(hk, d, scratch0, batch, p, st)) return 1;
    return launch_rows<F, 0, Packed, VB>(hk, scratch0, nullptr, d, batch, p, st);
}
template <class F, class VB = Boundary>
inline int inv(const Hooks &hk, typename View<F, VB>::T *d, size_t batch, const Params<F> &p, typename F::elem *scratch0,
               hipStream_t st) {
    if (batch == 0) return 0;
    if (p.c == 0) return launch_rows<F, 1, VB, VB>(hk, d, nullptr, d, batch, p, st);
    if (launch_rows<F, 1, VB, Packed>(hk, d, nullptr, scratch0, batch, p, st)) return 1;
    return strided_inv<F, VB>(hk, scratch0, d, batch, p, st);
}
// p must carry the FUSED stage-0 constants.  a and b are only read; out may alias a.
template <class F, class VB = Boundary>
inline int ring_mul(const Hooks &hk, typename View<F, VB>::T *out, const typename View<F, VB>::T *a, const typename View<F, VB>::T *b,
                    size_t batch, const Params<F> &p, typename F::elem *scratch0, typename F::elem *scratch1, hipStream_t st) {
    if (batch == 0) return 0;
    if (p.c == 0) return launch_rows<F, 2, VB, VB>(hk, a, b, out, batch, p, st);
    if (use_cols256(p)) {
        if (launch_cols256_pair<F, VB>(hk, a, scratch0, b, scratch1, batch, p, st)) return 1;
    } else {
        if (strided_fwd<F, VB>(hk, a, scratch0, batch, p, st)) return 1;
        if (strided_fwd<F, VB>(hk, b, scratch1, batch, p, st)) return 1;
    }
    if (launch_rows<F, 2, Packed, Packed>(hk, scratch0, scratch1, scratch0, batch, p, st)) return 1;
    return strided_inv<F, VB>(hk, scratch0, out, batch, p, st);
}

}  // namespace rt
}  // namespace sr
