// Generic (any field F from fields.hpp) negacyclic NTT kernels for Fp[X]/(X^D+1), D = 2^k.
//
// Algorithm and slot order: the reference's stark_prime CRT generalised to D = 2^k
// (crates/ring/src/cyclotomic_ring/models/stark_prime/ntt.rs:121-235 forward, :245-346 inverse):
//   forward  stage s = 0..k-1, half = D >> (s+1), block b: w = psi^brv_k(2^s + b)
//            (u, v) -> (u + w v, u - w v)                                   [Cooley-Tukey]
//   inverse  stage s = k-1..0: (u, v) -> (u + v, w^-1 (u - v)); D^-1 folded into stage 0
//            exactly like SIXTEEN_INV / SIXTEEN_INV_TIMES_ROOT (ntt.rs:51-55, 339-345).
//
// Two kernel shapes cover every degree:
//   rows_kernel  -- a workgroup owns TILE contiguous coefficients of the flat batch in LDS and runs
//                   the stages whose butterflies stay inside the tile (half < TILE).  For D <= TILE a
//                   tile holds TILE/D whole ring elements.  MODE_MUL fuses fwd(a), fwd(b), the slot
//                   product (ntt_form.rs:177-189) and the inverse stages in one pass over HBM.
//   cols_kernel  -- the stages with half >= TILE: a workgroup gathers C = D/TILE strided rows of W
//                   contiguous coefficients into LDS and runs log2(C) stages across the rows.
// These are the correctness-first, field-agnostic kernels (radix-2 butterflies through LDS, one
// barrier per stage); ntt_goldilocks.hpp holds the tuned Goldilocks path.
#pragma once
#include "fields.hpp"
#include "stark_lazy.hpp"

namespace sr {

enum { MODE_FWD = 0, MODE_INV = 1, MODE_MUL = 2 };
constexpr int kThreads = 256;

template <class F>
struct NttParams {
    int k;          // log2 D
    int s_rows;     // first stage handled by rows_kernel (0 when D <= TILE)
    int log_tile;   // log2 TILE
    const typename F::elem *tw;   // tw[2^s + b]  = psi^ brv_k(2^s+b)   (table form)
    const typename F::elem *itw;  // itw[2^s + b] = psi^-brv_k(2^s+b)
    typename F::elem scale0;      // inverse stage 0, sum leg   (D^-1 [* fused correction])
    typename F::elem scale1;      // inverse stage 0, diff leg  (scale0 * psi^-(D/2))
};

template <class F>
__device__ __forceinline__ void fwd_stage_lds(uint32_t *lds, int n_tile, int n_loc, size_t off, int k, int s,
                                              const typename F::elem *tw, bool relax = false) {
    const int lh = k - s - 1;
    const int half = 1 << lh;
    for (int j = threadIdx.x; j < (n_tile >> 1); j += kThreads) {
        int i = j & (half - 1);
        int lo = ((j >> lh) << (lh + 1)) + i;
        int hi = lo + half;
        if (hi < n_loc) {
            size_t g = off + lo;
            uint32_t b = (uint32_t)(g >> (lh + 1)) & ((1u << s) - 1u);
            typename F::elem w = tw[(1u << s) + b];
            typename F::elem u = F::lds_get(lds, lo, n_tile);
            typename F::elem v = F::mul_tw(F::lds_get(lds, hi, n_tile), w);
            if (Lazy<F>::value && relax) {  // lazy fields: every sixth stage pulls limbs and value back in (stark_lazy.hpp)
                F::lds_put(lds, lo, n_tile, Lazy<F>::weak(F::add(u, v)));
                F::lds_put(lds, hi, n_tile, Lazy<F>::weak(F::sub(u, v)));
            } else {
                F::lds_put(lds, lo, n_tile, F::add(u, v));
                F::lds_put(lds, hi, n_tile, F::sub(u, v));
            }
        }
    }
}

template <class F>
__device__ __forceinline__ void inv_stage_lds(uint32_t *lds, int n_tile, int n_loc, size_t off, int k, int s,
                                              const NttParams<F> &p) {
    const int lh = k - s - 1;
    const int half = 1 << lh;
    for (int j = threadIdx.x; j < (n_tile >> 1); j += kThreads) {
        int i = j & (half - 1);
        int lo = ((j >> lh) << (lh + 1)) + i;
        int hi = lo + half;
        if (hi < n_loc) {
            typename F::elem u = F::lds_get(lds, lo, n_tile);
            typename F::elem v = F::lds_get(lds, hi, n_tile);
            typename F::elem sum = F::add(u, v), dif = F::sub(u, v);
            if (s == 0) {
                F::lds_put(lds, lo, n_tile, F::mul_tw(sum, p.scale0));
                F::lds_put(lds, hi, n_tile, F::mul_tw(dif, p.scale1));
            } else {
                size_t g = off + lo;
                uint32_t b = (uint32_t)(g >> (lh + 1)) & ((1u << s) - 1u);
                F::lds_put(lds, lo, n_tile, Lazy<F>::weak(sum));  // the sum leg doubles every stage: lazy fields reduce it weakly
                F::lds_put(lds, hi, n_tile, F::mul_tw(dif, p.itw[(1u << s) + b]));
            }
        }
    }
}

// n_total = batch * D flat coefficients.  grid.x = ceil(n_total / TILE).
template <class F, int MODE>
__global__ __launch_bounds__(kThreads) void rows_kernel(typename F::storage *a, const typename F::storage *b,
                                                        typename F::storage *out, size_t n_total, NttParams<F> p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int n_tile = 1 << p.log_tile;
    const size_t off = (size_t)blockIdx.x << p.log_tile;
    const size_t rem = n_total - off;
    const int n_loc = rem < (size_t)n_tile ? (int)rem : n_tile;
    uint32_t *la = smem;
    uint32_t *lb = smem + n_tile * F::kLdsWords;

    for (int i = threadIdx.x; i < n_loc; i += kThreads) {
        F::lds_put(la, i, n_tile, F::load(a + off + i));
        if (MODE == MODE_MUL) F::lds_put(lb, i, n_tile, F::load(b + off + i));
    }
    __syncthreads();

    if (MODE == MODE_FWD || MODE == MODE_MUL) {
        for (int s = p.s_rows; s < p.k; s++) {
            const bool relax = (s - p.s_rows) % 6 == 5;
            fwd_stage_lds<F>(la, n_tile, n_loc, off, p.k, s, p.tw, relax);
            if (MODE == MODE_MUL) fwd_stage_lds<F>(lb, n_tile, n_loc, off, p.k, s, p.tw, relax);
            __syncthreads();
        }
    }
    if (MODE == MODE_MUL) {
        for (int i = threadIdx.x; i < n_loc; i += kThreads)
            F::lds_put(la, i, n_tile, Lazy<F>::mul_data(F::lds_get(la, i, n_tile), F::lds_get(lb, i, n_tile)));
        __syncthreads();
    }
    if (MODE == MODE_INV || MODE == MODE_MUL) {
        for (int s = p.k - 1; s >= p.s_rows; s--) {
            inv_stage_lds<F>(la, n_tile, n_loc, off, p.k, s, p);
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < n_loc; i += kThreads) F::store(out + off + i, F::lds_get(la, i, n_tile));
}

// Stages [0, s_rows): C = 2^s_rows rows of length Tr = D / C; a workgroup owns W = TILE / C
// consecutive columns of one ring element.  grid.x = batch * (Tr / W) = batch * D / TILE.
template <class F, int MODE>
__global__ __launch_bounds__(kThreads) void cols_kernel(typename F::storage *a, const typename F::storage *src, size_t batch,
                                                        NttParams<F> p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int n_tile = 1 << p.log_tile;
    const int lc = p.s_rows;              // log2 C
    const int lw = p.log_tile - lc;       // log2 W
    const int W = 1 << lw;
    const int ltr = p.k - lc;             // log2 row length
    const size_t chunks = (size_t)1 << (ltr - lw);
    const size_t poly = blockIdx.x / chunks;
    const size_t chunk = blockIdx.x % chunks;
    if (poly >= batch) return;
    typename F::storage *base = a + (poly << p.k) + (chunk << lw);
    const typename F::storage *sbase = src + (poly << p.k) + (chunk << lw);  // src == a: in place; otherwise src is only read

    for (int idx = threadIdx.x; idx < n_tile; idx += kThreads) {
        int c = idx >> lw, w = idx & (W - 1);
        F::lds_put(smem, idx, n_tile, F::load(sbase + ((size_t)c << ltr) + w));
    }
    __syncthreads();

    if (MODE == MODE_FWD) {
        for (int s = 0; s < lc; s++) {
            const int lhc = lc - s - 1;  // log2 of half, in rows
            for (int j = threadIdx.x; j < (n_tile >> 1); j += kThreads) {
                int w = j & (W - 1), r = j >> lw;
                int i = r & ((1 << lhc) - 1), grp = r >> lhc;
                int lo = (((grp << (lhc + 1)) + i) << lw) + w;
                int hi = lo + (1 << (lhc + lw));
                typename F::elem tw = p.tw[(1u << s) + grp];
                typename F::elem u = F::lds_get(smem, lo, n_tile);
                typename F::elem v = F::mul_tw(F::lds_get(smem, hi, n_tile), tw);
                if (Lazy<F>::value && s % 6 == 5) {
                    F::lds_put(smem, lo, n_tile, Lazy<F>::weak(F::add(u, v)));
                    F::lds_put(smem, hi, n_tile, Lazy<F>::weak(F::sub(u, v)));
                } else {
                    F::lds_put(smem, lo, n_tile, F::add(u, v));
                    F::lds_put(smem, hi, n_tile, F::sub(u, v));
                }
            }
            __syncthreads();
        }
    } else {
        for (int s = lc - 1; s >= 0; s--) {
            const int lhc = lc - s - 1;
            for (int j = threadIdx.x; j < (n_tile >> 1); j += kThreads) {
                int w = j & (W - 1), r = j >> lw;
                int i = r & ((1 << lhc) - 1), grp = r >> lhc;
                int lo = (((grp << (lhc + 1)) + i) << lw) + w;
                int hi = lo + (1 << (lhc + lw));
                typename F::elem u = F::lds_get(smem, lo, n_tile);
                typename F::elem v = F::lds_get(smem, hi, n_tile);
                typename F::elem sum = F::add(u, v), dif = F::sub(u, v);
                if (s == 0) {
                    F::lds_put(smem, lo, n_tile, F::mul_tw(sum, p.scale0));
                    F::lds_put(smem, hi, n_tile, F::mul_tw(dif, p.scale1));
                } else {
                    F::lds_put(smem, lo, n_tile, Lazy<F>::weak(sum));
                    F::lds_put(smem, hi, n_tile, F::mul_tw(dif, p.itw[(1u << s) + grp]));
                }
            }
            __syncthreads();
        }
    }
    for (int idx = threadIdx.x; idx < n_tile; idx += kThreads) {
        int c = idx >> lw, w = idx & (W - 1);
        F::store(base + ((size_t)c << ltr) + w, F::lds_get(smem, idx, n_tile));
    }
}

// ---- element-wise kernels ----------------------------------------------------------------
// Streaming kernels move 16 bytes per lane per access (MI355X_MICROARCH.md: 6.3 TB/s for a float4 copy against ~5 TB/s at
// 8 B per lane): two coefficients of the one-limb fields as one ulonglong2; a Stark coefficient is two such accesses already.
// Unaligned buffers and an odd tail fall back to one coefficient per lane.
template <class F, class Op>
__device__ __forceinline__ void elementwise2(typename F::storage *lhs, const typename F::storage *rhs, size_t n, Op op) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if constexpr (sizeof(typename F::storage) == 8) {
        if ((((uintptr_t)lhs | (uintptr_t)rhs) & 15u) == 0) {
            // non-temporal 16-byte accesses, one pair per lane when the grid allows (stream_blocks): the form that streams best on
            // this box (tools/ubench/stream_rates.hip: a += b at 6.45 TB/s; 5.86 with the default cache policy, 5.6 / 5.2 / 4.9 when
            // a lane loops 2 / 32 / 256 times)
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const size_t pairs = n >> 1;
            u64x2 *l2 = reinterpret_cast<u64x2 *>(lhs);
            const u64x2 *r2 = reinterpret_cast<const u64x2 *>(rhs);
            for (size_t i = gid; i < pairs; i += stride) {
                u64x2 x = __builtin_nontemporal_load(l2 + i);
                const u64x2 y = __builtin_nontemporal_load(r2 + i);
                typename F::storage xs[2] = {x.x, x.y}, ys[2] = {y.x, y.y};
                F::store(&xs[0], op(F::load(&xs[0]), F::load(&ys[0])));
                F::store(&xs[1], op(F::load(&xs[1]), F::load(&ys[1])));
                x.x = xs[0];
                x.y = xs[1];
                __builtin_nontemporal_store(x, l2 + i);
            }
            done = pairs << 1;
        }
    }
    for (size_t i = done + gid; i < n; i += stride) F::store(lhs + i, op(F::load(lhs + i), F::load(rhs + i)));
}
// lhs[i] = lhs[i] * rhs[i] on the in-memory images (ntt_form.rs:177-189)
template <class F>
__global__ __launch_bounds__(256) void pointwise_kernel(typename F::storage *lhs, const typename F::storage *rhs, size_t n) {
    elementwise2<F>(lhs, rhs, n, [](const typename F::elem &a, const typename F::elem &b) { return F::mul_boundary(a, b); });
}

// lhs[e D + i] = lhs[e D + i] * r[i]: every element of the batch times ONE ring element, slot-wise -- `Matrix<R> *= &R`
// (linear_algebra/src/matrix.rs:207-211), `SparseMatrix<R> *= &R` (sparse_matrix.rs:303-307).  D is a power of two; r stays in L2.
template <class F>
__global__ __launch_bounds__(256) void pointwise_bcast_kernel(typename F::storage *lhs, const typename F::storage *r, size_t n, size_t dmask) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if constexpr (sizeof(typename F::storage) == 8) {
        if (dmask != 0 && (((uintptr_t)lhs | (uintptr_t)r) & 15u) == 0) {   // D >= 2: a pair of coefficients never straddles two elements
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const size_t pairs = n >> 1, pmask = dmask >> 1;
            u64x2 *l2 = reinterpret_cast<u64x2 *>(lhs);
            const u64x2 *r2 = reinterpret_cast<const u64x2 *>(r);
            for (size_t i = gid; i < pairs; i += stride) {
                u64x2 x = __builtin_nontemporal_load(l2 + i);
                const u64x2 y = r2[i & pmask];
                typename F::storage xs[2] = {x.x, x.y}, ys[2] = {y.x, y.y};
                F::store(&xs[0], F::mul_boundary(F::load(&xs[0]), F::load(&ys[0])));
                F::store(&xs[1], F::mul_boundary(F::load(&xs[1]), F::load(&ys[1])));
                x.x = xs[0];
                x.y = xs[1];
                __builtin_nontemporal_store(x, l2 + i);
            }
            done = pairs << 1;
        }
    }
    for (size_t i = done + gid; i < n; i += stride) F::store(lhs + i, F::mul_boundary(F::load(lhs + i), F::load(r + (i & dmask))));
}

// lhs[i] = lhs[i] +- rhs[i] coefficient-wise: RqNTT / RqPoly Add and Sub (ntt_form.rs:227-285, 588-638; coeff_form.rs
// operator impls) -- the same in either form and for every ring, the slots being Fp-vector spaces
template <class F, bool SUB>
__global__ __launch_bounds__(256) void addsub_kernel(typename F::storage *lhs, const typename F::storage *rhs, size_t n) {
    elementwise2<F>(lhs, rhs, n, [](const typename F::elem &a, const typename F::elem &b) { return SUB ? F::sub(a, b) : F::add(a, b); });
}
// the unary form: data[i] = op(data[i]), same access shape
template <class F, class Op>
__device__ __forceinline__ void elementwise1(typename F::storage *data, size_t n, Op op) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if constexpr (sizeof(typename F::storage) == 8) {
        if (((uintptr_t)data & 15u) == 0) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            const size_t pairs = n >> 1;
            u64x2 *d2 = reinterpret_cast<u64x2 *>(data);
            for (size_t i = gid; i < pairs; i += stride) {
                u64x2 x = __builtin_nontemporal_load(d2 + i);
                typename F::storage xs[2] = {x.x, x.y};
                F::store(&xs[0], op(F::load(&xs[0])));
                F::store(&xs[1], op(F::load(&xs[1])));
                x.x = xs[0];
                x.y = xs[1];
                __builtin_nontemporal_store(x, d2 + i);
            }
            done = pairs << 1;
        }
    }
    for (size_t i = done + gid; i < n; i += stride) F::store(data + i, op(F::load(data + i)));
}
// data[i] = -data[i]: Neg of RqPoly / RqNTT (coeff_form.rs:270-278 `self.0.map(|x| -x)`, ntt_form.rs:191-203) -- the same word-wise
// map in either form and for every ring (an Fq3 / Fq9 / Fq4 slot negates component-wise)
template <class F>
__global__ __launch_bounds__(256) void neg_kernel(typename F::storage *data, size_t n) {
    elementwise1<F>(data, n, [](const typename F::elem &a) { return F::neg(a); });
}
// data[i] = data[i] * s on the in-memory images, s one base-field scalar (its memory image, a kernel argument): Mul<Fp> =
// poly_mul(from_scalar(rhs)) and Mul<u128 | u64 | ... | bool> / MulAssign of RqPoly (coeff_form.rs:390-408, 610-650:
// `self.0.iter_mut().for_each(|lhs| *lhs *= r)`) and of RqNTT (ntt_form.rs:373-425: `*lhs *= BaseCRTField::from(rhs)` -- a
// base-field element embedded in Fq3 / Fq9 / Fq4 is (r, 0, ..), so the slot product scales every component by r)
template <class F>
__global__ __launch_bounds__(256) void scale_kernel(typename F::storage *data, size_t n, typename F::storage scalar) {
    const typename F::elem s = F::load(&scalar);
    elementwise1<F>(data, n, [s](const typename F::elem &a) { return F::mul_boundary(a, s); });
}
// data[i * stride] += s for i < count: Add<primitive> / Sub<primitive> (the caller negates).  RqPoly: `self.0[0] += Fp::from(rhs)`,
// stride = D words, one per ring element (coeff_form.rs:652-700); RqNTT: `*lhs += BaseCRTField::from(rhs)` for every slot, i.e.
// component 0 of every slot, stride = the slot's extension degree (ntt_form.rs:427-505)
template <class F>
__global__ __launch_bounds__(256) void add_scalar_kernel(typename F::storage *data, size_t count, size_t stride, typename F::storage scalar) {
    const typename F::elem s = F::load(&scalar);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        typename F::storage *p = data + i * stride;
        F::store(p, F::add(F::load(p), s));
    }
}
// One stage of `Sum` / `Product` over a slice of ring elements (coeff_form.rs:507-537, ntt_form.rs:640-670: `iter.fold(zero(), acc + x)`,
// `iter.fold(one(), acc * x)`): n elements of w words each are folded word-wise into r partial elements.  Lane g < r * w folds the
// words g, g + r w, g + 2 r w, ... of the flat input (all of them word g mod w of some element), so a wave reads consecutive words;
// four independent accumulators keep four loads in flight.  The fold is associative and commutative, and field results are
// canonical, so the order changes nothing.  MUL: the fully split rings' slot product on the memory images (mul_boundary).
template <class F, bool MUL>
__global__ __launch_bounds__(256) void fold_stage_kernel(typename F::storage *out, const typename F::storage *in, size_t w, size_t n, size_t r) {
    using E = typename F::elem;
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x, t = r * w, total = n * w;
    if (g >= t) return;
    auto op = [](const E &a, const E &b) { return MUL ? F::mul_boundary(a, b) : F::add(a, b); };
    E acc[4];
    bool have[4] = {false, false, false, false};
    size_t i = g;
    for (; i + 3 * t < total; i += 4 * t) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const E x = F::load(in + i + u * t);
            acc[u] = have[u] ? op(acc[u], x) : x;
            have[u] = true;
        }
    }
    for (int u = 0; i < total; i += t, u++) {
        const E x = F::load(in + i);
        acc[u] = have[u] ? op(acc[u], x) : x;
        have[u] = true;
    }
    // lane g < t <= n w always owns word g itself, so acc[0] is set
    E s = acc[0];
#pragma unroll
    for (int u = 1; u < 4; u++)
        if (have[u]) s = op(s, acc[u]);
    F::store(out + g, s);
}
// workgroups for a streaming kernel over n coefficients: one 16-byte access per lane where the field allows it (no grid-stride
// loop below 2^24 workgroups (HIP limits a launch to 2^32 lanes): a lane that loops streams measurably worse, see elementwise2)
template <class F>
inline unsigned stream_blocks(size_t n) {
    const size_t per_lane = sizeof(typename F::storage) == 8 ? 2 : 1;
    size_t blocks = ((n + per_lane - 1) / per_lane + 255) / 256;
    if (blocks > 0xFFFFFFull) blocks = 0xFFFFFFull;  // gridDim.x * 256 must stay below 2^32
    return (unsigned)(blocks ? blocks : 1);
}

// sum_i a_i b_i on memory images (== sum_i mul_boundary(a_i, b_i)) for the linear-algebra kernels below.
// Generic form: sum of pre() terms, one post(); lazy fields (Stark) reduce the uncarried sum weakly every four terms.
template <class F>
struct SumOfProducts {
    typename F::elem acc;
    int since;
    SR_HD void init() {
        acc = F::zero();
        since = 0;
    }
    SR_HD void fma(const typename F::elem &a, const typename F::elem &b) {
        acc = F::add(acc, F::mul_boundary_pre(a, b));
        if (Lazy<F>::value && ++since == 4) {
            acc = Lazy<F>::weak(acc);
            since = 0;
        }
    }
    SR_HD typename F::elem finish() const { return F::boundary_post(acc); }
};
// Goldilocks: no reduction inside the sum at all.  With a = a0 + a1 2^32, b = b0 + b1 2^32 the four 64-bit partial products
// a_i b_j accumulate in four 96-bit integers (a 64-bit sum and a count of its carries: room for 2^32 terms), 12 VALU per term
// against 30 for mul_boundary + add; finish() reduces each sum once and combines
//     (S00 + (S01 + S10) 2^32 + S11 2^64) 2^-64 = S00 2^-64 + (S01 + S10) 2^-32 + S11.
template <>
struct SumOfProducts<Goldilocks> {
    uint64_t lo[4];
    uint32_t hi[4];
    SR_HD void init() {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo[i] = 0;
            hi[i] = 0;
        }
    }
    SR_HD void term(int i, uint32_t x, uint32_t y) {
        const uint64_t t = lo[i] + (uint64_t)x * y;
        hi[i] += t < lo[i];
        lo[i] = t;
    }
    SR_HD void fma(uint64_t a, uint64_t b) {
        const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
        term(0, a0, b0);
        term(1, a0, b1);
        term(2, a1, b0);
        term(3, a1, b1);
    }
    SR_HD uint64_t finish() const {
        using G = Goldilocks;
        const uint64_t s00 = G::reduce128(lo[0], hi[0]), s01 = G::reduce128(lo[1], hi[1]), s10 = G::reduce128(lo[2], hi[2]),
                       s11 = G::reduce128(lo[3], hi[3]);
        const uint64_t inv32 = 0xFFFFFFFE00000002ull;  // 2^-32 mod p = p - (2^32 - 1): 2^32 (2^32 - 1) = 2^64 - 2^32 = -1
        // x 2^-64 = mul_boundary(x, 1); x 2^-32 = mul(x, 2^-32)
        return G::add(G::add(G::mul_boundary(s00, 1), G::mul(G::add(s01, s10), inv32)), s11);
    }
};

// y[r] = sum_c M[r][c] * v[c] over ring elements in CRT/NTT form of a fully split ring (slot-wise Fp products and sums):
// Matrix<RqNTT>::checked_mul_vec (crates/linear_algebra/src/matrix.rs:168-178), one fused pass over M.
// Lane = one slot; a workgroup row-block of RB rows shares each v[c] slot it loads.  M is streamed once (HBM-bound).
template <class F, int RB>
__global__ __launch_bounds__(256) void matvec_kernel(typename F::storage *y, const typename F::storage *m,
                                                     const typename F::storage *v, size_t nrows, size_t ncols, int k) {
    const size_t d = (size_t)1 << k;
    const size_t chunks = (d + 255) >> 8;  // flat grid: row-block major, slot chunk minor (no 65535 limit on the rows)
    const size_t slot = (blockIdx.x % chunks) * (size_t)blockDim.x + threadIdx.x;
    const size_t r0 = (blockIdx.x / chunks) * RB;
    if (slot >= d) return;
    SumOfProducts<F> acc[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) acc[r].init();
    for (size_t c = 0; c < ncols; c++) {
        const typename F::elem x = F::load(v + (c << k) + slot);
#pragma unroll
        for (int r = 0; r < RB; r++)
            if (r0 + r < nrows) acc[r].fma(F::load(m + (((r0 + r) * ncols + c) << k) + slot), x);
    }
#pragma unroll
    for (int r = 0; r < RB; r++)
        if (r0 + r < nrows) F::store(y + ((r0 + r) << k) + slot, acc[r].finish());
}

// y[r] = sum over the stored entries (val, col) of row r of val * v[col]: SparseMatrix<RqNTT>::checked_mul_vec
// (crates/linear_algebra/src/sparse_matrix.rs:201-211; an empty row sums to zero).  CSR on the device: vals[j] is one ring
// element, cols[j] its column, row_ptr[r] .. row_ptr[r + 1] the entries of row r.  Lane = one slot, blockIdx.y = row.
// An entry whose column is >= ncols would index outside v (the reference panics there): it is skipped and counted in *bad.
template <class F>
__global__ __launch_bounds__(256) void spmv_kernel(typename F::storage *y, const typename F::storage *vals, const uint32_t *cols,
                                                   const uint64_t *row_ptr, const typename F::storage *v, size_t ncols, int k,
                                                   unsigned long long *bad) {
    const size_t d = (size_t)1 << k;
    const size_t chunks = (d + 255) >> 8;  // flat grid: row major, slot chunk minor
    const size_t slot = (blockIdx.x % chunks) * (size_t)blockDim.x + threadIdx.x;
    const size_t r = blockIdx.x / chunks;
    if (slot >= d) return;
    SumOfProducts<F> acc;
    acc.init();
    const uint64_t j1 = row_ptr[r + 1];
    for (uint64_t j = row_ptr[r]; j < j1; j++) {
        const uint32_t c = cols[j];
        if (c >= ncols) {
            if (slot == 0) atomicAdd(bad, 1ull);
            continue;
        }
        acc.fma(F::load(vals + (j << k) + slot), F::load(v + ((size_t)c << k) + slot));
    }
    F::store(y + (r << k) + slot, acc.finish());
}

// Y (n x p) = A (n x m) * B (m x p), dense row-major matrices of ring elements in CRT/NTT form:
// Matrix<RqNTT>::checked_mul_mat (crates/linear_algebra/src/matrix.rs:148-166).  Per slot this is a small Fp GEMM; a lane
// keeps an RB x CB block of outputs, so each loaded A and B slot feeds CB resp. RB multiply-adds.
template <class F, int RB, int CB>
__global__ __launch_bounds__(256) void matmul_kernel(typename F::storage *y, const typename F::storage *a,
                                                     const typename F::storage *b, size_t n, size_t m, size_t p, int k) {
    const size_t d = (size_t)1 << k;
    const size_t chunks = (d + 255) >> 8;  // flat grid: (row-block, column-block) major, slot chunk minor
    const size_t cblocks = (p + CB - 1) / CB;
    const size_t slot = (blockIdx.x % chunks) * (size_t)blockDim.x + threadIdx.x;
    const size_t tile = blockIdx.x / chunks;
    const size_t r0 = (tile / cblocks) * RB, c0 = (tile % cblocks) * CB;
    if (slot >= d) return;
    SumOfProducts<F> acc[RB][CB];
#pragma unroll
    for (int r = 0; r < RB; r++)
#pragma unroll
        for (int c = 0; c < CB; c++) acc[r][c].init();
    for (size_t t = 0; t < m; t++) {
        typename F::elem av[RB], bv[CB];
#pragma unroll
        for (int r = 0; r < RB; r++) av[r] = r0 + r < n ? F::load(a + (((r0 + r) * m + t) << k) + slot) : F::zero();
#pragma unroll
        for (int c = 0; c < CB; c++) bv[c] = c0 + c < p ? F::load(b + ((t * p + c0 + c) << k) + slot) : F::zero();
#pragma unroll
        for (int r = 0; r < RB; r++)
#pragma unroll
            for (int c = 0; c < CB; c++) acc[r][c].fma(av[r], bv[c]);
    }
#pragma unroll
    for (int r = 0; r < RB; r++)
#pragma unroll
        for (int c = 0; c < CB; c++)
            if (r0 + r < n && c0 + c < p) F::store(y + (((r0 + r) * p + c0 + c) << k) + slot, acc[r][c].finish());
}

// Cyclotomic::rot (crates/ring/src/traits.rs:54-66): out = X * in for every ring element of the batch.
// X^D + 1 (half = 0; stark_prime/mod.rs:87-95, frog_ring/mod.rs:126-134): out[0] = -in[D-1], out[i] = in[i-1];
// X^D - X^(D/2) + 1 (half = D/2; goldilocks/mod.rs:138-149, babybear/mod.rs:150-161): additionally out[D/2] += in[D-1].
template <class F>
__global__ void rot_kernel(typename F::storage *out, const typename F::storage *in, size_t d, size_t half, size_t batch) {
    const size_t n = batch * d;
    if constexpr (sizeof(typename F::storage) == 8) {
        // one-limb fields, even D and even D/2: a lane writes the pair (i, i + 1), i even, as one non-temporal 16-byte store from the
        // two 8-byte words in[i - 1], in[i] (the shift by one word makes the read side 8-byte aligned only)
        if ((d & 1) == 0 && (half & 1) == 0 && (((uintptr_t)out) & 15u) == 0) {
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            for (size_t t2 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t2 < (n >> 1); t2 += (size_t)gridDim.x * blockDim.x) {
                const size_t t = t2 << 1, e = t / d, i = t - e * d;
                const typename F::storage *src = in + e * d;
                typename F::elem v0 = i == 0 ? F::sub(F::zero(), F::load(src + d - 1)) : F::load(src + i - 1);
                if (half && i == half) v0 = F::add(v0, F::load(src + d - 1));
                const typename F::elem v1 = F::load(src + i);
                typename F::storage o[2];
                F::store(&o[0], v0);
                F::store(&o[1], v1);
                u64x2 ov;
                ov.x = o[0];
                ov.y = o[1];
                __builtin_nontemporal_store(ov, reinterpret_cast<u64x2 *>(out + t));
            }
            return;
        }
    }
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / d, i = t - e * d;
        const typename F::storage *src = in + e * d;
        typename F::elem v = i == 0 ? F::sub(F::zero(), F::load(src + d - 1)) : F::load(src + i - 1);
        if (half && i == half) v = F::add(v, F::load(src + d - 1));
        F::store(out + t, v);
    }
}

// out[e][i] = in[e][i] - in[e][D + i]   (stark_prime/mod.rs:40-47); in_len <= 2D per element.  Two coefficients (16 bytes) per
// lane for the one-limb fields when every row of `in` and `out` is 16-byte aligned (even in_len, D >= 2).
template <class F>
__global__ __launch_bounds__(256) void reduce_pow2_kernel(const typename F::storage *in, size_t in_len, typename F::storage *out,
                                                          int k, size_t batch) {
    const size_t d = (size_t)1 << k;
    const size_t n = batch << k;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    if constexpr (sizeof(typename F::storage) == 8) {
        if (k >= 1 && (in_len & 1) == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15u) == 0)) {
            for (size_t t2 = gid; t2 < (n >> 1); t2 += stride) {
                const size_t t = t2 << 1, e = t >> k, i = t & (d - 1);
                const typename F::storage *src = in + e * in_len;
                typename F::storage lo[2] = {0, 0}, hi[2] = {0, 0};
                if (i < in_len) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + i);
                    lo[0] = v.x;
                    lo[1] = v.y;
                }
                typename F::elem r0 = i < in_len ? F::load(&lo[0]) : F::zero(), r1 = i < in_len ? F::load(&lo[1]) : F::zero();
                if (d + i < in_len) {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + d + i);
                    hi[0] = v.x;
                    hi[1] = v.y;
                    r0 = F::sub(r0, F::load(&hi[0]));
                    r1 = F::sub(r1, F::load(&hi[1]));
                }
                typename F::storage o[2];
                F::store(&o[0], r0);
                F::store(&o[1], r1);
                ulonglong2 w;
                w.x = o[0];
                w.y = o[1];
                *reinterpret_cast<ulonglong2 *>(out + t) = w;
            }
            return;
        }
    }
    for (size_t t = gid; t < n; t += stride) {
        size_t e = t >> k, i = t & (d - 1);
        const typename F::storage *src = in + e * in_len;
        typename F::elem lo = i < in_len ? F::load(src + i) : F::zero();
        if (d + i < in_len) lo = F::sub(lo, F::load(src + d + i));
        F::store(out + t, lo);
    }
}

// tw[i] = psi^brv_k(i), itw[i] = psi^-brv_k(i) from psi^(2^j), psi^-(2^j)
template <class F>
__global__ void build_tables_kernel(typename F::elem *tw, typename F::elem *itw, int k,
                                    const typename F::elem *pows, const typename F::elem *ipows) {
    size_t d = (size_t)1 << k;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < d; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t r = bitrev((uint32_t)i, k);
        typename F::elem acc = F::tw_one(), iacc = F::tw_one();
        for (int j = 0; j < k; j++)
            if ((r >> j) & 1u) {
                acc = F::mul_tw(acc, pows[j]);
                iacc = F::mul_tw(iacc, ipows[j]);
            }
        tw[i] = Lazy<F>::table(acc);
        itw[i] = Lazy<F>::table(iacc);
    }
}

// ---- synthetic inputs: same counter-based definition as oracle/sr_oracle.c sro_fill_uniform ----
SR_HD uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
SR_HD uint64_t prng_word(uint64_t seed, uint64_t idx, unsigned limb, unsigned retry) {
    uint64_t x = mix64(seed + 0x9E3779B97F4A7C15ull * (idx + 1));
    return mix64(x ^ (0xD1B54A32D192ED03ull * (uint64_t)(limb + 4 * retry + 1)));
}
template <class F>
SR_HD void uniform_words(uint64_t seed, uint64_t idx, uint64_t *out);
template <>
SR_HD void uniform_words<Goldilocks>(uint64_t seed, uint64_t idx, uint64_t *out) {
    uint64_t v = 0;
    for (unsigned r = 0; r < 64; r++) {
        v = prng_word(seed, idx, 0, r);
        if (v < Goldilocks::P) break;
        v = 0;
    }
    out[0] = v;
}
template <>
SR_HD void uniform_words<Frog>(uint64_t seed, uint64_t idx, uint64_t *out) {
    uint64_t v = 0;
    for (unsigned r = 0; r < 64; r++) {
        v = prng_word(seed, idx, 0, r);
        if (v < Frog::P) break;
        v = 0;
    }
    out[0] = v;
}
template <>
SR_HD void uniform_words<BabyBear>(uint64_t seed, uint64_t idx, uint64_t *out) {
    uint64_t v = 0;
    for (unsigned r = 0; r < 64; r++) {
        v = prng_word(seed, idx, 0, r) & 0x7FFFFFFFull;
        if (v < BabyBear::P) break;
        v = 0;
    }
    out[0] = v;
}
template <>
SR_HD void uniform_words<Stark>(uint64_t seed, uint64_t idx, uint64_t *out) {
    for (unsigned r = 0; r < 64; r++) {
        uint64_t q[4];
        for (unsigned l = 0; l < 4; l++) q[l] = prng_word(seed, idx, l, r);
        q[3] &= 0x0FFFFFFFFFFFFFFFull;
        // q < p ?  p = {1, 0, 0, 0x0800000000000011}
        bool lt = q[3] < 0x0800000000000011ull ||
                  (q[3] == 0x0800000000000011ull && q[2] == 0 && q[1] == 0 && q[0] < 1);
        if (lt) {
            for (int l = 0; l < 4; l++) out[l] = q[l];
            return;
        }
    }
    for (int l = 0; l < 4; l++) out[l] = 0;
}
template <class F>
__global__ void fill_uniform_kernel(uint64_t seed, uint64_t first, size_t n, uint64_t *out) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t w[4];
        uniform_words<F>(seed, first + i, w);
        for (int l = 0; l < F::kStorageWords64; l++) out[i * F::kStorageWords64 + l] = w[l];
    }
}
template <class F>
__global__ void count_noncanonical_kernel(const typename F::storage *d, size_t n, unsigned long long *count) {
    unsigned long long bad = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (std::is_same<F, BabyBear>::value) {
            uint64_t raw = d[i];
            bad += (raw >= BabyBear::P);
        } else {
            bad += !F::valid(F::load(d + i));
        }
    }
    if (bad) atomicAdd(count, bad);
}

}  // namespace sr
