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

// One workgroup (256 lanes) per output ring element.  Lane t = (term group t / S, slot t % S) with S = D / W slots per element:
// consecutive lanes read consecutive slots, i.e. whole ring elements, so the streams of M (or A, B, the CSR values) are read
// with full cache lines; a lane sums every (256 / S)-th term of the inner dimension, and the partial sums of a slot meet in LDS.
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
// y[r] = sum_c M[r][c] * v[c]
template <class SL>
__global__ __launch_bounds__(256) void slot_matvec_kernel(typename SL::K k, uint64_t *y, const uint64_t *m, const uint64_t *v,
                                                          size_t nrows, size_t ncols) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 256 / S;
    static_assert(256 % S == 0 && SL::D <= 256, "slot layout");
    __shared__ E lds[256 * SL::W];
    const size_t r = blockIdx.x;
    const int g = threadIdx.x / S, s = threadIdx.x % S;
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int i = 0; i < SL::W; i++) acc[i] = SL::F::zero();
    for (size_t c = g; c < ncols; c += G) {
        slot_load<SL>(x, m + (r * ncols + c) * SL::D + s * SL::W);
        slot_load<SL>(z, v + c * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_reduce_store<SL>(lds, acc, y + r * SL::D);
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
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int i = 0; i < SL::W; i++) acc[i] = SL::F::zero();
    const uint64_t j1 = row_ptr[r + 1];
    for (uint64_t j = row_ptr[r] + g; j < j1; j += G) {
        const uint32_t c = cols[j];
        if (c >= ncols) {
            if (s == 0) atomicAdd(bad, 1ull);
            continue;
        }
        slot_load<SL>(x, vals + j * SL::D + s * SL::W);
        slot_load<SL>(z, v + (size_t)c * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_reduce_store<SL>(lds, acc, y + r * SL::D);
}
// Y (n x p) = A (n x m) * B (m x p); one workgroup per output element (i, j)
template <class SL>
__global__ __launch_bounds__(256) void slot_matmul_kernel(typename SL::K k, uint64_t *y, const uint64_t *a, const uint64_t *b,
                                                          size_t n, size_t m, size_t p) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W, G = 256 / S;
    __shared__ E lds[256 * SL::W];
    const size_t i = blockIdx.x / p, j = blockIdx.x % p;
    const int g = threadIdx.x / S, s = threadIdx.x % S;
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int q = 0; q < SL::W; q++) acc[q] = SL::F::zero();
    for (size_t t = g; t < m; t += G) {
        slot_load<SL>(x, a + (i * m + t) * SL::D + s * SL::W);
        slot_load<SL>(z, b + (t * p + j) * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_reduce_store<SL>(lds, acc, y + (i * p + j) * SL::D);
}

template <class SL>
inline int slot_matvec(const typename SL::K &k, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols,
                       hipStream_t st) {
    const size_t blocks = nrows;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_matvec_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, m, v, nrows, ncols);
    return hipGetLastError() != hipSuccess;
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
    const size_t blocks = n * p;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_matmul_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, a, b, n, m, p);
    return hipGetLastError() != hipSuccess;
}

}  // namespace sr
