// Linear algebra over the reference's OWN RqNTT types (finished in round 2): the only concrete
// CyclotomicPolyRingNTTGeneral the reference ships have extension-field slots -- Goldilocks-24: 8 x Fq3
// (goldilocks/mod.rs:69-119), BabyBear-72: 8 x Fq9 (babybear/mod.rs:81-131), Frog-16: 4 x Fq4 (frog_ring/mod.rs:62-107) -- and
// Matrix<R>::checked_mul_vec / checked_mul_mat (crates/linear_algebra/src/matrix.rs:148-178) and
// SparseMatrix<R>::checked_mul_vec (sparse_matrix.rs:201-212) are generic over R.  Here a lane owns ONE slot (W consecutive
// coefficients of one ring element): it multiplies slots with the ring's slot product (ntt_form.rs:159-189 through
// small_slot_mul / frog_fq4_mul, the very code the fused ring products use) and sums them coefficient-wise.
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

// y[r] = sum_c M[r][c] * v[c]; lane = (row r, slot s), slots of a row are consecutive lanes
template <class SL>
__global__ __launch_bounds__(256) void slot_matvec_kernel(typename SL::K k, uint64_t *y, const uint64_t *m, const uint64_t *v,
                                                          size_t nrows, size_t ncols) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= nrows * S) return;
    const size_t r = gid / S, s = gid % S;
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int i = 0; i < SL::W; i++) acc[i] = SL::F::zero();
    for (size_t c = 0; c < ncols; c++) {
        slot_load<SL>(x, m + (r * ncols + c) * SL::D + s * SL::W);
        slot_load<SL>(z, v + c * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_store<SL>(y + r * SL::D + s * SL::W, acc);
}
// CSR sparse matrix times vector; an entry with column >= ncols is skipped and counted once (the reference panics there)
template <class SL>
__global__ __launch_bounds__(256) void slot_spmv_kernel(typename SL::K k, uint64_t *y, const uint64_t *vals, const uint32_t *cols,
                                                        const uint64_t *row_ptr, const uint64_t *v, size_t nrows, size_t ncols,
                                                        unsigned long long *bad) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= nrows * S) return;
    const size_t r = gid / S, s = gid % S;
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int i = 0; i < SL::W; i++) acc[i] = SL::F::zero();
    const uint64_t j1 = row_ptr[r + 1];
    for (uint64_t j = row_ptr[r]; j < j1; j++) {
        const uint32_t c = cols[j];
        if (c >= ncols) {
            if (s == 0) atomicAdd(bad, 1ull);
            continue;
        }
        slot_load<SL>(x, vals + j * SL::D + s * SL::W);
        slot_load<SL>(z, v + (size_t)c * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_store<SL>(y + r * SL::D + s * SL::W, acc);
}
// Y (n x p) = A (n x m) * B (m x p); lane = (i, j, slot)
template <class SL>
__global__ __launch_bounds__(256) void slot_matmul_kernel(typename SL::K k, uint64_t *y, const uint64_t *a, const uint64_t *b,
                                                          size_t n, size_t m, size_t p) {
    using E = typename SL::F::elem;
    constexpr int S = SL::D / SL::W;
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (gid >= n * p * S) return;
    const size_t s = gid % S, ij = gid / S, i = ij / p, j = ij % p;
    E acc[SL::W], x[SL::W], z[SL::W];
#pragma unroll
    for (int q = 0; q < SL::W; q++) acc[q] = SL::F::zero();
    for (size_t t = 0; t < m; t++) {
        slot_load<SL>(x, a + (i * m + t) * SL::D + s * SL::W);
        slot_load<SL>(z, b + (t * p + j) * SL::D + s * SL::W);
        slot_fma<SL>(acc, x, z, k);
    }
    slot_store<SL>(y + (i * p + j) * SL::D + s * SL::W, acc);
}

template <class SL>
inline int slot_matvec(const typename SL::K &k, uint64_t *y, const uint64_t *m, const uint64_t *v, size_t nrows, size_t ncols,
                       hipStream_t st) {
    const size_t lanes = nrows * (SL::D / SL::W), blocks = (lanes + 255) / 256;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_matvec_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, m, v, nrows, ncols);
    return hipGetLastError() != hipSuccess;
}
template <class SL>
inline int slot_spmv(const typename SL::K &k, uint64_t *y, const uint64_t *vals, const uint32_t *cols, const uint64_t *row_ptr,
                     const uint64_t *v, size_t nrows, size_t ncols, unsigned long long *bad, hipStream_t st) {
    const size_t lanes = nrows * (SL::D / SL::W), blocks = (lanes + 255) / 256;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_spmv_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, vals, cols, row_ptr, v, nrows, ncols, bad);
    return hipGetLastError() != hipSuccess;
}
template <class SL>
inline int slot_matmul(const typename SL::K &k, uint64_t *y, const uint64_t *a, const uint64_t *b, size_t n, size_t m, size_t p,
                       hipStream_t st) {
    const size_t lanes = n * p * (SL::D / SL::W), blocks = (lanes + 255) / 256;
    if (blocks == 0) return 0;
    if (blocks > 0x7FFFFFFFull) return 1;
    hipLaunchKernelGGL((slot_matmul_kernel<SL>), dim3((unsigned)blocks), dim3(256), 0, st, k, y, a, b, n, m, p);
    return hipGetLastError() != hipSuccess;
}

}  // namespace sr
