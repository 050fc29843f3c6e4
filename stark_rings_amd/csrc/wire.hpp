// ark-serialize wire format of ring elements, on device -- SURVEY 8f #3.
//
//   RqPoly  = [Fp; D]              crates/ring/src/cyclotomic_ring/coeff_form.rs:154-189
//   RqNTT   = [BaseCRTField; N]    crates/ring/src/cyclotomic_ring/ntt_form.rs:24 (derive); Fq3 / Fq9 slots serialise their base
//                                  components in order, so a ring element is its flat coefficient array either way
//   Vec / Matrix / SparseMatrix framing (u64 little-endian lengths, (R, usize) pairs) is host-side:
//                                  crates/linear_algebra/src/matrix.rs:111-145, sparse_matrix.rs:158-200
//
// The per-coefficient format is ark-ff 0.4.2's `Fp::serialize_with_flags` with `EmptyFlags` (third party, Cargo.lock:59-62, source
// not in the tree; restated from the published algorithm): the standard-form integer (`into_bigint`, i.e. out of Montgomery form)
// as ceil(MODULUS_BIT_SIZE / 8) little-endian bytes -- 8 for Goldilocks and the frog prime, 4 for BabyBear, 32 for Stark.
// `deserialize_with_flags` reads the same bytes and rejects an integer >= p (`from_bigint` -> None -> InvalidData).
// PARITY UNPINNED for the byte layout: the reference holds no serialised golden bytes; the Montgomery <-> standard conversion
// underneath is pinned by every KAT (they are stated in standard form).
//
// One lane = one coefficient.  Memory-bound: D (8 + W) bytes per element (8 = memory image, W = wire bytes per coefficient).
// `offsets` (optional, device): byte offset of element e inside the wire buffer, so callers can interleave their own framing
// words (row lengths, column indices); nullptr = densely packed.  Offsets must be multiples of 8 (of 4 for BabyBear, whose
// coefficients are 4 bytes on the wire); an element with a misaligned offset is skipped and counted with the invalid coefficients.
#pragma once
#include <type_traits>

#include "decompose.hpp"
#include "fields.hpp"

namespace sr {
namespace wire {

template <class F>
struct Codec {  // Fp64 with an 8-byte wire image: Goldilocks, Frog
    static constexpr int W = 8;
    SR_HD static void put(uint8_t *dst, typename F::elem img) {
        *reinterpret_cast<uint64_t *>(dst) = (uint64_t)F::mul_boundary(img, dec::Consts<F>::one());
    }
    SR_HD static bool get(typename F::elem &img, const uint8_t *src) {
        const uint64_t v = *reinterpret_cast<const uint64_t *>(src);
        const bool ok = v < (uint64_t)F::P;
        img = F::mul_boundary(ok ? v : 0, dec::Consts<F>::r2());
        return ok;
    }
};
template <>
struct Codec<BabyBear> {
    static constexpr int W = 4;
    SR_HD static void put(uint8_t *dst, uint32_t img) {
        *reinterpret_cast<uint32_t *>(dst) = BabyBear::mul_boundary(img, 1u);
    }
    SR_HD static bool get(uint32_t &img, const uint8_t *src) {
        const uint32_t v = *reinterpret_cast<const uint32_t *>(src);
        const bool ok = v < BabyBear::P;
        img = BabyBear::mul_boundary(ok ? v : 0u, dec::Consts<BabyBear>::r2());
        return ok;
    }
};
template <>
struct Codec<Stark> {
    static constexpr int W = 32;
    SR_HD static void put(uint8_t *dst, const U256 &img) {
        const U256 v = Stark::mul_boundary(img, dec::Consts<Stark>::one());
        uint64_t *o = reinterpret_cast<uint64_t *>(dst);
#pragma unroll
        for (int i = 0; i < 4; i++) o[i] = (uint64_t)v.l[2 * i] | ((uint64_t)v.l[2 * i + 1] << 32);
    }
    SR_HD static bool get(U256 &img, const uint8_t *src) {
        const uint64_t *s = reinterpret_cast<const uint64_t *>(src);
        U256 v;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t q = s[i];
            v.l[2 * i] = (uint32_t)q;
            v.l[2 * i + 1] = (uint32_t)(q >> 32);
        }
        const bool ok = !Stark::geq_p(v);
        if (!ok) v = Stark::zero();
        img = Stark::mul_boundary(v, dec::Consts<Stark>::r2());
        return ok;
    }
};

template <class F>
__global__ __launch_bounds__(256) void serialize_kernel(uint8_t *out, const typename F::storage *in, size_t d, size_t batch,
                                                        const uint64_t *offsets, unsigned long long *bad) {
    constexpr size_t W = Codec<F>::W, ALIGN = W < 8 ? W : 8;
    const size_t n = batch * d;
    if constexpr (std::is_same<F, BabyBear>::value) {
        if (!offsets && !(n & 1) && !(((uintptr_t)in & 15) | ((uintptr_t)out & 7))) {  // dense BabyBear: two coefficients per lane, 16 B in, 8 B out
            const uint4 *src = reinterpret_cast<const uint4 *>(in);
            uint2 *dst = reinterpret_cast<uint2 *>(out);
            for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n / 2; t += (size_t)gridDim.x * blockDim.x) {
                const uint4 v = src[t];
                dst[t] = make_uint2(BabyBear::mul_boundary(v.x, 1u), BabyBear::mul_boundary(v.z, 1u));
            }
            return;
        }
    }
    if constexpr (W == 8 && sizeof(typename F::storage) == 8) {
        if (!offsets && !(n & 1) && !((((uintptr_t)in) | ((uintptr_t)out)) & 15)) {  // dense 8-byte fields: two coefficients per lane, 16 B each way
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(in);
            for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n / 2; t += (size_t)gridDim.x * blockDim.x) {
                const ulonglong2 v = src[t];
                typename F::storage w[2] = {v.x, v.y};
                alignas(16) uint8_t o[16];
                Codec<F>::put(o, F::load(&w[0]));
                Codec<F>::put(o + 8, F::load(&w[1]));
                reinterpret_cast<ulonglong2 *>(out)[t] = *reinterpret_cast<const ulonglong2 *>(o);
            }
            return;
        }
    }
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / d, i = t - e * d;
        const size_t base = offsets ? (size_t)offsets[e] : e * d * W;
        if (offsets && (base & (ALIGN - 1))) {  // the dense layout is aligned by construction
            if (i == 0) atomicAdd(bad, 1ull);
            continue;
        }
        Codec<F>::put(out + base + i * W, F::load(in + t));
    }
}

// a coefficient >= p is counted in *bad and read as 0 (the caller turns a non-zero count into InvalidData)
template <class F>
__global__ __launch_bounds__(256) void deserialize_kernel(typename F::storage *out, const uint8_t *in, size_t d, size_t batch,
                                                          const uint64_t *offsets, unsigned long long *bad) {
    constexpr size_t W = Codec<F>::W, ALIGN = W < 8 ? W : 8;
    const size_t n = batch * d;
    if constexpr (std::is_same<F, BabyBear>::value) {
        if (!offsets && !(n & 1) && !(((uintptr_t)in & 7) | ((uintptr_t)out & 15))) {  // dense BabyBear: two coefficients per lane, 8 B in, 16 B out
            const uint2 *src = reinterpret_cast<const uint2 *>(in);
            uint4 *dst = reinterpret_cast<uint4 *>(out);
            unsigned long long nbad = 0;
            for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n / 2; t += (size_t)gridDim.x * blockDim.x) {
                const uint2 v = src[t];
                const bool ok0 = v.x < BabyBear::P, ok1 = v.y < BabyBear::P;
                nbad += !ok0 + !ok1;
                dst[t] = make_uint4(BabyBear::mul_boundary(ok0 ? v.x : 0u, dec::Consts<BabyBear>::r2()), 0u,
                                    BabyBear::mul_boundary(ok1 ? v.y : 0u, dec::Consts<BabyBear>::r2()), 0u);
            }
            if (nbad) atomicAdd(bad, nbad);
            return;
        }
    }
    if constexpr (W == 8 && sizeof(typename F::storage) == 8) {
        if (!offsets && !(n & 1) && !((((uintptr_t)in) | ((uintptr_t)out)) & 15)) {  // dense 8-byte fields: two coefficients per lane
            unsigned long long nbad = 0;
            for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n / 2; t += (size_t)gridDim.x * blockDim.x) {
                alignas(16) uint8_t w[16];
                *reinterpret_cast<ulonglong2 *>(w) = reinterpret_cast<const ulonglong2 *>(in)[t];
                typename F::elem i0, i1;
                nbad += !Codec<F>::get(i0, w);
                nbad += !Codec<F>::get(i1, w + 8);
                typename F::storage o[2];
                F::store(&o[0], i0);
                F::store(&o[1], i1);
                ulonglong2 ov;
                ov.x = o[0];
                ov.y = o[1];
                reinterpret_cast<ulonglong2 *>(out)[t] = ov;
            }
            if (nbad) atomicAdd(bad, nbad);
            return;
        }
    }
    for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
        const size_t e = t / d, i = t - e * d;
        const size_t base = offsets ? (size_t)offsets[e] : e * d * W;
        if (offsets && (base & (ALIGN - 1))) {  // the dense layout is aligned by construction
            if (i == 0) atomicAdd(bad, 1ull);
            F::store(out + t, F::zero());
            continue;
        }
        typename F::elem img;
        if (!Codec<F>::get(img, in + base + i * W)) atomicAdd(bad, 1ull);
        F::store(out + t, img);
    }
}

}  // namespace wire
}  // namespace sr
