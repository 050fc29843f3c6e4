// Packed-u32 boundary for the BabyBear power-of-two rings (opt-in; BASELINE configs[2]: "packed 32-bit modmul").
//
// The reference stores a BabyBear coefficient as an ark-ff Fp64: ONE u64 limb holding a * 2^64 mod p, p = 15 * 2^27 + 1 < 2^31
// (crates/ring/src/cyclotomic_ring/models/babybear/mod.rs:18-26), so the upper 32 bits of every word in memory are zero.  The
// packed image of a coefficient is exactly the LOW HALF of that limb: the u32 (a * 2^64 mod p), canonical in [0, p) -- the same
// Montgomery residue (R = 2^64, NOT 2^32), four bytes instead of eight.  pack32 / unpack32 convert between the two images; the
// sr_*_packed32_* entry points compute on packed operands directly (the register-tiled kernels of ntt_regtile.hpp read and write
// 32-bit words at the boundary as they already do in their scratch), which halves the HBM bytes of the column passes.
#pragma once
#include "fields.hpp"

namespace sr {
namespace p32 {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// four coefficients per lane: two 16-byte reads, one 16-byte write (and the mirror image); tails one by one
__global__ __launch_bounds__(256) void pack32_kernel(uint32_t *out, const uint64_t *in, size_t n) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if ((((uintptr_t)out | (uintptr_t)in) & 15u) == 0) {
        const size_t quads = n >> 2;
        for (size_t i = gid; i < quads; i += stride) {
            const u64x2 lo = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in) + 2 * i);
            const u64x2 hi = __builtin_nontemporal_load(reinterpret_cast<const u64x2 *>(in) + 2 * i + 1);
            u32x4 v;
            v.x = (uint32_t)lo.x;
            v.y = (uint32_t)lo.y;
            v.z = (uint32_t)hi.x;
            v.w = (uint32_t)hi.y;
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out) + i);
        }
        done = quads << 2;
    }
    for (size_t i = done + gid; i < n; i += stride) out[i] = (uint32_t)in[i];
}
__global__ __launch_bounds__(256) void unpack32_kernel(uint64_t *out, const uint32_t *in, size_t n) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if ((((uintptr_t)out | (uintptr_t)in) & 15u) == 0) {
        const size_t quads = n >> 2;
        for (size_t i = gid; i < quads; i += stride) {
            const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in) + i);
            u64x2 lo, hi;
            lo.x = v.x;
            lo.y = v.y;
            hi.x = v.z;
            hi.y = v.w;
            __builtin_nontemporal_store(lo, reinterpret_cast<u64x2 *>(out) + 2 * i);
            __builtin_nontemporal_store(hi, reinterpret_cast<u64x2 *>(out) + 2 * i + 1);
        }
        done = quads << 2;
    }
    for (size_t i = done + gid; i < n; i += stride) out[i] = in[i];
}
// OP 0: slot product on the images (RqNTT MulAssign, ntt_form.rs:213-225), 1: add, 2: sub -- four packed coefficients per lane
template <int OP>
__device__ __forceinline__ uint32_t op32(uint32_t a, uint32_t b) {
    return OP == 0 ? BabyBear::mul_boundary(a, b) : (OP == 1 ? BabyBear::add(a, b) : BabyBear::sub(a, b));
}
template <int OP>
__global__ __launch_bounds__(256) void elementwise32_kernel(uint32_t *lhs, const uint32_t *rhs, size_t n) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    size_t done = 0;
    if ((((uintptr_t)lhs | (uintptr_t)rhs) & 15u) == 0) {
        const size_t quads = n >> 2;
        for (size_t i = gid; i < quads; i += stride) {
            u32x4 x = __builtin_nontemporal_load(reinterpret_cast<u32x4 *>(lhs) + i);
            const u32x4 y = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rhs) + i);
            x.x = op32<OP>(x.x, y.x);
            x.y = op32<OP>(x.y, y.y);
            x.z = op32<OP>(x.z, y.z);
            x.w = op32<OP>(x.w, y.w);
            __builtin_nontemporal_store(x, reinterpret_cast<u32x4 *>(lhs) + i);
        }
        done = quads << 2;
    }
    for (size_t i = done + gid; i < n; i += stride) lhs[i] = op32<OP>(lhs[i], rhs[i]);
}
inline unsigned blocks_for(size_t n) {
    size_t blocks = ((n + 3) / 4 + 255) / 256;
    if (blocks > 0xFFFFFFull) blocks = 0xFFFFFFull;
    return (unsigned)(blocks ? blocks : 1);
}

}  // namespace p32
}  // namespace sr
