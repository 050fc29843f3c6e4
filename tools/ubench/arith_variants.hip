// Register-only micro-benchmark of the two formulations VERDICT r2 asked to be BUILT and timed next to the current arithmetic
// (DESIGN_APPENDIX.md A.3, round 3): the same sixteen radix-2 stages of one transform's worth of butterflies and table products per
// coefficient, with every memory access taken out (operands live in registers for `reps` rounds; table words come from an
// L2-resident array like the real twiddles), so that only the instruction streams and their register footprints differ.
//
//   V0  current:      16 coefficients per lane, canonical u64:  D16  G  D16  G  D16  G  D16          (3 general layers)
//   V1  radix-64:     64 coefficients per lane, canonical u64:  D32 D32  G  D64  G  D32 D32          (2 general layers, 128 data VGPRs)
//   V2  2^96+1 limbs: 16 coefficients per lane as four signed 24-bit limbs modulo 2^96 + 1 (p divides it), conversions at both
//                     ends only (best case):                     in  D16' G' D16' G' D16' G' D16'  out
//                     D16': additions are four 32-bit adds, shifts by multiples of 24 are limb rotations (free), the four 12-bit
//                     shifts of a DFT_16 are three instructions per limb; G': 16 v_mad_i64_i32 and a carry sweep
//
// D_N = cyclic DFT_N out of compile-time shift butterflies (ntt_goldilocks.hpp: bf_dif).  G = one general product per coefficient
// (Goldilocks::mul).  V1 and V2 compute the same functions as V0 on their own data layout; V2's results are compared with V0's on
// the device (limb arithmetic is easy to get wrong).  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o arith_variants tools/ubench/arith_variants.hip && ./arith_variants
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o av.s tools/ubench/arith_variants.hip && python tools/isa_count.py av.s variant
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
using namespace sr::gl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// ---- generic cyclic DFT_N, N = 16 / 32 / 64, natural order in, bit-reversed out: omega_N = 2^(192 / N) ----
template <int N> SR_HD void dft_fwd(u64 *x);
template <> SR_HD void dft_fwd<16>(u64 *x) { dft16_fwd(x); }
template <> SR_HD void dft_fwd<32>(u64 *x) {
    dif_stage<16, 6>(x, std::make_integer_sequence<int, 1>{});
    dif_stage<8, 12>(x, std::make_integer_sequence<int, 2>{});
    dif_stage<4, 24>(x, std::make_integer_sequence<int, 4>{});
    dif_stage<2, 48>(x, std::make_integer_sequence<int, 8>{});
    dif_stage<1, 96>(x, std::make_integer_sequence<int, 16>{});
}
template <> SR_HD void dft_fwd<64>(u64 *x) {
    dif_stage<32, 3>(x, std::make_integer_sequence<int, 1>{});
    dif_stage<16, 6>(x, std::make_integer_sequence<int, 2>{});
    dif_stage<8, 12>(x, std::make_integer_sequence<int, 4>{});
    dif_stage<4, 24>(x, std::make_integer_sequence<int, 8>{});
    dif_stage<2, 48>(x, std::make_integer_sequence<int, 16>{});
    dif_stage<1, 96>(x, std::make_integer_sequence<int, 32>{});
}
template <int N>
__device__ __forceinline__ void general_layer(u64 *x, const u64 *__restrict__ tab, int lane) {
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = sr::Goldilocks::mul(x[i], tab[i * 64 + (lane & 63)]);
}

// V0: per lane 16 coefficients
__global__ __launch_bounds__(256, 4) void variant0_current(u64 *data, const u64 *__restrict__ tab, int reps) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = data[t * 16 + i];
    for (int r = 0; r < reps; r++) {
        dft_fwd<16>(x);
        general_layer<16>(x, tab, threadIdx.x);
        dft_fwd<16>(x);
        general_layer<16>(x, tab + 1024, threadIdx.x);
        dft_fwd<16>(x);
        general_layer<16>(x, tab + 2048, threadIdx.x);
        dft_fwd<16>(x);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) data[t * 16 + i] = x[i];
}
// V1: per lane 64 coefficients (two waves per SIMD at most: 128 data VGPRs)
__global__ __launch_bounds__(256, 2) void variant1_radix64(u64 *data, const u64 *__restrict__ tab, int reps) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u64 x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) x[i] = data[t * 64 + i];
    for (int r = 0; r < reps; r++) {
        dft_fwd<32>(x);
        dft_fwd<32>(x + 32);
        general_layer<64>(x, tab, threadIdx.x);
        dft_fwd<64>(x);
        general_layer<64>(x, tab + 4096, threadIdx.x);
        dft_fwd<32>(x);
        dft_fwd<32>(x + 32);
    }
#pragma unroll
    for (int i = 0; i < 64; i++) data[t * 64 + i] = x[i];
}

// ---- V2: four signed 24-bit limbs modulo M = 2^96 + 1 (p = 2^64 - 2^32 + 1 divides M) ----
struct L4 {
    int32_t l[4];
};
SR_HD L4 to_limbs(u64 x) {
    L4 r;
    r.l[0] = (int32_t)(x & 0xFFFFFFu);
    r.l[1] = (int32_t)((x >> 24) & 0xFFFFFFu);
    r.l[2] = (int32_t)(x >> 48);
    r.l[3] = 0;
    return r;
}
// value mod p from lazy limbs (|l_i| < 2^31): v = l0 + l1 2^24 + l2 2^48 + l3 2^72
SR_HD u64 from_limbs(const L4 &a) {
    using G = sr::Goldilocks;
    // lo = l0 + l1 2^24 + l2 2^48 as a signed 128-bit integer; hi3 = l3 2^72 = (l3 2^8) 2^64 = (l3 2^8) eps
    const __int128 lo = (__int128)a.l[0] + ((__int128)a.l[1] << 24) + ((__int128)a.l[2] << 48);
    const bool neg = lo < 0;
    const unsigned __int128 mag = neg ? (unsigned __int128)(-lo) : (unsigned __int128)lo;   // < 2^80
    u64 r = G::reduce128((u64)mag, (u64)(mag >> 64));
    if (neg) r = G::neg(r);
    const int64_t t3 = (int64_t)a.l[3] * 256;                                                // |.| < 2^39
    const u64 m3 = (u64)(t3 < 0 ? -t3 : t3) * (u64)G::EPS;                                   // < 2^71?  no: 2^39 * 2^32 = 2^71 overflows
    (void)m3;
    // l3 2^72 mod p = (l3 2^8) (2^32 - 1): at most 2^71, so split: (|t3| << 32) - |t3| as 128-bit, reduced
    const u64 at = (u64)(t3 < 0 ? -t3 : t3);
    const unsigned __int128 w = ((unsigned __int128)at << 32) - at;
    u64 r3 = G::reduce128((u64)w, (u64)(w >> 64));
    if (t3 < 0) r3 = G::neg(r3);
    return G::add(r, r3);
}
SR_HD L4 ladd(const L4 &a, const L4 &b) {
    L4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
SR_HD L4 lsub(const L4 &a, const L4 &b) {
    L4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
// a * 2^E, 0 <= E < 192 a multiple of 12: rotation by E / 24 limbs with 2^96 = -1, then (odd multiples of 12) a 12-bit shift that
// also renormalises: limb i keeps (l_i mod 2^12) 2^12 and passes l_i >> 12 on
template <int E>
SR_HD L4 lshift(const L4 &a) {
    static_assert(E % 12 == 0 && E >= 0 && E < 192, "limb shifts are multiples of 12");
    constexpr int q = (E / 24) % 8, odd = (E / 12) & 1;
    L4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int src = ((i - q) % 8 + 8) % 8;          // limb index in the length-8 signed cycle: positions 4..7 are -l[0..3]
        r.l[i] = src < 4 ? a.l[src] : -a.l[src - 4];
    }
    if (odd) {
        L4 s;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int32_t carry = i == 0 ? -(r.l[3] >> 12) : (r.l[i - 1] >> 12);
            s.l[i] = ((r.l[i] & 0xFFF) << 12) + carry;
        }
        return s;
    }
    return r;
}
template <int E>
SR_HD void lbf_dif(L4 &a, L4 &b) {
    const L4 s = ladd(a, b), d = lsub(a, b);
    a = s;
    b = lshift<E>(d);
}
template <int HALF, int STEP, int BASE, int... Js>
SR_HD void ldif_group(L4 *x, std::integer_sequence<int, Js...>) {
    (lbf_dif<(STEP * Js) % 192>(x[BASE + Js], x[BASE + Js + HALF]), ...);
}
template <int HALF, int STEP, int... Bs>
SR_HD void ldif_stage(L4 *x, std::integer_sequence<int, Bs...>) {
    (ldif_group<HALF, STEP, Bs * 2 * HALF>(x, std::make_integer_sequence<int, HALF>{}), ...);
}
// one carry sweep: limbs back to 24 bits (+ a small signed excess in limb 0 from the wrap)
SR_HD L4 lnorm(const L4 &a) {
    L4 r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int32_t v = a.l[i] + c;
        r.l[i] = v & 0xFFFFFF;
        c = v >> 24;
    }
    r.l[0] -= c;  // 2^96 = -1
    return r;
}
SR_HD void ldft16(L4 *x) {  // omega_16 = 2^12; a stage adds one bit per limb: 24 + 4 stays far below 31
    ldif_stage<8, 12>(x, std::make_integer_sequence<int, 1>{});
    ldif_stage<4, 24>(x, std::make_integer_sequence<int, 2>{});
    ldif_stage<2, 48>(x, std::make_integer_sequence<int, 4>{});
    ldif_stage<1, 96>(x, std::make_integer_sequence<int, 8>{});
}
// general product with a table element given as four unsigned 24-bit limbs: 16 multiply-adds, then the carry sweep
SR_HD L4 lmul(const L4 &a, const L4 &w) {
    int64_t c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t p = (int64_t)a.l[i] * w.l[j];
            if (i + j < 4) c[i + j] += p;
            else c[i + j - 4] -= p;
        }
    L4 r;
    int64_t carry = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int64_t v = c[i] + carry;
        r.l[i] = (int32_t)(v & 0xFFFFFF);
        carry = v >> 24;
    }
    // carry < 2^33: fold it twice more through limbs 0 and 1 (2^96 = -1)
    const int64_t v0 = (int64_t)r.l[0] - carry;
    r.l[0] = (int32_t)(v0 & 0xFFFFFF);
    r.l[1] += (int32_t)(v0 >> 24);
    return r;
}
__device__ __forceinline__ void lgeneral_layer(L4 *x, const u64 *__restrict__ tab, int lane) {
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = lmul(x[i], to_limbs(tab[i * 64 + (lane & 63)]));
}
__global__ __launch_bounds__(256, 2) void variant2_limbs96(u64 *data, const u64 *__restrict__ tab, int reps) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    L4 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = to_limbs(data[t * 16 + i]);
    for (int r = 0; r < reps; r++) {
        ldft16(x);
        lgeneral_layer(x, tab, threadIdx.x);
        ldft16(x);
        lgeneral_layer(x, tab + 1024, threadIdx.x);
        ldft16(x);
        lgeneral_layer(x, tab + 2048, threadIdx.x);
        ldft16(x);
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] = lnorm(x[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) data[t * 16 + i] = from_limbs(x[i]);
}
// the same function as variant2 in canonical arithmetic (omega_16 = 2^12 like ldft16), for the on-device comparison
SR_HD void dft16_w12(u64 *x) {
    dif_stage<8, 12>(x, std::make_integer_sequence<int, 1>{});
    dif_stage<4, 24>(x, std::make_integer_sequence<int, 2>{});
    dif_stage<2, 48>(x, std::make_integer_sequence<int, 4>{});
    dif_stage<1, 96>(x, std::make_integer_sequence<int, 8>{});
}
__global__ __launch_bounds__(256, 4) void variant2_check(u64 *data, const u64 *__restrict__ tab, int reps) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    u64 x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = data[t * 16 + i];
    for (int r = 0; r < reps; r++) {
        dft16_w12(x);
        general_layer<16>(x, tab, threadIdx.x);
        dft16_w12(x);
        general_layer<16>(x, tab + 1024, threadIdx.x);
        dft16_w12(x);
        general_layer<16>(x, tab + 2048, threadIdx.x);
        dft16_w12(x);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) data[t * 16 + i] = x[i];
}
__global__ void junk_kernel(u64 *p, size_t n, u64 seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x % sr::Goldilocks::P;
    }
}
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 64;
    const size_t ncoef = (size_t)1 << 24;  // 2^24 coefficients resident in registers across the chip per launch
    u64 *d0, *d1, *tab;
    CK(hipMalloc(&d0, ncoef * 8)); CK(hipMalloc(&d1, ncoef * 8)); CK(hipMalloc(&tab, 8192 * 8));
    hipLaunchKernelGGL(junk_kernel, dim3(1024), dim3(256), 0, 0, tab, (size_t)8192, (u64)3);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto fill = [&]() {
        hipLaunchKernelGGL(junk_kernel, dim3(4096), dim3(256), 0, 0, d0, ncoef, (u64)1);
        hipLaunchKernelGGL(junk_kernel, dim3(4096), dim3(256), 0, 0, d1, ncoef, (u64)1);
    };
    // correctness of the limb arithmetic: one round of variant2 against the canonical form of the same network
    fill();
    hipLaunchKernelGGL(variant2_limbs96, dim3((unsigned)(ncoef / 16 / 256)), dim3(256), 0, 0, d0, tab, 1);
    hipLaunchKernelGGL(variant2_check, dim3((unsigned)(ncoef / 16 / 256)), dim3(256), 0, 0, d1, tab, 1);
    CK(hipDeviceSynchronize());
    {
        std::vector<u64> h0(1 << 16), h1(1 << 16);
        CK(hipMemcpy(h0.data(), d0, h0.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), d1, h1.size() * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < h0.size(); i++) bad += h0[i] != h1[i];
        printf("limb arithmetic vs canonical on %zu words: %zu mismatches\n", h0.size(), bad);
    }
    struct V { const char *name; int per_lane; } vs[3] = {{"V0 current 16/lane, 3 general layers", 16}, {"V1 radix-64 64/lane, 2 general layers", 64},
                                                           {"V2 four 24-bit limbs mod 2^96+1 16/lane", 16}};
    for (int v = 0; v < 3; v++) {
        fill();
        const unsigned blocks = (unsigned)(ncoef / vs[v].per_lane / 256);
        float best = 1e30f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(variant0_current, dim3(blocks), dim3(256), 0, 0, d0, tab, reps);
            if (v == 1) hipLaunchKernelGGL(variant1_radix64, dim3(blocks), dim3(256), 0, 0, d0, tab, reps);
            if (v == 2) hipLaunchKernelGGL(variant2_limbs96, dim3(blocks), dim3(256), 0, 0, d0, tab, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it && ms < best) best = ms;
        }
        // one round = one transform's worth of arithmetic (16 stages + its general layers) on every coefficient
        printf("%-44s %8.3f ms for %d rounds over 2^24 coefficients = %.2f ps per coefficient-transform; config-2 batch (3 x 2^30) = %.2f ms\n", vs[v].name,
               best, reps, best * 1e9 / reps / (double)ncoef, best / reps / (double)ncoef * 3.0 * 1073741824.0);
    }
    return 0;
}
