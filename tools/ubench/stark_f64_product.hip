// "try the multiply the chip is better at" -- a 5 x 52-bit-limb product on v_fma_f64 hi/lo pairs against the
// nine 28-bit-limb integer product of stark_lazy.hpp (81 v_mad_i64_i32 on one accumulator chain).  MI355X issues v_fma_f64 at the
// same ~4.2 cycles per wave-instruction as any other VALU op and v_mad_*64_*32 at ~5.1 (profiles/r01/valu_issue_rates_gfx950.txt).
//
// What is timed is the PRODUCT PART ONLY, the part where doubles could win: all limb products of a 260-bit by 260-bit multiplication,
// accumulated per column, no reduction and no carries.
//   int28:  nine signed 28-bit limbs, 81 multiply-adds into 17 64-bit column sums (as StarkL::mul_tw does before its reduction steps)
//   f64x52: five limbs below 2^52 in doubles.  A limb product has up to 104 bits: hi = fma(a, b, 2^104) keeps the bits above 2^52
//           (round to zero), lo = fma(a, b, (2^104 + 2^52) - hi) the rest; both are accumulated per column as INTEGERS on their
//           bit patterns (same exponent, so mantissas add), the trick of Emmart et al.: 25 x (2 fma + 2 64-bit adds) = 100 VALU.
// The double form then still owes what the integer form does not: the reduction's q_i have to be moved between integer and double
// (or-and-subtract, 2 VALU each way), the special modulus' products q_i 2^43 and q_i 17 2^36 do not fit a double limb without
// another hi / lo pair, and additions / subtractions of lazy elements are five v_add_f64 (5 x 4.2 cycles) against nine quick-class
// v_add_u32 (9 x 2.4): equal.  So if the product part alone is not clearly faster, the whole product cannot be.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o stark_f64_product tools/ubench/stark_f64_product.hip && ./stark_f64_product
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 4) void prod_int28(int32_t *data, int reps) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    int32_t a[9], w[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        a[i] = data[t * 18 + i];
        w[i] = data[t * 18 + 9 + i];
    }
    int64_t col[17];
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int k = 0; k < 17; k++) col[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) {
                uint64_t cy;
                asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(col[i + j]), "=s"(cy) : "v"(a[i]), "v"(w[j]));
            }
#pragma unroll
        for (int i = 0; i < 9; i++) a[i] = (int32_t)(col[i] ^ (col[i + 8] >> 7)) & 0xFFFFFFF;  // feed back: keeps the loop alive, limbs 28-bit
    }
#pragma unroll
    for (int i = 0; i < 9; i++) data[t * 18 + i] = a[i];
}
__global__ __launch_bounds__(256, 4) void prod_f64x52(double *data, int reps) {
    // round-to-zero for f64 (MODE.fp_round bits 3:2 = 3), as the hi / lo split wants
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3");
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double a[5], w[5];
#pragma unroll
    for (int i = 0; i < 5; i++) {
        a[i] = data[t * 10 + i];
        w[i] = data[t * 10 + 5 + i];
    }
    const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
    uint64_t hi[9], lo[9];
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int k = 0; k < 9; k++) hi[k] = lo[k] = 0;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const double h = __builtin_fma(a[i], w[j], C1);
                const double l = __builtin_fma(a[i], w[j], C2 - h);
                hi[i + j] += (uint64_t)__double_as_longlong(h);
                lo[i + j] += (uint64_t)__double_as_longlong(l);
            }
#pragma unroll
        for (int i = 0; i < 5; i++) {  // feed back: an integer below 2^52 as a double (or the exponent of 2^52 in, subtract 2^52)
            const uint64_t m = ((hi[i] ^ lo[i + 4] ^ hi[i + 4] ^ lo[i]) & 0xFFFFFFFFFFFFFull) | 0x4330000000000000ull;  // every column sum is used
            a[i] = __longlong_as_double((long long)m) - 0x1p52;
        }
    }
#pragma unroll
    for (int i = 0; i < 5; i++) data[t * 10 + i] = a[i];
}
int main(int argc, char **argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 256;
    const size_t lanes = (size_t)1 << 22;
    int32_t *di;
    double *dd;
    CK(hipMalloc(&di, lanes * 18 * 4));
    CK(hipMalloc(&dd, lanes * 10 * 8));
    {
        int32_t *hi_ = (int32_t *)malloc(lanes * 18 * 4);
        double *hd = (double *)malloc(lanes * 10 * 8);
        uint64_t s = 88172645463325252ull;
        for (size_t i = 0; i < lanes * 18; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hi_[i] = (int32_t)(s & 0xFFFFFFF); }
        for (size_t i = 0; i < lanes * 10; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hd[i] = (double)(s & 0xFFFFFFFFFFFFFull); }
        CK(hipMemcpy(di, hi_, lanes * 18 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dd, hd, lanes * 10 * 8, hipMemcpyHostToDevice));
        free(hi_); free(hd);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int v = 0; v < 2; v++) {
        float best = 1e30f;
        for (int it = 0; it < 4; it++) {
            CK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(prod_int28, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, di, reps);
            else hipLaunchKernelGGL(prod_f64x52, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, dd, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (it && ms < best) best = ms;
        }
        const double per = best * 1e6 / reps / (double)lanes * 64.0 * 1024.0;  // ns per wave-product per SIMD (1024 SIMDs)
        printf("%-44s %8.3f ms for %d x 2^22 products: %.1f ns per wave-product per SIMD = %.0f cycles at 2.1 GHz\n",
               v == 0 ? "int28: 81 v_mad_i64_i32, 17 column sums" : "f64x52: 25 x (2 v_fma_f64 + 2 64-bit adds)", best, reps, per, per * 2.1);
    }
    return 0;
}
