// Micro-benchmark: issue rate of the integer VALU instructions the modular arithmetic is built from
// (gfx950).  Each kernel runs ITER iterations of UNROLL independent instructions per lane; the host
// prints cycles per wave-instruction per SIMD, measured with full occupancy (8 waves/SIMD) so that
// dependency latency is hidden and the number is the pipe's throughput.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 32768

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b0 = a0 ^ 0x9e3779b9u, b1 = a1 ^ 0x85ebca6bu, b2 = a2 ^ 0xc2b2ae35u, b3 = a3 ^ 0x27d4eb2fu;
    uint64_t q0 = ((uint64_t)a0 << 32) | b0, q1 = ((uint64_t)a1 << 32) | b1, q2 = ((uint64_t)a2 << 32) | b2, q3 = ((uint64_t)a3 << 32) | b3;
    uint64_t m = ((uint64_t)b3 << 32) | a2;
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {  // v_mad_u64_u32
            asm volatile("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %4, %6, %1\n v_mad_u64_u32 %2, s[10:11], %4, %7, %2\n v_mad_u64_u32 %3, s[10:11], %4, %8, %3"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s10", "s11");
        } else if (OP == 1) {  // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 2) {  // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 3) {  // v_lshl_add_u64 (64-bit add)
            asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m));
        } else if (OP == 4) {  // v_add_co_u32 + v_addc_co_u32 pair (counted as 2 instructions)
            asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
        } else if (OP == 5) {  // v_cmp_lt_u64
            asm volatile("v_cmp_lt_u64 s[10:11], %0, %4\n v_cmp_lt_u64 s[12:13], %1, %4\n v_cmp_lt_u64 s[14:15], %2, %4\n v_cmp_lt_u64 s[16:17], %3, %4"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
        } else if (OP == 6) {  // v_cndmask_b32 with vcc
            asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc");
        } else if (OP == 7) {  // v_lshlrev_b64
            asm volatile("v_lshlrev_b64 %0, 7, %0\n v_lshlrev_b64 %1, 7, %1\n v_lshlrev_b64 %2, 7, %2\n v_lshlrev_b64 %3, 7, %3"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
        } else if (OP == 8) {  // v_alignbit_b32
            asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 9) {  // v_add_u32 (baseline full rate)
            asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 10) {  // v_mul_u32_u24
            asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 11) {  // v_mad_u32_u24
            asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 12) {  // v_add3_u32
            asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));
        } else if (OP == 13) {  // v_mul_hi_u32_u24
            asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 14) {  // v_cndmask_b32 with an SGPR-pair mask
            asm volatile("v_cndmask_b32 %0, %0, %4, s[10:11]\n v_cndmask_b32 %1, %1, %4, s[10:11]\n v_cndmask_b32 %2, %2, %4, s[10:11]\n v_cndmask_b32 %3, %3, %4, s[10:11]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s10", "s11");
        } else if (OP == 15) {  // v_dot4_u32_u8
            asm volatile("v_dot4_u32_u8 %0, %0, %4, %0\n v_dot4_u32_u8 %1, %1, %4, %1\n v_dot4_u32_u8 %2, %2, %4, %2\n v_dot4_u32_u8 %3, %3, %4, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 16) {  // v_sub_co_u32 + v_subb_co_u32 via explicit SGPR carry (not vcc)
            asm volatile("v_sub_co_u32 %0, s[10:11], %0, %4\n v_subb_co_u32 %1, s[10:11], %1, %5, s[10:11]\n v_sub_co_u32 %2, s[12:13], %2, %4\n v_subb_co_u32 %3, s[12:13], %3, %5, s[12:13]"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "s10", "s11", "s12", "s13");
        } else if (OP == 17) {  // v_fma_f64
            double d0 = __longlong_as_double(q0), d1 = __longlong_as_double(q1), d2 = __longlong_as_double(q2), d3 = __longlong_as_double(q3), dm = 1.0000001;
            asm volatile("v_fma_f64 %0, %0, %4, %0\n v_fma_f64 %1, %1, %4, %1\n v_fma_f64 %2, %2, %4, %2\n v_fma_f64 %3, %3, %4, %3"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            q0 = __double_as_longlong(d0); q1 = __double_as_longlong(d1); q2 = __double_as_longlong(d2); q3 = __double_as_longlong(d3);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3);
}

template <int OP>
double run(uint32_t *d_out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    int blocks = cus * 8;  // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    uint32_t *d_out;
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_add_co+v_addc_co (per instr)",
                           "v_cmp_lt_u64", "v_cndmask_b32 (vcc)", "v_lshlrev_b64", "v_alignbit_b32", "v_add_u32", "v_mul_u32_u24",
                           "v_mad_u32_u24", "v_add3_u32", "v_mul_hi_u32_u24", "v_cndmask_b32 (sgpr mask)", "v_dot4_u32_u8",
                           "v_sub_co+v_subb_co sgpr carry (per instr)", "v_fma_f64"};
    double ms[18];
    ms[0] = run<0>(d_out, blocks); ms[1] = run<1>(d_out, blocks); ms[2] = run<2>(d_out, blocks); ms[3] = run<3>(d_out, blocks);
    ms[4] = run<4>(d_out, blocks); ms[5] = run<5>(d_out, blocks); ms[6] = run<6>(d_out, blocks); ms[7] = run<7>(d_out, blocks);
    ms[8] = run<8>(d_out, blocks); ms[9] = run<9>(d_out, blocks); ms[10] = run<10>(d_out, blocks); ms[11] = run<11>(d_out, blocks);
    ms[12] = run<12>(d_out, blocks); ms[13] = run<13>(d_out, blocks); ms[14] = run<14>(d_out, blocks); ms[15] = run<15>(d_out, blocks);
    ms[16] = run<16>(d_out, blocks); ms[17] = run<17>(d_out, blocks);
    double clk_ghz = p.clockRate / 1e6;
    printf("device %s, %d CUs, clockRate %.2f GHz (nominal; DVFS may lower it)\n", p.name, cus, clk_ghz);
    printf("%-44s %10s %22s\n", "instruction", "ms", "ns per wave-instr per SIMD");
    for (int i = 0; i < 18; i++) {
        // per SIMD: 8 waves x ITER x 4 instructions
        double instr = 8.0 * ITER * 4;
        double ns = ms[i] * 1e6 / instr;
        printf("%-44s %10.4f %14.3f  (~%.2f cyc @2.4GHz)\n", names[i], ms[i], ns, ns * 2.4);
    }
    return 0;
}
