// Energy per instruction on a power-capped chip.  MI355X runs the ring product AT its 1.4 kW socket cap (tools/power_trace.sh:
// 1378 W, sclk 2.22 GHz against 2.4 nominal), so joules per instruction matter as much as issue cycles.
// Each op runs alone on every SIMD (8 waves per SIMD) for a chosen time while tools/power_trace.sh samples socket power and sclk;
// the binary prints wave-instructions per second.  (power - idle) / rate = joules per wave-instruction.
//   hipcc --offload-arch=gfx950 -O3 -o valu_energy tools/ubench/valu_energy.hip && tools/power_trace.sh out.txt ./valu_energy <op> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define ITER 8192
#define REP4(s) s s s s
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = threadIdx.x * 0x9e3779b9u + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b0 = a0 ^ 0x9e3779b9u, b1 = a1 ^ 0x85ebca6bu;
    uint64_t q0 = ((uint64_t)a0 << 32) | b0, q1 = ((uint64_t)a1 << 32) | b1, q2 = ((uint64_t)a2 << 32) | a3, q3 = ((uint64_t)a3 << 32) | a1;
    uint64_t m = ((uint64_t)b1 << 32) | a2;
    const uint32_t eps = 0xFFFFFFFFu;
    uint64_t sv;
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {  // v_add_u32
            asm volatile(REP4("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 1) {  // 64-bit add as a carry pair
            asm volatile(REP4("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %5, vcc\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
        } else if (OP == 2) {  // v_lshl_add_u64
            asm volatile(REP4("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m));
        } else if (OP == 3) {  // v_mad_u64_u32, general operands
            asm volatile(REP4("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %6, %5, %1\n v_mad_u64_u32 %2, s[10:11], %4, %6, %2\n v_mad_u64_u32 %3, s[10:11], %5, %5, %3\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(b0), "v"(b1) : "s10", "s11");
        } else if (OP == 4) {  // v_mad_u64_u32 by the constant 2^32 - 1 (the eps fold)
            asm volatile(REP4("v_mad_u64_u32 %0, s[10:11], %4, %5, %0\n v_mad_u64_u32 %1, s[10:11], %6, %5, %1\n v_mad_u64_u32 %2, s[10:11], %7, %5, %2\n v_mad_u64_u32 %3, s[10:11], %4, %5, %3\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "s"(eps), "v"(b0), "v"(b1) : "s10", "s11");
        } else if (OP == 5) {  // the same fold with shifts: (x << 32) - x + q as sub pair + add to the high word (3 instructions)
            asm volatile(REP4("v_sub_co_u32 %0, vcc, %0, %4\n v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n v_add_u32 %1, %1, %4\n v_sub_co_u32 %2, vcc, %2, %5\n v_subbrev_co_u32 %3, vcc, 0, %3, vcc\n v_add_u32 %3, %3, %5\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1) : "vcc");
        } else if (OP == 6) {  // v_cmp_le_u64 into an SGPR pair
            asm volatile(REP4("v_cmp_le_u64 s[10:11], %0, %4\n v_cmp_le_u64 s[12:13], %1, %4\n v_cmp_le_u64 s[14:15], %2, %4\n v_cmp_le_u64 s[16:17], %3, %4\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(m) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
        } else if (OP == 7) {  // v_lshl_add_u64 with three quarters of the lanes masked off (the EXEC-masked corrections)
            asm volatile("s_mov_b64 %4, exec\n s_mov_b64 exec, %6\n"
                         REP4("v_lshl_add_u64 %0, %0, 0, %5\n v_lshl_add_u64 %1, %1, 0, %5\n v_lshl_add_u64 %2, %2, 0, %5\n v_lshl_add_u64 %3, %3, 0, %5\n")
                         "s_mov_b64 exec, %4\n"
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "=&s"(sv) : "v"(m), "s"(0x1111111111111111ull));
        } else if (OP == 8) {  // v_lshlrev_b64
            asm volatile(REP4("v_lshlrev_b64 %0, 7, %0\n v_lshlrev_b64 %1, 9, %1\n v_lshlrev_b64 %2, 11, %2\n v_lshlrev_b64 %3, 13, %3\n")
                         : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            q0 ^= m; q1 ^= m; q2 ^= m; q3 ^= m;
        } else if (OP == 9) {  // v_alignbit_b32
            asm volatile(REP4("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 9\n v_alignbit_b32 %2, %2, %4, 11\n v_alignbit_b32 %3, %3, %4, 13\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 10) {  // v_cndmask_b32 on an SGPR mask
            asm volatile(REP4("v_cndmask_b32 %0, %0, %4, s[10:11]\n v_cndmask_b32 %1, %1, %4, s[10:11]\n v_cndmask_b32 %2, %2, %4, s[10:11]\n v_cndmask_b32 %3, %3, %4, s[10:11]\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "s10", "s11");
        } else if (OP == 11) {  // v_mul_lo_u32 + v_mul_hi_u32 (a 32 x 32 -> 64 product without the add)
            asm volatile(REP4("v_mul_lo_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0));
        } else if (OP == 12) {  // s_nop only: an idle but resident wave
            asm volatile(REP4("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3);
}
template <int OP>
double run(uint32_t *d, int blocks, double seconds, double per_iter) {
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    while (el < seconds) {
        for (int i = 0; i < 8; i++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, (uint32_t)launches + i);
        hipDeviceSynchronize();
        launches += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // wave-instructions per second over the whole chip
    return launches * (double)blocks * 4 * ITER * per_iter / el;
}
int main(int argc, char **argv) {
    const int op = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 1.0;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8;
    uint32_t *d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    const char *names[] = {"v_add_u32", "v_add_co+v_addc_co", "v_lshl_add_u64", "v_mad_u64_u32", "v_mad_u64_u32 x eps", "eps fold by sub pair + add (3 instr)",
                           "v_cmp_le_u64", "v_lshl_add_u64, 1 lane in 4 active", "v_lshlrev_b64", "v_alignbit_b32", "v_cndmask_b32", "v_mul_lo+v_mul_hi", "s_nop"};
    double r = 0;
    switch (op) {
        case 0: r = run<0>(d, blocks, secs, 16); break;
        case 1: r = run<1>(d, blocks, secs, 16); break;
        case 2: r = run<2>(d, blocks, secs, 16); break;
        case 3: r = run<3>(d, blocks, secs, 16); break;
        case 4: r = run<4>(d, blocks, secs, 16); break;
        case 5: r = run<5>(d, blocks, secs, 24); break;
        case 6: r = run<6>(d, blocks, secs, 16); break;
        case 7: r = run<7>(d, blocks, secs, 16); break;
        case 8: r = run<8>(d, blocks, secs, 16); break;
        case 9: r = run<9>(d, blocks, secs, 16); break;
        case 10: r = run<10>(d, blocks, secs, 16); break;
        case 11: r = run<11>(d, blocks, secs, 16); break;
        case 12: r = run<12>(d, blocks, secs, 16); break;
        default: return 1;
    }
    printf("op %d %-40s %.4e wave-instructions/s  (%.3f per SIMD per ns)\n", op, names[op], r, r / (p.multiProcessorCount * 4) / 1e9);
    return 0;
}
