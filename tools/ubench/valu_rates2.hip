#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 32768
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b0 = a0 ^ 0x9e3779b9u;
  asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "vcc");
  for (int i = 0; i < ITER; i++) {
    if (OP == 0) asm volatile("v_add_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, %1, %4\n v_add_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 1) asm volatile("v_sub_u32_e32 %0, %0, %4\n v_sub_u32_e32 %1, %1, %4\n v_sub_u32_e32 %2, %2, %4\n v_sub_u32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 2) asm volatile("v_xor_b32_e32 %0, %0, %4\n v_xor_b32_e32 %1, %1, %4\n v_xor_b32_e32 %2, %2, %4\n v_xor_b32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 3) asm volatile("v_and_b32_e32 %0, %0, %4\n v_and_b32_e32 %1, %1, %4\n v_and_b32_e32 %2, %2, %4\n v_and_b32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 4) asm volatile("v_lshlrev_b32_e32 %0, 3, %0\n v_lshlrev_b32_e32 %1, 3, %1\n v_lshlrev_b32_e32 %2, 3, %2\n v_lshlrev_b32_e32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 5) asm volatile("v_lshrrev_b32_e32 %0, 3, %0\n v_lshrrev_b32_e32 %1, 3, %1\n v_lshrrev_b32_e32 %2, 3, %2\n v_lshrrev_b32_e32 %3, 3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 6) asm volatile("v_mov_b32_e32 %0, %4\n v_mov_b32_e32 %1, %4\n v_mov_b32_e32 %2, %4\n v_mov_b32_e32 %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 7) asm volatile("v_min_u32_e32 %0, %0, %4\n v_min_u32_e32 %1, %1, %4\n v_min_u32_e32 %2, %2, %4\n v_min_u32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 8) asm volatile("v_add_f32_e32 %0, %0, %4\n v_add_f32_e32 %1, %1, %4\n v_add_f32_e32 %2, %2, %4\n v_add_f32_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 9) asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 10) asm volatile("v_fmac_f32_e32 %0, %4, %4\n v_fmac_f32_e32 %1, %4, %4\n v_fmac_f32_e32 %2, %4, %4\n v_fmac_f32_e32 %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 11) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %4\n v_add_co_u32_e32 %1, vcc, %1, %4\n v_add_co_u32_e32 %2, vcc, %2, %4\n v_add_co_u32_e32 %3, vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 12) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %4, vcc\n v_addc_co_u32_e32 %1, vcc, %1, %4, vcc\n v_addc_co_u32_e32 %2, vcc, %2, %4, vcc\n v_addc_co_u32_e32 %3, vcc, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 13) asm volatile("v_add_co_u32_e64 %0, s[10:11], %0, %4\n v_add_co_u32_e64 %1, s[10:11], %1, %4\n v_add_co_u32_e64 %2, s[10:11], %2, %4\n v_add_co_u32_e64 %3, s[10:11], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 14) asm volatile("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 15) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %4\n v_cmp_lt_u32_e32 vcc, %1, %4\n v_cmp_lt_u32_e32 vcc, %2, %4\n v_cmp_lt_u32_e32 vcc, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 16) asm volatile("v_cmp_lt_u32_e64 s[10:11], %0, %4\n v_cmp_lt_u32_e64 s[10:11], %1, %4\n v_cmp_lt_u32_e64 s[10:11], %2, %4\n v_cmp_lt_u32_e64 s[10:11], %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 17) asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 18) asm volatile("v_add3_u32 %0, %0, %4, %4\n v_add3_u32 %1, %1, %4, %4\n v_add3_u32 %2, %2, %4, %4\n v_add3_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 19) asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 20) asm volatile("v_perm_b32 %0, %0, %4, %4\n v_perm_b32 %1, %1, %4, %4\n v_perm_b32 %2, %2, %4, %4\n v_perm_b32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 21) asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 22) asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 23) asm volatile("v_mul_u32_u24_e32 %0, %0, %4\n v_mul_u32_u24_e32 %1, %1, %4\n v_mul_u32_u24_e32 %2, %2, %4\n v_mul_u32_u24_e32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 24) asm volatile("v_and_or_b32 %0, %0, %4, %4\n v_and_or_b32 %1, %1, %4, %4\n v_and_or_b32 %2, %2, %4, %4\n v_and_or_b32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 25) asm volatile("v_xad_u32 %0, %0, %4, %4\n v_xad_u32 %1, %1, %4, %4\n v_xad_u32 %2, %2, %4, %4\n v_xad_u32 %3, %3, %4, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 26) asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 27) asm volatile("v_pk_mul_lo_u16 %0, %0, %4\n v_pk_mul_lo_u16 %1, %1, %4\n v_pk_mul_lo_u16 %2, %2, %4\n v_pk_mul_lo_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 28) asm volatile("v_mad_u32_u16 %0, %0, %4, %0\n v_mad_u32_u16 %1, %1, %4, %1\n v_mad_u32_u16 %2, %2, %4, %2\n v_mad_u32_u16 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP> double run(uint32_t *d, int blocks) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize(); hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }
int main() { hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); int blocks = p.multiProcessorCount * 8; uint32_t *d; hipMalloc(&d, (size_t)blocks * 1024);
  const char *names[] = {"v_add_u32_e32", "v_sub_u32_e32", "v_xor_b32_e32", "v_and_b32_e32", "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_mov_b32_e32", "v_min_u32_e32", "v_add_f32_e32", "v_fma_f32", "v_fmac_f32_e32", "v_add_co_u32_e32(vcc)", "v_addc_co_u32_e32(vcc)", "v_add_co_u32_e64(sgpr)", "v_cndmask_b32_e32(vcc)", "v_cmp_lt_u32_e32", "v_cmp_lt_u32_e64(sgpr)", "v_lshl_add_u32", "v_add3_u32", "v_bfe_u32", "v_perm_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_u32_u24_e32", "v_and_or_b32", "v_xad_u32", "v_pk_add_u16", "v_pk_mul_lo_u16", "v_mad_u32_u16"};
  double ms[29];
  ms[0] = run<0>(d, blocks);
  ms[1] = run<1>(d, blocks);
  ms[2] = run<2>(d, blocks);
  ms[3] = run<3>(d, blocks);
  ms[4] = run<4>(d, blocks);
  ms[5] = run<5>(d, blocks);
  ms[6] = run<6>(d, blocks);
  ms[7] = run<7>(d, blocks);
  ms[8] = run<8>(d, blocks);
  ms[9] = run<9>(d, blocks);
  ms[10] = run<10>(d, blocks);
  ms[11] = run<11>(d, blocks);
  ms[12] = run<12>(d, blocks);
  ms[13] = run<13>(d, blocks);
  ms[14] = run<14>(d, blocks);
  ms[15] = run<15>(d, blocks);
  ms[16] = run<16>(d, blocks);
  ms[17] = run<17>(d, blocks);
  ms[18] = run<18>(d, blocks);
  ms[19] = run<19>(d, blocks);
  ms[20] = run<20>(d, blocks);
  ms[21] = run<21>(d, blocks);
  ms[22] = run<22>(d, blocks);
  ms[23] = run<23>(d, blocks);
  ms[24] = run<24>(d, blocks);
  ms[25] = run<25>(d, blocks);
  ms[26] = run<26>(d, blocks);
  ms[27] = run<27>(d, blocks);
  ms[28] = run<28>(d, blocks);
  for (int i = 0; i < 29; i++) printf("%-28s %8.4f ms  %6.3f ns per wave-instr per SIMD\n", names[i], ms[i], ms[i] * 1e6 / (8.0 * ITER * 4));
  return 0; }
