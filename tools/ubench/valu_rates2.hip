#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 32768
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, b0 = a0 ^ 0x9e3779b9u;
  asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" :: "v"(a0), "v"(b0) : "vcc");
  uint32_t pv = 0x78000001u; asm volatile("" : "+v"(pv));
  asm volatile("s_mov_b32 s12, 0x87ffffff\n s_mov_b32 s13, 0xfffffff\n s_mov_b32 s14, 0x78000001" ::: "s12", "s13", "s14");
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
    // round 4: the same quick-class opcodes with a 32-bit LITERAL operand (8-byte encoding) against an SGPR and a VGPR operand
    if (OP == 29) asm volatile("v_add_u32_e32 %0, 0x87ffffff, %0\n v_add_u32_e32 %1, 0x87ffffff, %1\n v_add_u32_e32 %2, 0x87ffffff, %2\n v_add_u32_e32 %3, 0x87ffffff, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 30) asm volatile("v_add_u32_e32 %0, s12, %0\n v_add_u32_e32 %1, s12, %1\n v_add_u32_e32 %2, s12, %2\n v_add_u32_e32 %3, s12, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 31) asm volatile("v_and_b32_e32 %0, 0xfffffff, %0\n v_and_b32_e32 %1, 0xfffffff, %1\n v_and_b32_e32 %2, 0xfffffff, %2\n v_and_b32_e32 %3, 0xfffffff, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 32) asm volatile("v_and_b32_e32 %0, s13, %0\n v_and_b32_e32 %1, s13, %1\n v_and_b32_e32 %2, s13, %2\n v_and_b32_e32 %3, s13, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    // BabyBear's canonical subtract d = a - b; min(d, d + p): literal / SGPR / VGPR p  (3 VALU each, two per iteration)
    if (OP == 33) asm volatile("v_sub_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, 0x78000001, %0\n v_min_u32_e32 %0, %0, %1\n v_sub_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, 0x78000001, %2\n v_min_u32_e32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 34) asm volatile("v_sub_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, s14, %0\n v_min_u32_e32 %0, %0, %1\n v_sub_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, s14, %2\n v_min_u32_e32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 35) asm volatile("v_sub_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, %5, %0\n v_min_u32_e32 %0, %0, %1\n v_sub_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, %5, %2\n v_min_u32_e32 %2, %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(pv) : "vcc", "s10", "s11");
    // Stark relax step: carry = x >> 28 (arithmetic), x &= M28, next += carry: literal / SGPR mask (3 VALU each; one per iteration + one add)
    if (OP == 36) asm volatile("v_ashrrev_i32_e32 %1, 28, %0\n v_and_b32_e32 %0, 0xfffffff, %0\n v_add_u32_e32 %2, %2, %1\n v_ashrrev_i32_e32 %3, 28, %2\n v_and_b32_e32 %2, 0xfffffff, %2\n v_add_u32_e32 %0, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 37) asm volatile("v_ashrrev_i32_e32 %1, 28, %0\n v_and_b32_e32 %0, s13, %0\n v_add_u32_e32 %2, %2, %1\n v_ashrrev_i32_e32 %3, 28, %2\n v_and_b32_e32 %2, s13, %2\n v_add_u32_e32 %0, %0, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
    if (OP == 38) asm volatile("v_bfe_u32 %1, %0, 0, 28\n v_ashrrev_i32_e32 %3, 28, %0\n v_add_u32_e32 %2, %2, %3\n v_bfe_u32 %0, %2, 0, 28\n v_ashrrev_i32_e32 %3, 28, %2\n v_add_u32_e32 %1, %1, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0) : "vcc", "s10", "s11");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP> double run(uint32_t *d, int blocks) { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize(); hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }
int main() { hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); uint32_t *d; hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 1024);
  const char *names[] = {"v_add_u32_e32", "v_sub_u32_e32", "v_xor_b32_e32", "v_and_b32_e32", "v_lshlrev_b32_e32", "v_lshrrev_b32_e32", "v_mov_b32_e32", "v_min_u32_e32", "v_add_f32_e32", "v_fma_f32", "v_fmac_f32_e32", "v_add_co_u32_e32(vcc)", "v_addc_co_u32_e32(vcc)", "v_add_co_u32_e64(sgpr)", "v_cndmask_b32_e32(vcc)", "v_cmp_lt_u32_e32", "v_cmp_lt_u32_e64(sgpr)", "v_lshl_add_u32", "v_add3_u32", "v_bfe_u32", "v_perm_b32", "v_alignbit_b32", "v_mul_lo_u32", "v_mul_u32_u24_e32", "v_and_or_b32", "v_xad_u32", "v_pk_add_u16", "v_pk_mul_lo_u16", "v_mad_u32_u16",
    "v_add_u32_e32 v, LITERAL, v", "v_add_u32_e32 v, sN, v", "v_and_b32_e32 v, LITERAL, v", "v_and_b32_e32 v, sN, v",
    "bb sub (sub,add LIT,min)", "bb sub (sub,add sN,min)", "bb sub (sub,add vN,min)", "stark relax (ashr,and LIT,add)", "stark relax (ashr,and sN,add)", "stark relax (bfe,ashr,add)"};
  const int per_iter[] = {4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4, 4,4,4,4, 6,6,6,6,6,6};
  typedef double (*runner)(uint32_t *, int);
  const runner runs[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>, run<14>, run<15>, run<16>, run<17>, run<18>,
    run<19>, run<20>, run<21>, run<22>, run<23>, run<24>, run<25>, run<26>, run<27>, run<28>, run<29>, run<30>, run<31>, run<32>, run<33>, run<34>, run<35>, run<36>, run<37>, run<38>};
  const int nops = (int)(sizeof(runs) / sizeof(runs[0]));
  for (int wps = 8; wps >= 2; wps /= 2) {   // waves per SIMD: 256-lane workgroups = one wave per SIMD each
    const int blocks = p.multiProcessorCount * wps;
    printf("--- %d waves per SIMD (%d workgroups of 256) ---\n", wps, blocks);
    for (int i = 0; i < nops; i++) {
      double ms = runs[i](d, blocks);
      printf("%-34s %8.4f ms  %6.3f ns per wave-instr per SIMD\n", names[i], ms, ms * 1e6 / ((double)wps * ITER * per_iter[i]));
    }
  }
  return 0; }
