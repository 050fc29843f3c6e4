// Test of the test (round 4): the invariant counters of the checking build (csrc/fields.hpp: repcheck, -DSR_GL_CHECK_REPS) must FIRE when a
// precondition is violated on purpose, each one by the routine it guards, and stay silent on legal operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSR_GL_CHECK_REPS -o repcheck_selftest tools/ubench/repcheck_selftest.hip && ./repcheck_selftest
#include <cstdio>
#include <cstring>
#include "../../stark_rings_amd/csrc/fields.hpp"
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
#include "../../stark_rings_amd/csrc/stark_lazy.hpp"
using G = sr::Goldilocks;
typedef uint64_t u64;
// mode 0: legal operands everywhere; 1: canonical add handed p; 2: lazy legs with a non-canonical t that makes the sum wrap twice;
// 3: ... that makes the difference borrow twice; 4: a word >= p through st_result; 5: a canonical butterfly stage (dit_phased, twiddle 1) fed p + 1;
// 6: a nine-limb Stark sum whose limbs leave int32; 7: a nine-limb Stark product whose operands could overflow the column accumulator
__global__ void k(int mode, u64 *out) {
    const u64 P = G::P, lane = threadIdx.x;
    u64 a = 5 + lane, b = P - 7 - lane, s, d;
    if (mode == 1) a = P;
    u64 r = G::add(a % (mode == 1 ? ~0ull : P), b);
    u64 la = 123456789ull * (lane + 1), t = P - 1 - lane;          // any a, canonical t: legal
    if (mode == 2) { la = ~0ull; t = ~0ull; }                      // a + t carries and lands above 2^64 - eps: + eps wraps again
    if (mode == 3) { la = 0; t = ~0ull; }                          // a - t borrows and lands below eps: + p borrows again
    G::addsub_lazy(la, t, s, d);
    u64 x[16];
    for (int i = 0; i < 16; i++) x[i] = (lane * 16 + i) * 0x9E3779B97F4A7C15ull % P;
    if (mode == 5) x[3] = P + 1;
    sr::gl::dft16_fwd_dit<sr::gl::kPhased, true>(x);               // stage 0 is all twiddle-1 (canonical) butterflies
    sr::S9 sa, sb;                                                  // legal: limbs of a weakly reduced element and of a six-stage sum
    for (int i = 0; i < 9; i++) {
        sa.l[i] = (int)(6u * (1u << 28) - 1u - lane);
        sb.l[i] = (int)((1u << 28) - 1u - i);
    }
    if (mode == 6) sa.l[4] = 2147483647 - 20;                      // + (2^28 - 5) leaves int32
    if (mode == 7) sb.l[2] = 1 << 30;                              // 9 * (6 * 2^28) * 2^30 > 2^63 (the sum leaves int32 as well)
    const sr::S9 ss = sr::StarkL::add(sa, sb), sp = sr::StarkL::mul_tw(sa, sb);
    u64 res = r ^ s ^ d ^ (u64)(unsigned)ss.l[3] ^ (u64)(unsigned)sp.l[5];
    for (int i = 0; i < 16; i++) res ^= x[i];
    sr::gl::st_result(out + lane, mode == 4 ? P + lane : res % P);
}
int main() {
    u64 *out;
    if (hipMalloc(&out, 64 * 8) != hipSuccess) return 2;
    const char *what[] = {"legal operands", "add(p, b)", "lazy sum wrapping twice", "lazy difference borrowing twice", "st_result(p + lane)",
                          "twiddle-1 butterfly fed p + 1", "Stark limb sum beyond int32", "Stark product operand of 2^30"};
    const int expect_idx[] = {-1, 0, 1, 2, 3, 0, 5, 6};
    int bad = 0;
    for (int mode = 0; mode < 8; mode++) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0}, h[8];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(sr::repcheck::g_counters), z, sizeof(z));
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, out);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(sr::repcheck::g_counters), sizeof(h));
        bool ok = true;
        for (int i = 0; i < 7; i++) {
            const bool want = i == expect_idx[mode] || (mode == 2 && i == 0) || (mode == 3 && i == 0) || (mode == 7 && i == 5);  // modes 2, 3 also hand a non-canonical t
            if (i == expect_idx[mode] && h[i] == 0) ok = false;      // the guarded counter must fire ...
            if (!want && h[i] != 0) ok = false;                      // ... and nothing else may
        }
        printf("%-34s counters %llu %llu %llu %llu %llu %llu %llu  high-water %llu  %s\n", what[mode], h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7],
               ok ? "ok" : "WRONG");
        bad += !ok;
    }
    printf("repcheck_selftest: %d wrong\n", bad);
    return bad != 0;
}
