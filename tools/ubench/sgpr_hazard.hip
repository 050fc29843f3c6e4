// Development aid: does a SALU write to an SGPR pair that a v_mad_u64_u32 carry-out (sdst) was written to a few instructions
// earlier survive, i.e. is VALU-sdst -> SALU write-after-write ordered on gfx950?  Mirrors the sequence hipcc produced in
// gl::strided_kernel<4,0,true> around Goldilocks::mad_eps_fix.  Writes values only (no addressing depends on the result).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
template <int PAD>
__global__ void probe(uint64_t *out, const uint64_t *l2s, const uint32_t *hls, int iters) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t l2 = l2s[i];
    uint32_t hl = hls[i];
    uint64_t acc = 0;
    for (int it = 0; it < iters; it++) {
        uint64_t r;
        uint32_t a0, a1;
        asm volatile(
            "v_mad_u64_u32 v[10:11], s[20:21], %[hl], -1, %[l2]\n\t"
            "s_nop 0\n\t"
            "v_add_co_u32_e64 v12, s[22:23], -1, v10\n\t"
            "s_nop 1\n\t"
            "v_addc_co_u32_e64 v13, s[22:23], 0, v11, s[22:23]\n\t"
            "s_or_b64 s[22:23], s[22:23], s[20:21]\n\t"
            "v_cndmask_b32_e64 %[a0], v10, v12, s[22:23]\n\t"
            "v_cndmask_b32_e64 %[a1], v11, v13, s[22:23]\n\t"
            ".if %[pad] > 0\n\t"
            "s_nop %[pad] - 1\n\t"
            ".endif\n\t"
            "s_or_b32 s20, %[k], 1\n\t"
            "s_mov_b32 s21, 0\n\t"
            "s_lshl_b64 s[20:21], s[20:21], 15\n\t"
            "v_lshl_add_u64 %[r], %[base], 0, s[20:21]\n\t"
            : [r] "=&v"(r), [a0] "=&v"(a0), [a1] "=&v"(a1)
            : [hl] "v"(hl), [l2] "v"(l2), [base] "v"(acc), [k] "s"(it * 2), [pad] "n"(PAD)
            : "v10", "v11", "v12", "v13", "s20", "s21", "s22", "s23");
        // expected: r = acc + ((2*it | 1) << 15)
        const uint64_t want = acc + ((uint64_t)((it * 2) | 1) << 15);
        if (r != want) {
            out[i] = r ^ want;  // record the first discrepancy
            return;
        }
        acc = (acc + a0 + a1) & 0xFFFFFFFFull;
        l2 += 0x9E3779B97F4A7C15ull;
        hl = hl * 1664525u + 1013904223u;
    }
    out[i] = 0;
}
template <int PAD>
int run(uint64_t *d_out, uint64_t *d_l2, uint32_t *d_hl, size_t n) {
    hipMemset(d_out, 0xFF, n * 8);
    hipLaunchKernelGGL(probe<PAD>, dim3(n / 256), dim3(256), 0, 0, d_out, d_l2, d_hl, 2000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
    std::vector<uint64_t> h(n);
    hipMemcpy(h.data(), d_out, n * 8, hipMemcpyDeviceToHost);
    size_t bad = 0; uint64_t first = 0;
    for (auto v : h) if (v) { if (!bad) first = v; bad++; }
    printf("pad %d: %zu of %zu lanes saw a wrong SGPR-derived value (first xor %016llx)\n", PAD, bad, n, (unsigned long long)first);
    return 0;
}
int main() {
    const size_t n = 256 * 4096;
    std::vector<uint64_t> l2(n); std::vector<uint32_t> hl(n);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; l2[i] = s | 0xF000000000000000ull; hl[i] = (uint32_t)(s >> 11) | 0x80000000u; }
    uint64_t *d_out, *d_l2; uint32_t *d_hl;
    hipMalloc(&d_out, n * 8); hipMalloc(&d_l2, n * 8); hipMalloc(&d_hl, n * 4);
    hipMemcpy(d_l2, l2.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(d_hl, hl.data(), n * 4, hipMemcpyHostToDevice);
    run<0>(d_out, d_l2, d_hl, n);
    run<2>(d_out, d_l2, d_hl, n);
    run<8>(d_out, d_l2, d_hl, n);
    return 0;
}
