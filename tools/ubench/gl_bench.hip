// Timing-only harness for the Goldilocks D = 2^16 kernels (junk data and tables: instruction streams and memory traffic do not
// depend on the values).  Build variants with -D flags and compare; correctness is checked through the library, not here.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DSR_...] -o gl_bench tools/ubench/gl_bench.hip && ./gl_bench [npoly] [reps]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
#ifdef PERSIST
#include "cols256p_experiment.hpp"
#endif
using namespace sr::gl;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void junk_kernel(u64 *p, size_t n, u64 seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u64 x = (i + seed) * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = x % sr::Goldilocks::P;
    }
}
int main(int argc, char **argv) {
    const size_t npoly = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    const int k = 16;
    const size_t n = npoly << k;
    u64 *a, *b, *tab;
    CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8));
    const size_t tabn = (3u << k) + 3 * 4096 + 4 * 256;
    CK(hipMalloc(&tab, tabn * 8));
    hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, a, n, 1);
    hipLaunchKernelGGL(junk_kernel, dim3(8192), dim3(256), 0, 0, b, n, 2);
    hipLaunchKernelGGL(junk_kernel, dim3(1024), dim3(256), 0, 0, tab, tabn, 3);
    CK(hipDeviceSynchronize());
    u64 *p = tab;
    Tables T{};
    T.twist_f = p; p += (1u << k); T.twist_i_plain = p; p += (1u << k); T.twist_i_mul = p; p += (1u << k);
    T.w1f = p; p += 4096; T.w1i = p; p += 4096; T.w1i_mul = p; p += 4096;
    T.w2f = p; p += 256; T.w2i = p; p += 256; T.wcf = p; p += 256; T.wci = p;
    hipEvent_t ev[5];
    for (auto &e : ev) CK(hipEventCreate(&e));
    #ifndef LCV
#define LCV 4
#endif
    const unsigned blocks = (unsigned)(npoly << (k - 12));
    const unsigned cblocks = (unsigned)(npoly << (k - 8 - LCV));
#ifdef PERSIST
    const unsigned ntiles = cblocks;
    const unsigned pgrid = PERSIST * (LCV == 6 ? 1 : LCV == 5 ? 2 : 4);  // PERSIST = number of CUs to fill
#define COLS(DIRV, buf, wcp, twp) hipLaunchKernelGGL((cols256p_kernel<DIRV, LCV>), dim3(pgrid < ntiles ? pgrid : ntiles), dim3(16 << LCV), 0, 0, buf, k, wcp, twp, ntiles)
#else
#define COLS(DIRV, buf, wcp, twp) hipLaunchKernelGGL((cols256_kernel<DIRV, LCV>), dim3(cblocks), dim3(16 << LCV), 0, 0, buf, buf, k, wcp, twp)
#endif
    double acc[4] = {0, 0, 0, 0};
    for (int r = -1; r < reps; r++) {
        CK(hipEventRecord(ev[0]));
        COLS(0, a, T.wcf, T.twist_f);
        CK(hipEventRecord(ev[1]));
        COLS(0, b, T.wcf, T.twist_f);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL((rows256_kernel<2>), dim3(blocks), dim3(256), 0, 0, a, b, a, T);
        CK(hipEventRecord(ev[3]));
        COLS(1, a, T.wci, T.twist_i_mul);
        CK(hipEventRecord(ev[4]));
        CK(hipDeviceSynchronize());
        if (r < 0) continue;
        for (int i = 0; i < 4; i++) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
    }
    printf("npoly %zu  cols_a %.3f  cols_b %.3f  rows %.3f  cols_inv %.3f  total %.3f ms\n", npoly, acc[0] / reps, acc[1] / reps,
           acc[2] / reps, acc[3] / reps, (acc[0] + acc[1] + acc[2] + acc[3]) / reps);
    return 0;
}
