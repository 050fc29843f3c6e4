// Development aid: do the HBM-bound and VALU-bound halves of an 8-stage column pass overlap on gfx950?
// Times the strided kernels of ntt_goldilocks.hpp on junk data (timing only, results are not checked).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../stark_rings_amd/csrc/ntt_goldilocks.hpp"
using namespace sr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(uint64_t *p, size_t n, uint64_t mul) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (i * mul + 12345) % 0xFFFFFFFF00000001ull;
}
template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const size_t n = (size_t)1 << 30;  // 8 GiB
    uint64_t *d, *tw, *twist;
    CK(hipMalloc(&d, n * 8)); CK(hipMalloc(&tw, (size_t)8 << 20)); CK(hipMalloc(&twist, (size_t)8 << 20));
    fill<<<n / 256, 256>>>(d, n, 0x9E3779B97F4A7C15ull);
    fill<<<4096, 256>>>(tw, 1 << 20, 0xD1B54A32D192ED03ull);
    fill<<<4096, 256>>>(twist, 1 << 20, 0x94D049BB133111EBull);
    CK(hipDeviceSynchronize());
    const double gb = 2.0 * n * 8 / 1e9;
    auto rep = [&](const char *name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, gb / ms * 1e3); };
    {
        const int k = 16; const size_t npoly = n >> k;
        rep("strided M=4 fwd no twist k=16", timeit([&] { hipLaunchKernelGGL((gl::strided_kernel<4, 0, false>), dim3(npoly * ((1u << (k - 4)) >> 8)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided M=4 fwd twist k=16", timeit([&] { hipLaunchKernelGGL((gl::strided_kernel<4, 0, true>), dim3(npoly * ((1u << (k - 4)) >> 8)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided256 fwd no twist k=16", timeit([&] { hipLaunchKernelGGL((gl::strided256_kernel<0, false>), dim3(npoly * ((1u << (k - 8)) >> 4)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided256 inv no twist k=16", timeit([&] { hipLaunchKernelGGL((gl::strided256_kernel<1, false>), dim3(npoly * ((1u << (k - 8)) >> 4)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
    }
    {
        const int k = 20; const size_t npoly = n >> k;
        rep("strided256 fwd no twist k=20", timeit([&] { hipLaunchKernelGGL((gl::strided256_kernel<0, false>), dim3(npoly * ((1u << (k - 8)) >> 4)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided256 fwd twist k=20", timeit([&] { hipLaunchKernelGGL((gl::strided256_kernel<0, true>), dim3(npoly * ((1u << (k - 8)) >> 4)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided M=4 fwd no twist k=20 s_lo=0", timeit([&] { hipLaunchKernelGGL((gl::strided_kernel<4, 0, false>), dim3(npoly * ((1u << (k - 4)) >> 8)), dim3(256), 0, 0, d, k, 0, tw, twist); }));
        rep("strided M=4 fwd no twist k=20 s_lo=4", timeit([&] { hipLaunchKernelGGL((gl::strided_kernel<4, 0, false>), dim3(npoly * 16 * ((1u << (k - 8)) >> 8)), dim3(256), 0, 0, d, k, 4, tw, twist); }));
    }
    return 0;
}
