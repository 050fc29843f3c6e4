// Joules per byte of traffic that stays in an XCD's L2 (run under tools/power_trace.sh): every workgroup adds two 64 KiB regions of
// its own into a third, over and over (regions: 192 KiB per workgroup, 1 024 workgroups -> 24 MiB per XCD... too much; so GROUPS
// workgroups share a region set and the total working set is argv[2] MiB).  Plain loads / stores (no non-temporal hint).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void l2loop(u64x2 *buf, size_t region16, int iters, int nregions) {
    // region = 3 * region16 16-byte words: a, b, c; workgroup w works on region (w % nregions); XCD = w % 8 keeps its own regions
    const size_t r = blockIdx.x % nregions;
    u64x2 *a = buf + r * 3 * region16, *b = a + region16, *c = b + region16;
    for (int it = 0; it < iters; it++)
        for (size_t i = threadIdx.x; i < region16; i += 256) {
            u64x2 x = a[i];
            const u64x2 y = b[i];
            x += y;
            c[i] = x;
        }
}
int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    const size_t total_mib = argc > 2 ? strtoull(argv[2], 0, 10) : 16;
    const size_t region16 = 4096;                              // 64 KiB per stream, 192 KiB per region
    const int nregions = (int)(total_mib * 1024 / 192);        // regions are picked by blockIdx % nregions
    u64x2 *buf;
    if (hipMalloc(&buf, (size_t)nregions * 3 * region16 * 16) != hipSuccess) return 1;
    hipMemset(buf, 1, (size_t)nregions * 3 * region16 * 16);
    hipDeviceSynchronize();
    const int iters = 64;
    auto t0 = std::chrono::steady_clock::now();
    long launches = 0; double el = 0;
    while (el < secs) {
        for (int i = 0; i < 4; i++) hipLaunchKernelGGL(l2loop, dim3(1024), dim3(256), 0, 0, buf, region16, iters, nregions);
        hipDeviceSynchronize();
        launches += 4;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double moved = (double)launches * 1024 * iters * 3 * region16 * 16;
    printf("working set %zu MiB (%d regions): %.2f TB/s through the load/store path\n", total_mib, nregions, moved / el / 1e12);
    return 0;
}
