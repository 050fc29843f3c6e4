// Socket power of pure streaming (run under tools/power_trace.sh): a += b with one non-temporal 16-byte access per lane, repeated
// for the given time; prints the rate.  With profiles/r02/valu_energy.txt this splits the ring product's joules into
// arithmetic and memory traffic.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void add16(u64x2 *a, const u64x2 *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) {
        u64x2 x = __builtin_nontemporal_load(a + i);
        const u64x2 y = __builtin_nontemporal_load(b + i);
        x += y;
        __builtin_nontemporal_store(x, a + i);
    }
}
__global__ __launch_bounds__(256) void copy16(u64x2 *a, const u64x2 *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(b + i), a + i);
}
__global__ __launch_bounds__(256) void read16(u64x2 *a, const u64x2 *b, size_t n) {
    const size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    if (i < n) {
        const u64x2 y = __builtin_nontemporal_load(b + i);
        if (y.x == 0x123456789abcdefull) a[0] = y;
    }
}
int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 1.5;
    const size_t bytes = (size_t)8 << 30, n = bytes / 16;
    u64x2 *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    hipDeviceSynchronize();
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto t0 = std::chrono::steady_clock::now();
    long reps = 0; double el = 0;
    while (el < secs) {
        for (int i = 0; i < 8; i++) {
            if (mode == 0) hipLaunchKernelGGL(add16, dim3(blocks), dim3(256), 0, 0, a, b, n);
            else if (mode == 1) hipLaunchKernelGGL(copy16, dim3(blocks), dim3(256), 0, 0, a, b, n);
            else hipLaunchKernelGGL(read16, dim3(blocks), dim3(256), 0, 0, a, b, n);
        }
        hipDeviceSynchronize();
        reps += 8;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    const double moved = (mode == 0 ? 3.0 : mode == 1 ? 2.0 : 1.0) * bytes * reps;
    printf("mode %d (%s): %.2f TB/s\n", mode, mode == 0 ? "a += b" : mode == 1 ? "copy" : "read", moved / el / 1e12);
    return 0;
}
